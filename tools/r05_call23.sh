# round 5, GPU call 23: the CLI's device-side FASTQ route (SYLPH_HIP_FEED_DEVICE=1) — its test, the CLI suite, then the feed bench with both routes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_fastq; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_fastq.py -m gpu -x -q > $out/pytest_cli.txt 2>&1; tail -n 15 $out/pytest_cli.txt
for dev in 0 1 0 1; do
  echo "== SYLPH_HIP_FEED_DEVICE=$dev"; FEED_BENCH_ONLY=plain SYLPH_HIP_FEED_DEVICE=$dev python tools/feed_bench.py 3333334 2> $out/feed_dev$dev.err | tail -n 3
done > $out/feed_ab.txt 2>&1
cat $out/feed_ab.txt
d=/tmp/feed_bench
( for dev in 0 1; do echo "== four plain pairs, SYLPH_HIP_FEED_DEVICE=$dev"; ( time env SYLPH_HIP_FEED_DEVICE=$dev SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -t 1 -1 $d/p0_1.fq $d/p1_1.fq $d/p2_1.fq $d/p3_1.fq -2 $d/p0_2.fq $d/p1_2.fq $d/p2_2.fq $d/p3_2.fq -d $d/out$dev ) 2>&1 | grep -v "pgunzip\]" | head -150; done
  for i in 0 1 2 3; do cmp $d/out0/p${i}_1.fq.paired.sylsp $d/out1/p${i}_1.fq.paired.sylsp && echo "p$i identical"; done ) > $out/cli_four_pairs_trace.txt 2>&1
tail -n 8 $out/cli_four_pairs_trace.txt
