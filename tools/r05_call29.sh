# round 5, GPU call 29: the inflated text of gzip files through the device-side route — CLI tests, then the feed bench (gzip legs included) with the route on / off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_fastq; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest_cli3.txt 2>&1; tail -n 3 $out/pytest_cli3.txt
for dev in 1 0; do
  echo "== SYLPH_HIP_FEED_DEVICE=$dev"; SYLPH_HIP_FEED_DEVICE=$dev python tools/feed_bench.py 3333334 2> $out/feed_gz_dev$dev.err | tail -n 1
done > $out/feed_gz_ab.txt 2>&1
cat $out/feed_gz_ab.txt
d=/tmp/feed_bench
( echo "== gz pair, device route"; ( time env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/s_1.fq.gz -2 $d/s_2.fq.gz -d $d/outg1 ) 2>&1 | grep -v "stretch " | head -70
  SYLPH_HIP_FEED_DEVICE=0 sylph_amd/sylph-hip sketch -1 $d/s_1.fq.gz -2 $d/s_2.fq.gz -d $d/outg0 > /dev/null 2>&1; cmp $d/outg0/s_1.fq.gz.paired.sylsp $d/outg1/s_1.fq.gz.paired.sylsp && echo "gz pair: identical sketches" ) > $out/cli_gz_device_trace.txt 2>&1
grep -n "device route\|real\|identical\|timing" $out/cli_gz_device_trace.txt
