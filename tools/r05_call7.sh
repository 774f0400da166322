# round 5, GPU call 7: the feed again after the release changes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05g; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest_cli.txt 2>&1; tail -3 $out/pytest_cli.txt
python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err; cat $out/feed.txt; tail -3 $out/feed.err
d=/tmp/feed_bench
echo "== gz pair" > $out/bringup_trace.txt
( time env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/s_1.fq.gz -2 $d/s_2.fq.gz -d $d/out ) 2>&1 | grep -v "stretch " | head -80 >> $out/bringup_trace.txt
echo "== four plain pairs" >> $out/bringup_trace.txt
( time env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/p0_1.fq $d/p1_1.fq $d/p2_1.fq $d/p3_1.fq -2 $d/p0_2.fq $d/p1_2.fq $d/p2_2.fq $d/p3_2.fq -d $d/out -t 1 ) 2>&1 | grep -v "stretch " | head -120 >> $out/bringup_trace.txt
cat $out/bringup_trace.txt | head -170
