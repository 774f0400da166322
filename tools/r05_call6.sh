# round 5, GPU call 6: the CLI after the feed changes (tests), the feed measurement, the first-sample trace again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest_cli.txt 2>&1; tail -5 $out/pytest_cli.txt
python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err; cat $out/feed.txt; tail -3 $out/feed.err
d=/tmp/feed_bench
for rep in 1 2 3; do
  echo "== run $rep (default flags, plain pair)"
  ( time env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/s_1.fq -2 $d/s_2.fq -d $d/out ) 2>&1 | grep -v "pgunzip\]" | head -70
done > $out/bringup_trace.txt 2>&1
echo "== gz pair" >> $out/bringup_trace.txt
( time env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/s_1.fq.gz -2 $d/s_2.fq.gz -d $d/out ) 2>&1 | grep -v "stretch " | head -80 >> $out/bringup_trace.txt
cat $out/bringup_trace.txt | head -150
