cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03_final; mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err
tail -c 200 $out/bench_c3.json
python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err
tail -c 600 $out/feed.txt
