# round 5, GPU call 30: the whole -m gpu suite + smoke() at the round's last tree (host: gzip text through the device route, size rule)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|error" $out/pytest_gpu.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
d=/tmp/feed_bench
python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err; tail -n 1 $out/feed.txt
