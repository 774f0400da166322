# round 5, GPU call 9: the whole -m gpu suite (two-rank bench tests follow the replicate default) and the database load again
# (the early-return fork is now `sketch` only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt | head -1; grep -n "passed\|failed" $out/pytest_gpu.txt
timeout 900 python tools/db_load_bench.py > $out/db_load.txt 2> $out/db_load.err; cat $out/db_load.txt
