# round 5, GPU call 33: the whole -m gpu suite + smoke() at the round's last commit
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|error" $out/pytest_gpu.txt | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
