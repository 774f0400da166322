# round 5, GPU call 14: the whole -m gpu suite + smoke() at the final tree, and the toolchain probe DESIGN §2 cites
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
{ echo "which cargo rustc:"; which cargo rustc; echo "(rc $?)"; ls ~/.cargo 2>&1 | head -3; } > $out/probe_toolchain.txt 2>&1; cat $out/probe_toolchain.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_final.txt 2>&1; grep -n "passed\|failed" $out/pytest_gpu_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke_final.txt 2>&1; tail -n 2 $out/smoke_final.txt
