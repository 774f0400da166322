# round 5, GPU call 26: the whole -m gpu suite with the device-side FASTQ route as the CLI's default
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_fastq; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_all.txt 2>&1; grep -n "passed\|failed\|error" $out/pytest_gpu_all.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
