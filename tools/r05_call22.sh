# round 5, GPU call 22: FASTQ text parsed on the device (csrc/fastq.hip) — its tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_fastq; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_fastq.py -m gpu -x -q > $out/pytest_fastq.txt 2>&1; tail -n 30 $out/pytest_fastq.txt
