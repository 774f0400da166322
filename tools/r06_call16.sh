cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_16; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -k "device_fastq or damaged or parallel_feed or several_gpus" > $out/pytest_cli.txt 2>&1; tail -3 $out/pytest_cli.txt
timeout 900 python tools/gz_e2e_trace.py 2>&1 | grep -v 'pool miss' | tee $out/gz_trace.txt | grep -v 'engine:\|main' | cut -c1-200
