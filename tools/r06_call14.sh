cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_14; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
SYLPH_HIP_INFLATE_SMALL_RING=1 timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $out/pytest_small_ring.txt 2>&1; tail -2 $out/pytest_small_ring.txt
SYLPH_HIP_INFLATE_STATS=1 timeout 900 python tools/gz_e2e_trace.py 2>&1 | grep -v 'pool miss' | tee $out/gz_trace.txt | grep -v 'engine:\|main' | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -x -q -k "replicas or several_gpus or device_fastq or damaged or parallel_feed" > $out/pytest_misc.txt 2>&1; tail -3 $out/pytest_misc.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "read_sketch or extract or shapes" > $out/pytest_reads.txt 2>&1; tail -2 $out/pytest_reads.txt
