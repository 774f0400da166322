cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_22; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
GZ_TRACE_HOLD_GB=120 timeout 1200 python tools/gz_e2e_trace.py 2>&1 | grep -v 'pool miss' | tee $out/gz_trace_hold.txt | grep '====\|main: done\|gzip inflated' | cut -c1-160
GZ_TRACE_HOLD_GB=120 SYLPH_HIP_CLEAN_EXIT=1 timeout 1200 python tools/gz_e2e_trace.py 2>&1 | grep -v 'pool miss' | tee $out/gz_trace_hold_clean_exit.txt | grep '====\|main: done' | cut -c1-160
