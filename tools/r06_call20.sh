cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_20; mkdir -p $out
GZ_TRACE_HOLD_GB=120 SYLPH_HIP_TRACE=1 timeout 1200 python tools/gz_e2e_trace.py 2>&1 | grep -v 'pool miss' | tee $out/gz_trace_hold.txt | grep -v 'engine:' | cut -c1-220 | tail -120
