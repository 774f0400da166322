cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_12; mkdir -p $out
timeout 900 python tools/gz_e2e_trace.py 2>&1 | tee $out/gz_trace.txt | cut -c1-220
