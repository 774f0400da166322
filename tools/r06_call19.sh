cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_19; mkdir -p $out
GZ_TRACE_READS=genome SYLPH_HIP_INFLATE_STATS=1 timeout 1200 python tools/gz_e2e_trace.py 2>&1 | grep -v 'pool miss' | tee $out/gz_trace_genome.txt | grep -v 'engine:' | cut -c1-220
