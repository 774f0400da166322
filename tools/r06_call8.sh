cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_8; mkdir -p $out
export SYLPH_HIP_INFLATE_STATS=1
timeout 600 python tools/inflate_bench.py --mbp 250 --levels 1,6 --qual const --reps 1 2>&1 | tee $out/bench_const.txt
timeout 600 python tools/inflate_bench.py --mbp 60 --levels 6 --qual binned --reps 1 2>&1 | tee $out/bench_binned.txt
