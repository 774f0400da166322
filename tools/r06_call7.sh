cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_9; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
export SYLPH_HIP_INFLATE_STATS=1
timeout 600 python tools/inflate_bench.py --mbp 250 --levels 1,6 --qual const --reps 2 2>&1 | tee $out/bench_const.txt
timeout 600 python tools/inflate_bench.py --mbp 60 --levels 6 --qual binned --bgzf --reps 2 2>&1 | tee $out/bench_binned.txt
