cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_24; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
SYLPH_HIP_INFLATE_SMALL_RING=1 timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $out/pytest_small_ring.txt 2>&1; tail -2 $out/pytest_small_ring.txt
export SYLPH_HIP_INFLATE_STATS=1
timeout 600 python tools/inflate_bench.py --mbp 500 --levels 1,6 --qual const --reps 3 2>&1 | grep -v 'inflate\]   ' | tee $out/bench_const.txt | cut -c1-420
timeout 600 python tools/inflate_bench.py --mbp 120 --levels 6 --qual binned --bgzf --reps 3 2>&1 | grep -v 'inflate\]   ' | tee $out/bench_binned.txt | cut -c1-420
