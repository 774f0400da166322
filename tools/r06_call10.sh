cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_10; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
SYLPH_HIP_INFLATE_SMALL_RING=1 timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $out/pytest_small_ring.txt 2>&1; tail -3 $out/pytest_small_ring.txt
export SYLPH_HIP_INFLATE_STATS=1
timeout 600 python tools/inflate_bench.py --mbp 250 --levels 1,6 --qual const --reps 2 2>&1 | grep -v 'inflate\]' | tee $out/bench_const.txt
SYLPH_HIP_INFLATE_SMALL_RING=1 timeout 600 python tools/inflate_bench.py --mbp 250 --levels 1,6 --qual const --reps 2 2>&1 | grep -v 'inflate\]' | tee $out/bench_const_small_ring.txt
timeout 600 python tools/inflate_bench.py --mbp 60 --levels 6 --qual binned --bgzf --reps 2 2>&1 | grep -v 'inflate\]' | tee $out/bench_binned.txt
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -k "device_fastq_route or damaged_gzip or sketch_outputs or parallel_feed" > $out/pytest_cli.txt 2>&1; tail -5 $out/pytest_cli.txt
