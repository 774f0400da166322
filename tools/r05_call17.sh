# round 5, GPU call 17: bench.py with the dominant kernel's timers only in its timed region — the bench-driven GPU tests, then the PMC
# refresh + default line + long run of tools/r05_pmc_refresh.sh at this tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -k "bench or pipeline or hash_spellings" > $out/pytest_bench_tests.txt 2>&1; grep -n "passed\|failed" $out/pytest_bench_tests.txt
bash tools/r05_pmc_refresh.sh
