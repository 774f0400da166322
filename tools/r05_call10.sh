# round 5, GPU call 10: the database load again (profile / query free their device memory themselves), the CLI tests, and the default bench
# line once more on the final tree (its roofline_a10.traffic from the corrected profiles/seeds_traffic.json)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest_cli_final.txt 2>&1; tail -1 $out/pytest_cli_final.txt
timeout 900 python tools/db_load_bench.py > $out/db_load.txt 2> $out/db_load.err; cat $out/db_load.txt
python bench.py --steps 20 --warmup 5 > $out/bench_c3_final.json 2> $out/bench_c3_final.err; tail -c 200 $out/bench_c3_final.json
