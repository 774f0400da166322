# round 5, GPU call 11: the database load once more (profile / query now leave through exit()), twice, and the CLI tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
timeout 900 python tools/db_load_bench.py > $out/db_load.txt 2> $out/db_load.err; cat $out/db_load.txt
timeout 900 python tools/db_load_bench.py > $out/db_load_2.txt 2>> $out/db_load.err; cat $out/db_load_2.txt
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest_cli_final.txt 2>&1; tail -1 $out/pytest_cli_final.txt
