# round 5, GPU call 31: the CLI tests once more (a named pipe as input; the gzip test stats before it reads)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest_cli_final.txt 2>&1; tail -n 12 $out/pytest_cli_final.txt
