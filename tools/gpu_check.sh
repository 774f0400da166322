cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03_n; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
python tools/deep_coverage_check.py 2>/dev/null | tee $out/deep.txt
DEEP_CASES=extreme python tools/deep_coverage_check.py 2>/dev/null | grep "paired=True" | tee -a $out/deep.txt
for st in 1 2 3; do echo "== stage $st"; SYLPH_REPLAY_STAGE=$st DEEP_CASES=all python tools/deep_coverage_check.py 2>/dev/null | grep "paired=True"; done | tee $out/stages.txt
