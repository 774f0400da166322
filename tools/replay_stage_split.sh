# replay stage split (SYLPH_REPLAY_STAGE: 1 = after gather + sub-range placement, 2 = after sorted write, 3 = after flags) for the 200x, 50x and 2x cases
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03_l; mkdir -p $out
for st in 0 1 2 3; do
  echo "== stage $st" >> $out/stages.txt
  SYLPH_REPLAY_STAGE=$st DEEP_CASES=all python tools/deep_coverage_check.py 2>/dev/null | grep "paired=True" >> $out/stages.txt
done
cat $out/stages.txt
