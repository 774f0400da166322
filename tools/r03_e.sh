# replay kernel: where the time goes (SYLPH_REPLAY_STAGE cuts the kernel short after a stage; results are wrong then — timing only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03e; mkdir -p $out
for st in 1 2 3 0; do
  SYLPH_REPLAY_STAGE=$st rocprofv3 --kernel-trace --output-format csv -d $out/tr$st -o s -- python tools/trace_run.py > /dev/null 2> /dev/null
  f=$(find $out/tr$st -name '*kernel_trace.csv' | head -1)
  echo "stage $st: $(python tools/kernel_trace_summary.py $f --tail-ms 6 --top 16 | grep -E 'bucket_replay|reads_kernel' | awk -F'|' '{print $2, $6}' | tr '\n' ' ')"
  rm -rf $out/tr$st
done | tee $out/stages.txt
