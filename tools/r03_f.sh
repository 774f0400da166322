cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03f; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; grep -n "passed\|failed" $out/pytest.txt; grep -n "Error\|assert" $out/pytest.txt | head -20
for st in 1 2 3 0; do
  SYLPH_REPLAY_STAGE=$st rocprofv3 --kernel-trace --output-format csv -d $out/tr$st -o s -- python tools/trace_run.py > /dev/null 2> /dev/null
  f=$(find $out/tr$st -name '*kernel_trace.csv' | head -1)
  echo "stage $st: $(python tools/kernel_trace_summary.py $f --tail-ms 6 --top 16 | grep -E 'bucket_replay|reads_kernel' | awk -F'|' '{print $2, $6}' | tr '\n' ' ')"
  rm -rf $out/tr$st
done | tee $out/stages.txt
for bt in 64 96 160 192; do
  SYLPH_BENCH_CTX_OPTIONS=bucket_target=$bt python bench.py --steps 5 --warmup 1 --min-seconds 0.4 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bucket_target',$bt,'ms/sample',d['ms_per_sample'],d['kernel_ms'])"
done | tee $out/bt.txt
python bench.py --steps 10 --warmup 2 --min-seconds 1.5 --no-cpu-baseline --no-h2d > $out/bench_c3.json 2> $out/bench_c3.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03f/bench_c3.json").read().strip().splitlines()[-1])
for k in ("value","mode","ms_per_sample","timed_region_s","sample_interval_ms","calibration","kernel_ms"): print(k, d.get(k))
for k in ("one_step_at_a_time",): print(k, {x:d[k].get(x) for x in ("value","ms_per_sample","kernel_ms","sketch_ms","profile_ms")})
print("verify", d.get("verify"))
PY
