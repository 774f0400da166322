# after the partition rewrite: GPU suite, bench, one-sample timeline under the tracer
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03c; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; grep -n "passed\|failed" $out/pytest.txt; grep -n "Error\|error\|assert" $out/pytest.txt | head -20
python bench.py --steps 10 --warmup 2 --min-seconds 1.5 --no-cpu-baseline --no-h2d > $out/bench_c3.json 2> $out/bench_c3.err; tail -c 400 $out/bench_c3.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03c/bench_c3.json").read().strip().splitlines()[-1])
for k in ("value","mode","ms_per_sample","timed_region_s","sample_interval_ms","calibration","kernel_ms"): print(k, d.get(k))
for k in ("one_step_at_a_time",): print(k, {x:d[k].get(x) for x in ("value","ms_per_sample","kernel_ms","sketch_ms","profile_ms")})
print("verify", d.get("verify"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o c3 -- python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify > $out/bench_prof_seq.json 2> $out/prof.err
f=$(find $out/stats -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 5 --anchor reads_kernel > $out/step_timeline.md
rm -rf $out/stats
head -60 $out/step_timeline.md
