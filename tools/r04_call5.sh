# round 4, GPU call 5: row assembly with LDS-aggregated atomics + deferred seeding verdict: parity suite, A/B against the sorted path, timeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04e; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; grep -E "passed|failed|error" $out/pytest.txt | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --min-seconds 1.0 --no-cpu-baseline --no-h2d > $out/bench_rows.json 2> $out/bench_rows.err
SYLPH_HIP_HIT_SORT=1 timeout 600 python bench.py --steps 10 --warmup 3 --min-seconds 1.0 --no-cpu-baseline --no-h2d > $out/bench_sorted.json 2> $out/bench_sorted.err
python - <<PY
import json
for n in ("rows", "sorted"):
    try:
        d = json.loads(open("$out/bench_%s.json" % n).read().strip().splitlines()[-1])
        o = d.get("one_step_at_a_time", {})
        print(n, "value", d["value"], "ms", d["ms_per_sample"], "seq ms", o.get("ms_per_sample"), "profile_ms seq", o.get("profile_ms"), "kernels seq", o.get("kernel_ms"), "verify", d.get("verify", {}).get("mismatches"), "2bit", d.get("resident_2bit", {}).get("pipelined", {}).get("value"), d.get("resident_2bit", {}).get("table_equal_to_ascii"))
    except Exception as e:
        print(n, "failed", e)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o c3 -- python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-packed-leg > $out/bench_prof_seq.json 2> $out/prof.err
f=$(find $out/stats -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 5 --anchor reads_kernel > $out/step_timeline.md
rm -rf $out/stats
head -40 $out/step_timeline.md
