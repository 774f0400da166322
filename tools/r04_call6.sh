# round 4, GPU call 6: parity suite with the new host paths, the default bench line with its files leg, feed + database-load measurements
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04f; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; grep -E "passed|failed|error" $out/pytest.txt | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
python - <<PY
import json
d = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
o = d.get("one_step_at_a_time", {})
print("value", d["value"], "ms", d["ms_per_sample"], "seq", o.get("ms_per_sample"), o.get("kernel_ms"), "verify", d.get("verify", {}).get("mismatches"), d.get("verify", {}).get("sample_table_equal"))
print("files", json.dumps(d.get("end_to_end_from_files")))
print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
SYLPH_HIP_FEED_TRACE=1 timeout 900 python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err; tail -2 $out/feed.txt; grep pgunzip $out/feed.err | head -12
timeout 900 python tools/db_load_bench.py > $out/db_load.txt 2> $out/db_load.err; tail -1 $out/db_load.txt; tail -3 $out/db_load.err
