cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_23; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.err
python - <<'PY'
import json
o = json.load(open("gpurun_out/r06_23/bench.json"))
print(json.dumps({k: o.get(k) for k in ("value", "value_default_flags", "ms_per_step", "rates_gbp_per_s", "roofline", "cpu_baseline")}, indent=1)[:6000])
e = o.get("end_to_end_from_files", {})
print(json.dumps({k: e.get(k) for k in ("plain_one_sample", "plain_four_samples_one_command", "gz_one_sample", "gz_four_samples_one_command")}, indent=1))
PY
