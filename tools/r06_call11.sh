cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_11; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.err
python - <<'PY'
import json
o = json.load(open("gpurun_out/r06_11/bench.json"))
print(json.dumps({k: o.get(k) for k in ("value", "value_default_flags", "rates_gbp_per_s")}, indent=1))
print(json.dumps(o.get("end_to_end_from_files"), indent=1))
PY
