#!/bin/bash
# round 4, call 11: the -m gpu suite with the filter dedup in, then the default bench line (with the default_pair_dedup leg)
mkdir -p gpurun_out/r04_a10
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_a10/pytest_gpu.txt
tail -5 gpurun_out/r04_a10/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r04_a10/bench_c3.json 2> gpurun_out/r04_a10/bench_c3.log
tail -3 gpurun_out/r04_a10/bench_c3.log
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_a10/bench_c3.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/sample", j["ms_per_sample"])
print("default_pair_dedup", json.dumps(j.get("default_pair_dedup"))[:1500])
PY
