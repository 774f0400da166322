#!/bin/bash
# round 4, call 13: filter dedup parity + the bench line with its default_pair_dedup leg
mkdir -p gpurun_out/r04_a10
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "filter_dedup or synthetic" 2>&1 | tail -3
timeout 900 python bench.py --no-packed-leg --no-files-leg > gpurun_out/r04_a10/bench_c3_b.json 2> gpurun_out/r04_a10/bench_c3_b.log
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_a10/bench_c3_b.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/sample", j["ms_per_sample"], "one at a time", j["one_step_at_a_time"]["ms_per_sample"])
print("default_pair_dedup", json.dumps(j.get("default_pair_dedup"))[:1800])
PY
