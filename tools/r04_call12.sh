#!/bin/bash
# round 4, call 12: kernel trace of the bench with the default_pair_dedup leg (what the a10 kernels cost)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r04_a10
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_a10/prof -o a10 -- python bench.py --steps 3 --no-packed-leg --no-cpu-baseline --no-files-leg > gpurun_out/r04_a10/bench_prof.json 2> gpurun_out/r04_a10/bench_prof.log
tail -2 gpurun_out/r04_a10/bench_prof.log
f=$(find gpurun_out/r04_a10/prof -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>7s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.1f}')
PY
find gpurun_out/r04_a10/prof -name "*.db" -delete; find gpurun_out/r04_a10/prof -name "*kernel_trace.csv" -delete
