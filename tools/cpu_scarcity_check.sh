# the default bench with the process confined to few host cores (what a CPU-limited container would do to it): the calibrated mode
# choice must keep `value` at or above the one-sample-at-a-time rate, whatever the pipeline's threads can still get
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/cpus; mkdir -p $out
for cores in 0 0-1 0-3 0-7; do
  taskset -c $cores python bench.py --steps 6 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-verify $BENCH_FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cores $cores', 'mode', d['mode'], 'value', d['value'], 'calibration', d['calibration'], 'p99', d['sample_interval_ms']['p99'])"
done | tee $out/cpu_scarcity.txt
