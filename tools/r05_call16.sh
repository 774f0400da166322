# round 5, GPU call 16: what the in-library kernel timers (HIP event pairs around every kernel family) cost the timed region
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_hv2; mkdir -p $out
for i in 1 2 3; do for t in "" "--no-kernel-timers"; do
  python bench.py --steps 8 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-packed-leg --no-files-leg --no-filter-leg --no-verify $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('pipelined',{}); s=d.get('one_step_at_a_time',{})
print('timers ${t:-on}', 'value', d['value'], 'pipelined ms', p.get('ms_per_sample'), 'sequential ms', s.get('ms_per_sample'))"
done; done | tee $out/ab_timers.txt
