cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04m; mkdir -p $out
cp sylph_amd/libsylph_hip.so /tmp/keep.so
for rep in 1 2; do for v in w5 w4 w3; do
  cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
  python bench.py --steps 8 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-verify --no-packed-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d.get('one_step_at_a_time',{}); print('$v', 'pipelined', d['value'], d['ms_per_sample'], 'sequential', o.get('ms_per_sample'), (o.get('kernel_ms') or {}).get('seeds'), d['kernel_ms'])"
done; done | tee $out/ab_waves.txt
cp /tmp/keep.so sylph_amd/libsylph_hip.so
