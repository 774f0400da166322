# A/B of two builds of libsylph_hip.so (sylph_amd/libsylph_hip.so.base / .new), alternated on ONE box (boxes differ by a few per cent):
# the default C3 bench, both modes, AB_ROUNDS times each.  gpurun -- 'bash tools/ab_lib_r03.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/ab; mkdir -p $out
for i in $(seq ${AB_ROUNDS:-3}); do
  for v in base new; do
    cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
    python bench.py --steps 6 --warmup 2 --min-seconds ${AB_SECONDS:-1.0} --no-cpu-baseline --no-h2d --no-verify ${AB_FLAGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('pipelined',{}); s=d.get('one_step_at_a_time',{})
print('$v', 'pipelined', p.get('value'), p.get('ms_per_sample'), 'sequential', s.get('ms_per_sample'), s.get('kernel_ms'))"
  done
done | tee -a $out/ab_lib_r03.txt
cp sylph_amd/libsylph_hip.so.base sylph_amd/libsylph_hip.so
