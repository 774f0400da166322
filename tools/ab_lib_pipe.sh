out=gpurun_out/ab; mkdir -p $out
for i in $(seq ${AB_ROUNDS:-3}); do
  for v in base new; do
    cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-verify --no-sequential-leg ${AB_FLAGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'step', d['ms_per_step'], 'value', d['value'], {k:v[0] for k,v in d['kernel_ms'].items()})"
  done
done | tee -a $out/ab_lib_pipe.txt
cp sylph_amd/libsylph_hip.so.new sylph_amd/libsylph_hip.so
