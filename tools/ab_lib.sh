# A/B of two builds of libsylph_hip.so on ONE GPU box (boxes differ by a few per cent): sylph_amd/libsylph_hip.so.base and
# .so.new are copied over the library in turn, AB_ROUNDS times; the C3 bench runs one step at a time.  AB_FLAGS: extra bench flags.
out=gpurun_out/ab; mkdir -p $out
for i in $(seq ${AB_ROUNDS:-3}); do
  for v in base new; do
    cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2d --no-verify --pipeline-depth 1 ${AB_FLAGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'step', d['ms_per_step'], 'sketch', d['sketch_ms'], 'profile', d['profile_ms'], {k:v[0] for k,v in d['kernel_ms'].items()})"
  done
done | tee -a $out/ab_lib.txt
cp sylph_amd/libsylph_hip.so.new sylph_amd/libsylph_hip.so
