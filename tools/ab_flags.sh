out=gpurun_out/ab; mkdir -p $out
for i in 1 2; do
for fl in "--sketch-workers 2" "--sketch-workers 3" "--sketch-workers 4" "--sketch-workers 3 --pipeline-depth 5" "--sketch-workers 2 --pipeline-depth 4"; do
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-h2d --no-verify --no-sequential-leg $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$fl', 'step', d['ms_per_step'], 'value', d['value'], {k:v[0] for k,v in d['kernel_ms'].items()})"
done; done | tee -a $out/ab_flags.txt
