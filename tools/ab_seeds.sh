# A/B helper for the GPU box: runs the C3 bench one step at a time under each environment setting given as an argument
# ("VAR=value" or "VAR1=a,VAR2=b"; "-" = no setting) and prints step / stage / kernel times.  Usage: bash tools/ab_seeds.sh - SYLPH_HIP_HASH_VARIANT=2
out=gpurun_out/ab; mkdir -p $out
for setting in "$@"; do
  envs=$(echo "$setting" | tr ',' ' '); [ "$setting" = "-" ] && envs=""
  env $envs python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-h2d --no-verify --pipeline-depth 1 ${AB_FLAGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$setting', 'step', d['ms_per_step'], 'sketch', d['sketch_ms'], 'profile', d['profile_ms'], {k:v[0] for k,v in d['kernel_ms'].items()})"
done | tee -a $out/ab.txt
