# A/B of environment settings with the DEFAULT (pipelined) C3 bench on one GPU box.  Usage: bash tools/ab_pipe.sh - VAR=value ...
out=gpurun_out/ab; mkdir -p $out
for i in $(seq ${AB_ROUNDS:-2}); do
for setting in "$@"; do
  envs=$(echo "$setting" | tr ';' ' '); [ "$setting" = "-" ] && envs=""
  env $envs python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-h2d --no-verify --no-sequential-leg ${AB_FLAGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$setting', 'step', d['ms_per_step'], 'value', d['value'], {k:v[0] for k,v in d['kernel_ms'].items()})"
done; done | tee -a $out/ab_pipe.txt
