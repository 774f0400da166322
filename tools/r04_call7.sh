cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04j; mkdir -p $out
for mask in "" 0:64 0:128 128:256; do
  SYLPH_BENCH_DB_CU_MASK=$mask python bench.py --steps 6 --warmup 2 --min-seconds 0.8 --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --sweep tools/r04_sweep3.json > $out/bench_mask_$mask.json 2> $out/bench_mask_$mask.err
  echo "== db cu_mask '$mask'"; grep "\[sweep\]" $out/bench_mask_$mask.err | grep "round\": 1"
  python -c "
import json; d=json.loads(open('$out/bench_mask_$mask.json').read().strip().splitlines()[-1]); print('main', d['value'], d['ms_per_sample'], 'seq', d['one_step_at_a_time']['ms_per_sample'], d['kernel_ms'])"
done
