cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04k; mkdir -p $out
for bt in 64 96 128 160 192; do
  SYLPH_BENCH_CTX_OPTIONS=bucket_target=$bt python bench.py --steps 6 --warmup 2 --min-seconds 0.8 --no-cpu-baseline --no-h2d --no-verify --no-packed-leg > $out/bench_bt$bt.json 2> $out/bench_bt$bt.err
  python -c "
import json; d=json.loads(open('$out/bench_bt$bt.json').read().strip().splitlines()[-1]); o=d['one_step_at_a_time']; print('bucket_target $bt', 'pipelined', d['value'], d['ms_per_sample'], 'seq', o['ms_per_sample'], o['kernel_ms'])"
done
