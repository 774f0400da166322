# A/B on one box: seeding kernels on a low-priority stream (SYLPH_HIP_BULK_PRIORITY=low), contexts' own streams high priority
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03b; mkdir -p $out
python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
run() { # label, env..., args
  label=$1; shift
  env "$@" python bench.py --steps 10 --warmup 2 --mode pipelined --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-verify --no-second-leg $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$ARGS', 'value',d['value'],'ms/sample',d['ms_per_sample'],d['sample_interval_ms'],d['kernel_ms'])"
}
for rep in 1 2; do
for ARGS in "--sketch-workers 2 --pipeline-depth 4" "--sketch-workers 3 --pipeline-depth 6"; do
  run base A=1
  run bulk_low SYLPH_HIP_BULK_PRIORITY=low
  run bulk_low+high SYLPH_HIP_BULK_PRIORITY=low SYLPH_HIP_STREAM_PRIORITY=high
done; done | tee $out/ab.txt
