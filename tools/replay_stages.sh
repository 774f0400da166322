mkdir -p gpurun_out/s3b
for st in 1 2 3 0; do
SYLPH_REPLAY_STAGE=$st python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-h2d --no-verify --pipeline-depth 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage', $st, d['ms_per_step'], d['sketch_ms'], d['kernel_ms'])"
done > gpurun_out/s3b/stages.txt 2>&1
cat gpurun_out/s3b/stages.txt
