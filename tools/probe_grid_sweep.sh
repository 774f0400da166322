for g in 128 256 512 1024; do
SYLPH_HIP_PROBE_GRID=$g python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-h2d --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid', $g, d['ms_per_step'], d['profile_ms'], d['kernel_ms']['probe'])"
done
