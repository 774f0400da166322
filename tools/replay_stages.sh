for st in 1 2 3 0; do
SYLPH_REPLAY_STAGE=$st python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage', $st, d['ms_per_step'], d['sketch_ms'], d['kernel_ms']['replay'])"
done
