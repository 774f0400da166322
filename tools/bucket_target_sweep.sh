for t in 96 128 160 200; do
SYLPH_BENCH_CTX_OPTIONS=bucket_target=$t python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('target', $t, d['ms_per_step'], d['sketch_ms'], d['kernel_ms']['replay'], d['kernel_ms']['sort'])"
done
