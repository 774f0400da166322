# postings per 64-byte index line aimed for (ctx option index_lambda): probe time and index size at C3
for p in 8 4 3 2 1; do
SYLPH_BENCH_CTX_OPTIONS=index_lambda=$p python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-h2d --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lambda', $p, d['ms_per_step'], d['profile_ms'], d['kernel_ms']['probe'], d['setup']['index_gb'])"
done
