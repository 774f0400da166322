for p in 8 4 2 1; do
SYLPH_HIP_BUCKET_BITS_MAX=31 SYLPH_HIP_POSTINGS_PER_BUCKET=$p python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ppb', $p, d['ms_per_step'], d['profile_ms'], d['kernel_ms']['probe'])"
done
