# round 5, GPU call 20: the last share of a sample's blocks of reads launched behind its seeding turn's event ("reads_tail_pct")
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_gaps; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "pipeline" > $out/pytest_tail.txt 2>&1; grep -n "passed\|failed" $out/pytest_tail.txt
for i in 1 2; do for pct in 0 3 6 10 15; do
  SYLPH_BENCH_PIPE_OPTIONS=reads_tail_pct=$pct python bench.py --steps 8 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-packed-leg --no-files-leg --no-second-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('reads_tail_pct $pct', 'value', d['value'], 'default_flags', d.get('value_default_flags'), 'pipelined ms', d['ms_per_sample'], 'reads in mix', r.get('avg_launch_ms'), 'verify', d.get('verify',{}).get('mismatches'))"
done; done | tee $out/ab_tail.txt
