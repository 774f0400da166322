# round 5, GPU call 15: the read kernel's high-word candidate test (ctx option "reads_hash" = 2 / SYLPH_HIP_HASH_VARIANT=2):
# its parity test, the whole -m gpu suite with it as the process default, then the bench alternated between spellings 1 and 2
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_hv2; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hash_spellings" > $out/pytest_hv_test.txt 2>&1; tail -n 5 $out/pytest_hv_test.txt
SYLPH_HIP_HASH_VARIANT=2 timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_hv2.txt 2>&1; grep -n "passed\|failed" $out/pytest_gpu_hv2.txt
for i in 1 2 3; do for hv in 1 2; do
  SYLPH_HIP_HASH_VARIANT=$hv python bench.py --steps 8 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-packed-leg --no-files-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('pipelined',{}); s=d.get('one_step_at_a_time',{}); r=d['roofline']; a=r.get('alone_on_gpu',{})
print('hv $hv', 'value', d['value'], 'default_flags', d.get('value_default_flags'), 'pipelined ms', p.get('ms_per_sample'), 'sequential ms', s.get('ms_per_sample'), 'reads alone ms', a.get('avg_launch_ms'), 'in mix', r.get('avg_launch_ms'), 'verify', d.get('verify',{}).get('mismatches'))"
done; done | tee $out/ab_hv.txt
