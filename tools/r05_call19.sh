# round 5, GPU call 19: the wait for the previous sample's seeding kernel placed immediately in front of this sample's seeding kernel
# (common.h SeedTurn) against in front of the whole push (pipeline option serialize_outside=1: rounds 1-4) — tests, gaps, bench A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_gaps; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -m gpu -x -q > $out/pytest_turn.txt 2>&1; grep -n "passed\|failed" $out/pytest_turn.txt
B="python bench.py --steps 4 --warmup 1 --min-seconds 0.4 --mode pipelined --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --no-filter-leg --no-files-leg"
for t in inside outside; do
  o=0; [ $t = outside ] && o=1
  SYLPH_BENCH_PIPE_OPTIONS=serialize_outside=$o rocprofv3 --kernel-trace --output-format csv -d $out/tr_$t -o c3 -- $B > $out/bench_$t.json 2> $out/err_$t.txt
  f=$(find $out/tr_$t -name '*kernel_trace.csv' | head -1)
  echo "== $t (under the tracer): $(python -c "import json;d=json.loads(open('$out/bench_$t.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_sample'])")" | tee -a $out/gaps_turn.txt
  python tools/seeding_gaps.py $f --last 300 | tee -a $out/gaps_turn.txt
  rm -rf $out/tr_$t
done
for i in 1 2 3; do for o in 0 1; do
  SYLPH_BENCH_PIPE_OPTIONS=serialize_outside=$o python bench.py --steps 8 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-packed-leg --no-files-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('pipelined',{}); s=d.get('one_step_at_a_time',{}); r=d['roofline']
print('serialize_outside $o', 'value', d['value'], 'default_flags', d.get('value_default_flags'), 'pipelined ms', d['ms_per_sample'], 'sequential ms', s.get('ms_per_sample'), 'reads in mix', r.get('avg_launch_ms'), 'verify', d.get('verify',{}).get('mismatches'))"
done; done | tee $out/ab_seed_turn.txt
