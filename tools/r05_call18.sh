# round 5, GPU call 18: how the seeding kernels of consecutive samples follow each other in the pipelined mode (tools/seeding_gaps.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_gaps; mkdir -p $out
B="python bench.py --steps 4 --warmup 1 --min-seconds 0.4 --mode pipelined --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --no-filter-leg --no-files-leg"
for t in timers notimers; do
  x=""; [ $t = notimers ] && x="--no-kernel-timers"
  rocprofv3 --kernel-trace --output-format csv -d $out/tr_$t -o c3 -- $B $x > $out/bench_$t.json 2> $out/err_$t.txt
  f=$(find $out/tr_$t -name '*kernel_trace.csv' | head -1)
  echo "== $t: $(python -c "import json;d=json.loads(open('$out/bench_$t.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_sample'])")" | tee -a $out/gaps.txt
  python tools/seeding_gaps.py $f --last 300 | tee -a $out/gaps.txt
  rm -rf $out/tr_$t
done
