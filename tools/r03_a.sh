# round 3, first GPU call: the whole GPU suite, then the new bench (calibrated mode, >= 2 s timed region), then fixed modes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err; tail -c 600 $out/bench_c3.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03a/bench_c3.json").read().strip().splitlines()[-1])
for k in ("value","mode","ms_per_sample","timed_region_s","step_ms","sample_interval_ms","calibration","kernel_ms","probe_batch_mean","sketch_ms","profile_ms"): print(k, d.get(k))
for k in ("pipelined","one_step_at_a_time"):
    if k in d: print(k, {x:d[k].get(x) for x in ("value","ms_per_sample","step_ms","sample_interval_ms","kernel_ms","sketch_ms","profile_ms","probe_batch_mean")})
print("roofline", d.get("roofline")); print("roofline_profile", d.get("roofline_profile")); print("verify", d.get("verify"))
PY
for w in 1 2 3; do for dp in 3 4 6; do
python bench.py --steps 10 --warmup 2 --mode pipelined --sketch-workers $w --pipeline-depth $dp --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-verify --no-second-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers',$w,'depth',$dp,'value',d['value'],'ms/sample',d['ms_per_sample'],d['sample_interval_ms'],d['kernel_ms'])"
done; done | tee $out/sweep.txt
