# round 4, GPU call 1: the -m gpu suite, the default bench line, the pipeline-shape / option sweep, index_lambda 3..6
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04a; mkdir -p $out
(which cargo rustc; nproc; grep MemAvailable /proc/meminfo; rocm-smi --showclocks 2>/dev/null | head -20) > $out/probe.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json
timeout 600 python bench.py --steps 6 --warmup 2 --min-seconds 0.8 --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --sweep tools/r04_sweep.json > $out/bench_sweep.json 2> $out/bench_sweep.err
grep "\[sweep\]" $out/bench_sweep.err | tee $out/sweep.txt
for lam in 4 5 6; do
  SYLPH_BENCH_CTX_OPTIONS=index_lambda=$lam timeout 300 python bench.py --steps 4 --warmup 1 --min-seconds 0.5 --no-cpu-baseline --no-h2d --no-packed-leg > $out/bench_lambda$lam.json 2> $out/bench_lambda$lam.err
  python - <<PY
import json
d=json.loads(open("$out/bench_lambda$lam.json").read().strip().splitlines()[-1])
print("lambda $lam", "index_gb", d["setup"]["index_gb"], "value", d["value"], "ms", d["ms_per_sample"], "probe alone", d["roofline_profile"].get("alone_on_gpu",{}).get("avg_launch_ms"), "frac8d", d["roofline_profile"].get("alone_on_gpu",{}).get("frac"), "verify", d.get("verify",{}).get("mismatches"))
PY
done | tee $out/lambda.txt
