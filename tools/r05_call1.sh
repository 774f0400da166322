# round 5, GPU call 1: the filter dedup's partitioned pass (csrc/a10.hip) — its tests, the whole -m gpu suite, the default bench line,
# and the same line with the round-4 walk forced (SYLPH_HIP_A10=walk) for the A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05a; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "filter" > $out/pytest_filter.txt 2>&1; tail -15 $out/pytest_filter.txt
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.json
SYLPH_HIP_A10=walk timeout 600 python bench.py --steps 6 --warmup 2 --min-seconds 0.8 --no-cpu-baseline --no-h2d --no-verify --no-packed-leg > $out/bench_walk.json 2> $out/bench_walk.err
python - <<PY
import json
for f in ("bench_default", "bench_walk"):
    try:
        d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
        fl = d.get("default_pair_dedup", {})
        print(f, "value", d["value"], "ms/sample", d["ms_per_sample"], "filter pipelined", fl.get("pipelined", {}).get("value"), "one at a time", fl.get("one_step_at_a_time", {}).get("value"),
              "verify", fl.get("verify", {}).get("table_equal"), "a10 ms", fl.get("one_step_at_a_time", {}).get("kernel_ms", {}).get("a10"), "err", fl.get("error"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
