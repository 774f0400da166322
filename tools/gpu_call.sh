#!/usr/bin/env bash
# ONE recipe for the GPU calls of a round (replaces the numbered r04_call*/r05_call*/r06_call* scratch scripts).  Run on the
# GPU box through `gpurun -- 'bash tools/gpu_call.sh <what> [args]'`; everything lands under gpurun_out/<what>/ (scratch: copy what is to
# be judged into profiles/).
#   tests [pytest -k expr]     the -m gpu suite (or a selection of it)
#   inflate                    tests/test_gpu_inflate.py with both LDS ring sizes + tools/inflate_bench.py on bench.py's file shape and an
#                              Illumina-like one (gzip -1 / -6, BGZF), per-wave statistics on
#   gz-trace [Gbp]             tools/gz_e2e_trace.py: `sylph-hip sketch` on a plain / gzip pair with the feed's and the library's phase traces
#   bench [bench.py args]      the default bench line -> gpurun_out/bench/bench.json (+ the rates summary on stdout)
#   a10                        filter-dedup tests + the default-flag bench leg with one and with two partition levels
#   a10-profile [--no-pmc]     rocprofv3 kernel trace (+ FETCH_SIZE / WRITE_SIZE) of the filter pass's kernels, one sample at a time
#   ab-env cfg...              alternate environment configurations on the pipelined default-flag rate (AB_ROUNDS, AB_SECONDS, AB_FPR)
#   ab-reads                   tools/r06_ab_reads.sh (build the variants first, here: tools/r06_build_reads_variants.sh)
#   inflate-profile            tools/r06_inflate_profile.sh (rocprofv3 kernel trace + counters of the inflate kernels)
#   round-profile              tools/r05_profile.sh (the whole-pipeline rocprofv3 recipe of round 5, unchanged)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
what=${1:-tests}; shift || true
out=gpurun_out/$what; mkdir -p "$out"
case "$what" in
  tests)
    if [ $# -gt 0 ]; then timeout 1800 python -m pytest tests -m gpu -x -q -k "$*" > "$out/pytest.txt" 2>&1; else timeout 1800 python -m pytest tests -m gpu -x -q > "$out/pytest.txt" 2>&1; fi
    tail -5 "$out/pytest.txt" ;;
  inflate)
    timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > "$out/pytest.txt" 2>&1; tail -3 "$out/pytest.txt"
    SYLPH_HIP_INFLATE_SMALL_RING=1 timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > "$out/pytest_small_ring.txt" 2>&1; tail -2 "$out/pytest_small_ring.txt"
    export SYLPH_HIP_INFLATE_STATS=1
    timeout 600 python tools/inflate_bench.py --mbp 500 --levels 1,6 --qual const --reps 3 --check 2>&1 | tee "$out/bench_const.txt" | cut -c1-300
    timeout 600 python tools/inflate_bench.py --mbp 120 --levels 6 --qual binned --bgzf --reps 3 --check 2>&1 | tee "$out/bench_binned.txt" | cut -c1-300 ;;
  gz-trace)
    SYLPH_HIP_INFLATE_STATS=1 timeout 900 python tools/gz_e2e_trace.py "$@" 2>&1 | grep -v 'pool miss' | tee "$out/gz_trace.txt" | grep -v 'engine:' | cut -c1-200 ;;
  bench)
    timeout 1500 python bench.py "$@" > "$out/bench.json" 2> "$out/bench.err"; tail -c 400 "$out/bench.err"
    python -c "import json; o = json.load(open('$out/bench.json')); print(json.dumps({k: o.get(k) for k in ('value', 'value_default_flags', 'ms_per_step', 'rates_gbp_per_s')}, indent=1))" ;;
  a10)
    # the filter dedup's pass with one partition level (default) and with round 5's two (SYLPH_HIP_A10_LEVELS=2): tests, then the default-flag leg
    for lv in 1 2; do
      SYLPH_HIP_A10_LEVELS=$lv timeout 900 python -m pytest tests -m gpu -x -q -k "filter or cuckoo or a10" > "$out/pytest_levels$lv.txt" 2>&1; tail -3 "$out/pytest_levels$lv.txt"
    done
    # configurations levels:split:pad — partition levels (1 | 2), workgroups per range of the one-level pass, extra LDS bytes per workgroup (footprint A/B)
    for cfg in ${A10_CONFIGS:-1:2:0 2:2:0 1:1:0 1:4:0 1:2:24576 1:2:0 2:2:0}; do
      lv=${cfg%%:*}; rest=${cfg#*:}; split=${rest%%:*}; pad=${rest##*:}; tpb=${split}_$pad
      SYLPH_HIP_A10_LEVELS=$lv SYLPH_HIP_A10_RANGE_SPLIT=$split SYLPH_HIP_A10_RANGE_LDS_PAD=$pad timeout 600 python bench.py --no-files-leg --no-packed-leg --no-h2d --no-cpu-baseline --all-kernel-timers "$@" > "$out/bench_levels${lv}_$tpb.json" 2> "$out/bench_levels${lv}_$tpb.err"
      python - "$out/bench_levels${lv}_$tpb.json" $cfg <<'PY'
import json, sys
o = json.load(open(sys.argv[1])); f = o.get("default_pair_dedup", {})
print("levels:split:pad", sys.argv[2], "value", o.get("value"), "default flags", o.get("value_default_flags"), "ms/sample pipelined", f.get("pipelined", {}).get("ms_per_sample"),
      "one at a time", f.get("one_step_at_a_time", {}).get("ms_per_sample"), "a10 alone", f.get("one_step_at_a_time", {}).get("kernel_ms", {}).get("a10"))
PY
    done ;;
  a10-profile)
    # per-kernel durations (rocprofv3 kernel trace) and HBM counters of the filter pass: every sample behind sylph's default filter, one at a time
    BF="python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-packed-leg --no-files-leg --no-verify --main-dedup-fpr 1e-4 --no-kernel-timers"
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- $BF > /dev/null 2> "$out/trace.err"
    python - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/s_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(sys.argv[1] + "/kernel_stats.txt", "w") as o:
    for r in rows:
        line = f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f} pct {r["Percentage"]}'
        o.write(line + "\n")
        if "a10_" in r["Name"] or "part_" in r["Name"] or "reads_kernel" in r["Name"]: print(line)
PY
    if [ "${1:-}" != "--no-pmc" ]; then
      for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-include-regex 'a10_|part_' --output-format csv -d $out/pmca_$c -o s -- $BF > /dev/null 2>&1; done
      rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-include-regex 'a10_|part_|reads_kernel' --output-format csv -d $out/pmca_SQ -o s -- $BF > /dev/null 2>&1
      python tools/pmc_by_kernel.py $out/pmca_FETCH_SIZE $out/pmca_WRITE_SIZE $out/pmca_SQ > $out/pmc_a10.json; rm -rf $out/pmca_FETCH_SIZE $out/pmca_WRITE_SIZE $out/pmca_SQ $out/trace
      python -c "import json; d = json.load(open('$out/pmc_a10.json')); print(json.dumps(d, indent=1)[:3000])"
    fi ;;
  ab-env)
    # A/B of environment knobs on the pipelined default-flag rate (every session behind sylph's default filter): AB_ROUNDS rounds over the
    # configurations given as arguments ("-" = no variable; "A=1,B=2" = two), alternating, one short bench process each
    BF="python bench.py --main-dedup-fpr ${AB_FPR:-1e-4} --no-filter-leg --no-second-leg --no-verify --no-files-leg --no-packed-leg --no-h2d --no-cpu-baseline --min-seconds ${AB_SECONDS:-3} ${AB_EXTRA:-}"
    : > "$out/ab.txt"
    for r in $(seq 1 ${AB_ROUNDS:-4}); do
      for cfg in "$@"; do
        envs=""; [ "$cfg" != "-" ] && envs=$(echo "$cfg" | tr ',' ' ')
        env $envs timeout 300 $BF > "$out/one.json" 2> "$out/one.err"
        python -c "import json; o = json.load(open('$out/one.json')); print('round $r', '$cfg', 'Gbp/s', o['value'], 'interval_ms', o.get('sample_interval_ms'))" | tee -a "$out/ab.txt"
      done
    done
    python - "$out/ab.txt" <<'PY'
import sys, collections, statistics
v = collections.defaultdict(list)
for l in open(sys.argv[1]):
    t = l.split()
    v[t[2]].append(float(t[4]))
for k, x in v.items(): print(f"{k:60s} n {len(x)} median {statistics.median(x):8.1f} min {min(x):8.1f} max {max(x):8.1f} Gbp/s")
PY
    ;;
  ab-reads) bash tools/r06_ab_reads.sh ;;
  inflate-profile) bash tools/r06_inflate_profile.sh ;;
  round-profile) bash tools/r05_profile.sh ;;
  *) echo "unknown: $what"; exit 2 ;;
esac
