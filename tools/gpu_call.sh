#!/usr/bin/env bash
# ONE recipe for the GPU calls of a round (replaces the numbered r04_call*/r05_call*/r06_call* scratch scripts).  Run on the
# GPU box through `gpurun -- 'bash tools/gpu_call.sh <what> [args]'`; everything lands under gpurun_out/<what>/ (scratch: copy what is to
# be judged into profiles/).
#   tests [pytest -k expr]     the -m gpu suite (or a selection of it)
#   inflate                    tests/test_gpu_inflate.py with both LDS ring sizes + tools/inflate_bench.py on bench.py's file shape and an
#                              Illumina-like one (gzip -1 / -6, BGZF), per-wave statistics on
#   gz-trace [Gbp]             tools/gz_e2e_trace.py: `sylph-hip sketch` on a plain / gzip pair with the feed's and the library's phase traces
#   bench [bench.py args]      the default bench line -> gpurun_out/bench/bench.json (+ the rates summary on stdout)
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
  ab-reads) bash tools/r06_ab_reads.sh ;;
  inflate-profile) bash tools/r06_inflate_profile.sh ;;
  round-profile) bash tools/r05_profile.sh ;;
  *) echo "unknown: $what"; exit 2 ;;
esac
