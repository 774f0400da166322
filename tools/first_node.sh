#!/usr/bin/env bash
# First run on a node with more than one MI355X (VERDICT r05 #6c): the four library modes + the CLI at N = 1, 2, 4, 8 GPUs (as many of
# them as the node has), one JSON line per (mode, N) into gpurun_out/first_node/SCALE_<mode>.jsonl — the shape of the driver's SCALE files
# ({"n_gpus", "value", "unit", "ms_per_step", ...}: bench.py's own line, untouched).  No efficiency is computed here: the curves are the
# reader's.  What each mode exercises for the first time on real hardware:
#   replicate   bench.py --db-mode replicate   one rank per GPU, every rank a whole index, no data-path collective (the default)
#   shard       bench.py --db-mode shard       k-mer-range shards, RCCL all-to-all of table slices + all-to-all of hits (csrc/shard.hip)
#   genome      bench.py --db-mode genome      north_star's cut: whole genomes per rank, sample tables all-gathered, hits to the owner by all-to-all
#   genome-allgather   + --shard-reduce allgather: the hits by ONE all-gather of padded blocks (north_star's literal wording) — the A/B of the two reductions
#   router      tools/multi_gpu_pipeline_bench.py   ONE process, index copied device to device over xGMI (hipMemcpyPeer), one sample loop
#   cli         sylph-hip sketch --gpus N / profile --gpus N on files (tools/feed_bench.py's shapes)
# Usage: bash tools/first_node.sh [steps] [warmup]          (from the repository root; ~10 minutes on 8 GPUs)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=${TMPDIR:-/tmp} HSA_ENABLE_IPC_MODE_LEGACY=0
STEPS=${1:-6}; WARMUP=${2:-2}
OUT=gpurun_out/first_node; mkdir -p "$OUT"
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
echo "[first_node] $NGPU GPU(s) visible; RCCL $(python -c 'import torch; print(torch.cuda.nccl.version())' 2>/dev/null)" | tee "$OUT/node.txt"
rocm-smi --showtopo > "$OUT/topology.txt" 2>&1 || true
python -m pytest tests -m gpu -x -q -k "dist or rccl or replicas or several_gpus" > "$OUT/pytest_multi.txt" 2>&1; tail -2 "$OUT/pytest_multi.txt"
PORT=29611
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || continue
  for MODE in replicate shard genome genome-allgather; do
    [ "$N" -eq 1 ] && [ "$MODE" != replicate ] && continue
    PORT=$((PORT + 1))
    if [ "$N" -eq 1 ]; then CMD="python bench.py --gpus 1"; else CMD="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N"; fi
    echo "[first_node] $MODE, $N GPU(s)"
    RED=alltoall; DBM=$MODE; [ "$MODE" = genome-allgather ] && { RED=allgather; DBM=genome; }
    timeout 1500 $CMD --steps "$STEPS" --warmup "$WARMUP" --db-mode "$DBM" --shard-reduce "$RED" --no-cpu-baseline --no-files-leg --no-h2d --no-packed-leg 2> "$OUT/bench_${MODE}_$N.err" | tail -1 >> "$OUT/SCALE_$MODE.jsonl" \
      || echo "{\"n_gpus\": $N, \"mode\": \"$MODE\", \"error\": \"see bench_${MODE}_$N.err\"}" >> "$OUT/SCALE_$MODE.jsonl"
  done
  echo "[first_node] router (one process), $N GPU(s)"
  timeout 900 python tools/multi_gpu_pipeline_bench.py --gpus "$N" 2> "$OUT/router_$N.err" | tail -1 >> "$OUT/SCALE_router.jsonl"
done
# the CLI over the node: four plain + four gzip samples, 1 and all GPUs
python - "$NGPU" <<'PY' 2> "$OUT/cli.err" | tee "$OUT/SCALE_cli.jsonl" || true
import json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, "tools")
import feed_bench as FB
ngpu = int(sys.argv[1])
d = "/tmp/first_node_cli"; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
n_pairs, L = 1_000_000, 150
for m in (1, 2):
    FB.write_fastq(f"{d}/s_{m}.fq", rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n_pairs * L), L)
subprocess.run(["gzip", "-1", "-k", "-f", f"{d}/s_1.fq", f"{d}/s_2.fq"], check=True)
S = 16                                                     # samples per command
for i in range(S):
    for m in (1, 2):
        for ext in ("fq", "fq.gz"):
            dst = f"{d}/p{i}_{m}.{ext}"
            if os.path.lexists(dst):
                os.remove(dst)
            os.symlink(f"{d}/s_{m}.{ext}", dst)
gbp = S * 2 * n_pairs * L / 1e9
for ext in ("fq", "fq.gz"):
    for n in (1, 2, 4, 8):
        if n > ngpu:
            continue
        t = time.perf_counter()
        p = subprocess.run(["sylph_amd/sylph-hip", "sketch", "-1", *[f"{d}/p{i}_1.{ext}" for i in range(S)], "-2", *[f"{d}/p{i}_2.{ext}" for i in range(S)],
                            "-d", f"{d}/out", "-t", str(2 * n), "--gpus", str(n)], capture_output=True, text=True)
        dt = time.perf_counter() - t
        print(json.dumps({"mode": "cli sketch " + ext, "n_gpus": n, "value": round(gbp / dt, 3), "unit": "Gbp/s", "command_seconds": round(dt, 3),
                          "samples": S, "rc": p.returncode}), flush=True)
PY
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/first_node/SCALE_*.jsonl")):
    for ln in open(f):
        try:
            o = json.loads(ln)
        except Exception:
            continue
        print(f.split("SCALE_")[1][:-6], o.get("n_gpus"), o.get("value"), o.get("unit"), o.get("ms_per_step"), o.get("error", ""))
PY
