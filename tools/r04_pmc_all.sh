# VALU / SALU / LDS instruction counts of EVERY kernel of one sample (sequential mode), one counter group per run
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04h; mkdir -p $out
B="python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-kernel-timers --no-packed-leg"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-include-regex 'sylph::' --output-format csv -d $out/pmc_all -o s -- $B > /dev/null 2> $out/pmc.err
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04h/pmc_all/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in agg.items():
    n = len(c["SQ_INSTS_VALU"])
    last = lambda name: sum(c[name][-4:]) / max(1, len(c[name][-4:]))
    rows.append((last("SQ_INSTS_VALU"), k, n, last("SQ_INSTS_SALU"), last("SQ_INSTS_LDS"), last("SQ_WAVES"), last("GRBM_GUI_ACTIVE") / 8))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("| kernel | launches | VALU wave-instr | share | SALU | LDS | waves | cycles/XCD |\n|---|---|---|---|---|---|---|---|")
for v, k, n, s, l, w, cy in rows:
    print(f"| `{k}` | {n} | {v:.4g} | {100 * v / tot:.1f} % | {s:.3g} | {l:.3g} | {w:.3g} | {cy:.4g} |")
PY
rm -rf $out/pmc_all
