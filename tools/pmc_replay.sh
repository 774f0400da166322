cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/pmc2; rm -rf $out; mkdir -p $out
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-verify --no-kernel-timers --pipeline-depth 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex 'bucket_replay_kernel' --output-format csv -d $out/sq -o s -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-include-regex 'bucket_replay_kernel' --output-format csv -d $out/sq2 -o s -- $B > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("sq","sq2"):
    for f in glob.glob(f"gpurun_out/pmc2/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            v = v[-3:]
            print(d, c, "%.4e" % (sum(v)/len(v)), len(agg[c]))
PY
find $out -name '*.csv' -size +200k -delete
