# the PMC passes of tools/r04_profile.sh alone (kernel sources changed outside the profiled kernels: the traffic figures are only
# quoted for the exact tree they were measured on), then the -m gpu suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04_final; mkdir -p $out
B="python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-kernel-timers --no-packed-leg --no-filter-leg --no-files-leg"
K='reads_kernel|probe_kernel|bucket_replay_kernel|hits_scatter_kernel|rows_sort_kernel|hits_count_kernel'
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "$K" --output-format csv -d $out/pmc_$c -o s -- $B > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d $out/pmc_SQ -o s -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-include-regex "$K" --output-format csv -d $out/pmc_SQ2 -o s -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-include-regex 'reads_kernel' --output-format csv -d $out/pmc_SQ_c3r -o s -- $B --workload c3r > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex 'seeds_slots_kernel' --output-format csv -d $out/pmc_c5_$c -o s -- $B --workload c5 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-include-regex 'seeds_slots_kernel' --output-format csv -d $out/pmc_c5_SQ -o s -- $B --workload c5 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, json
out = "gpurun_out/r04_final"
res = {}
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ", "pmc_SQ2", "pmc_SQ_c3r", "pmc_c5_FETCH_SIZE", "pmc_c5_WRITE_SIZE", "pmc_c5_SQ"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            k = ("probe" if "probe_kernel" in kn else "replay" if "bucket_replay" in kn else "scatter" if "hits_scatter" in kn else "rowsort" if "rows_sort" in kn
                 else "count" if "hits_count" in kn else "slots" if "seeds_slots" in kn else "reads")
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            v = v[-6:]
            res.setdefault(d, {})[f"{k}.{c}"] = sum(v) / len(v)
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
print({k: len(v) for k, v in res.items()})
PY
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ $out/pmc_SQ2 $out/pmc_SQ_c3r $out/pmc_c5_FETCH_SIZE $out/pmc_c5_WRITE_SIZE $out/pmc_c5_SQ
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -2 $out/pytest_gpu.txt
