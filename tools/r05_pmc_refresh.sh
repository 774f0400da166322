# The PMC passes of tools/r05_profile.sh alone + profiles/seeds_traffic.json for THIS tree (bench.py quotes the traffic figures only for the
# exact kernel sources they were measured on: any later change under csrc/ — here: the bucket line size became a compile-time
# choice — needs this again), then the default bench line once more on it and a 30-second run of the sample loop.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
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
out = "gpurun_out/r05_final"
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
            v = v[-6:]                                   # the last dispatches (the first ones are settle / calibration samples)
            res.setdefault(d, {})[f"{k}.{c}"] = sum(v) / len(v)
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
print({k: len(v) for k, v in res.items()})
PY
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ $out/pmc_SQ2 $out/pmc_SQ_c3r $out/pmc_c5_FETCH_SIZE $out/pmc_c5_WRITE_SIZE $out/pmc_c5_SQ
# the filter dedup's partitioned pass: every sample behind sylph's default filter (--main-dedup-fpr), counters per (kernel, grid)
BF="python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-packed-leg --no-files-leg --main-dedup-fpr 1e-4"
KA='a10_|part_'
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "$KA" --output-format csv -d $out/pmca_$c -o s -- $BF --no-kernel-timers > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-include-regex "$KA" --output-format csv -d $out/pmca_SQ -o s -- $BF --no-kernel-timers > /dev/null 2>&1
python tools/pmc_by_kernel.py $out/pmca_FETCH_SIZE $out/pmca_WRITE_SIZE $out/pmca_SQ > $out/pmc_a10.json
rm -rf $out/pmca_FETCH_SIZE $out/pmca_WRITE_SIZE $out/pmca_SQ
# a first bench line (probes per launch, hashed k-mers) -> profiles/seeds_traffic.json for THIS tree, on the box
python bench.py --steps 4 --warmup 1 --min-seconds 0.5 --no-cpu-baseline --no-h2d --no-packed-leg --no-filter-leg --no-files-leg > $out/bench_c3_pre.json 2> $out/bench_c3_pre.err
python tools/make_r05_profile_md.py --traffic-only > $out/traffic_only.txt 2>&1; tail -2 $out/traffic_only.txt
python bench.py --steps 20 --warmup 5 > $out/bench_c3_final.json 2> $out/bench_c3_final.err
tail -c 300 $out/bench_c3_final.json; echo
python bench.py --steps 40 --warmup 5 --min-seconds 30 --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --no-filter-leg --no-files-leg --no-second-leg > $out/bench_long_run.json 2> $out/bench_long_run.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_final/bench_long_run.json").read().strip().splitlines()[-1])
print("long run:", d["timed_region_s"], "s,", d["steps"] * d["config"]["samples_per_gpu_per_step"], "samples,", d["value"], "Gbp/s, step ms", d["step_ms"], "sample interval", d["sample_interval_ms"])
PY
