# round-4 profile recipe (run on the GPU box through gpurun, started by tools/run_r04_profile.sh): bench lines of every single-GPU
# configuration, rocprofv3 kernel trace + stats of the C3 bench (pipelined and one sample at a time), PMC passes (each counter
# group in its own run, counters only) for the kernels with a roofline object, the feed measurement, the stress checks.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04_final; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -2 $out/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err
tail -c 300 $out/bench_c3.json
for wl in c3r c2 c4 c5; do
  python bench.py --workload $wl --steps 6 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d > $out/bench_$wl.json 2> $out/bench_$wl.err
done
python bench.py --gpus 2 --workload small --steps 3 --warmup 1 --min-seconds 0.3 --no-cpu-baseline --no-h2d > $out/bench_small_2ranks.json 2> $out/bench_small_2ranks.err
MASTER_PORT=29581 python bench.py --gpus 2 --workload small --db-mode genome --steps 3 --warmup 1 --min-seconds 0.3 --no-cpu-baseline --no-h2d > $out/bench_small_2ranks_genome.json 2> $out/bench_small_2ranks_genome.err
# (a) the default (pipelined) command under the tracer: per-kernel durations as the bench line's HIP events see them
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_p -o c3 -- python bench.py --steps 4 --warmup 1 --min-seconds 0.3 --mode pipelined --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --no-filter-leg --no-files-leg > $out/bench_prof.json 2> $out/prof.err
f=$(find $out/stats_p -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 60 --anchor reads_kernel --summary-only > $out/kernels_pipelined.md
rm -rf $out/stats_p
# (b) one sample at a time: the dispatch sequence of a sample with its gaps, every kernel alone on the GPU
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o c3 -- python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --no-filter-leg --no-files-leg > $out/bench_prof_seq.json 2>> $out/prof.err
f=$(find $out/stats -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 5 --anchor reads_kernel > $out/step_timeline.md
find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/stats
# (c) the same with sylph's default pair dedup (the cuckoo filter, csrc/a10.hip): what the a10 kernels cost alone on the GPU
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_f -o c3 -- python bench.py --steps 2 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --no-files-leg > $out/bench_prof_filter.json 2>> $out/prof.err
find $out/stats_f -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_filter.csv \;
rm -rf $out/stats_f
B="python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-kernel-timers --no-packed-leg --no-filter-leg --no-files-leg"
K='reads_kernel|probe_kernel|bucket_replay_kernel|hits_scatter_kernel|rows_sort_kernel|hits_count_kernel'
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "$K" --output-format csv -d $out/pmc_$c -o s -- $B > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d $out/pmc_SQ -o s -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-include-regex "$K" --output-format csv -d $out/pmc_SQ2 -o s -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-include-regex 'reads_kernel' --output-format csv -d $out/pmc_SQ_c3r -o s -- $B --workload c3r > /dev/null 2>&1
# the long-read position kernel (c5): traffic + instruction counters of its own
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
            v = v[-6:]                                   # the last dispatches (the first ones are settle / calibration samples)
            res.setdefault(d, {})[f"{k}.{c}"] = sum(v) / len(v)
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:2500])
PY
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ $out/pmc_SQ2 $out/pmc_SQ_c3r $out/pmc_c5_FETCH_SIZE $out/pmc_c5_WRITE_SIZE $out/pmc_c5_SQ
python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err
timeout 900 python tools/db_load_bench.py > $out/db_load.txt 2> $out/db_load.err
python tools/shared_kmers_check.py 2> /dev/null > $out/stress_shared_kmers.txt
python tools/deep_coverage_check.py 2> /dev/null > $out/stress_deep_coverage.txt
DEEP_CASES=extreme python tools/deep_coverage_check.py 2> /dev/null >> $out/stress_deep_coverage.txt
python tools/deep_long_reads_check.py 2> /dev/null > $out/stress_deep_long_reads.txt
tail -3 $out/feed.txt $out/stress_shared_kmers.txt $out/stress_deep_coverage.txt
