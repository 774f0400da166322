# round-5 profile recipe (run on the GPU box through gpurun, started by tools/run_r05_profile.sh).  Order matters: the PMC passes come
# FIRST and profiles/seeds_traffic.json is written on the box from them (tools/make_r05_profile_md.py --traffic-only), so that every
# bench line written afterwards carries its `traffic` for exactly these kernel sources (VERDICT r04 #7b).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_final; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -2 $out/pytest_gpu.txt
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
# ---- the bench lines (they read the traffic figures written just now)
python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err
tail -c 300 $out/bench_c3.json
for wl in c3r c2; do
  python bench.py --workload $wl --steps 6 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d > $out/bench_$wl.json 2> $out/bench_$wl.err
done
for wl in c4 c5; do      # with the CPU leg: the WHOLE sample verified (table + every genome), as c3 has it
  python bench.py --workload $wl --steps 6 --warmup 2 --min-seconds 1.0 --no-h2d --no-files-leg > $out/bench_$wl.json 2> $out/bench_$wl.err
done
python bench.py --gpus 2 --workload small --steps 3 --warmup 1 --min-seconds 0.3 --no-cpu-baseline --no-h2d > $out/bench_small_2ranks.json 2> $out/bench_small_2ranks.err
MASTER_PORT=29581 python bench.py --gpus 2 --workload small --db-mode genome --steps 3 --warmup 1 --min-seconds 0.3 --no-cpu-baseline --no-h2d > $out/bench_small_2ranks_genome.json 2> $out/bench_small_2ranks_genome.err
MASTER_PORT=29583 python bench.py --gpus 2 --workload small --db-mode shard --steps 3 --warmup 1 --min-seconds 0.3 --no-cpu-baseline --no-h2d > $out/bench_small_2ranks_shard.json 2> $out/bench_small_2ranks_shard.err
python tools/multi_gpu_pipeline_bench.py --gpus 2 --share > $out/multi_pipeline_2replicas_one_gpu.json 2> $out/multi.err
# (a) the default (pipelined) command under the tracer: per-kernel durations as the bench line's HIP events see them
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_p -o c3 -- python bench.py --steps 4 --warmup 1 --min-seconds 0.3 --mode pipelined --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --no-filter-leg --no-files-leg > $out/bench_prof.json 2> $out/prof.err
f=$(find $out/stats_p -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 60 --anchor reads_kernel --summary-only > $out/kernels_pipelined.md
{ echo; echo "## how the seeding kernels of consecutive samples follow each other (tools/seeding_gaps.py over the same trace)"; echo; python tools/seeding_gaps.py $f --last 200 | sed 's/^/    /'; } >> $out/kernels_pipelined.md
rm -rf $out/stats_p
# (b) one sample at a time: the dispatch sequence of a sample with its gaps, every kernel alone on the GPU
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o c3 -- python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --no-filter-leg --no-files-leg > $out/bench_prof_seq.json 2>> $out/prof.err
f=$(find $out/stats -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 5 --anchor reads_kernel > $out/step_timeline.md
find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/stats
# (c) the same with sylph's default pair dedup in every leg: the filter pass's six dispatches alone on the GPU
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_f -o c3 -- $BF > $out/bench_prof_filter.json 2>> $out/prof.err
f=$(find $out/stats_f -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 5 --anchor reads_kernel > $out/step_timeline_filter.md
find $out/stats_f -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_filter.csv \;
rm -rf $out/stats_f
# a 30-second run of the sample loop (35,000 samples): no drift, no growth
python bench.py --steps 40 --warmup 2 --min-seconds 30 --no-cpu-baseline --no-h2d --no-packed-leg --no-filter-leg --no-files-leg --no-second-leg --no-verify > $out/bench_long_run.json 2> $out/bench_long_run.err
# R05_SKIP_HOST=1 (the refresh at the round's last kernel sources): the host-side legs below did not change, their outputs stay
[ -n "$R05_SKIP_HOST" ] && exit 0
python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err
d=/tmp/feed_bench
( for rep in 1 2; do echo "== plain pair, run $rep"; ( time env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/s_1.fq -2 $d/s_2.fq -d $d/out ) 2>&1 | grep -v "pgunzip\]" | head -60; done
  echo "== gz pair"; ( time env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/s_1.fq.gz -2 $d/s_2.fq.gz -d $d/out ) 2>&1 | grep -v "stretch " | head -70 ) > $out/cli_first_sample_trace.txt 2>&1
timeout 900 python tools/db_load_bench.py > $out/db_load.txt 2> $out/db_load.err
python tools/shared_kmers_check.py 2> /dev/null > $out/stress_shared_kmers.txt
python tools/deep_coverage_check.py 2> /dev/null > $out/stress_deep_coverage.txt
DEEP_CASES=extreme python tools/deep_coverage_check.py 2> /dev/null >> $out/stress_deep_coverage.txt
DEEP_DEDUP_FPR=1e-4 python tools/deep_coverage_check.py 2> /dev/null > $out/stress_deep_coverage_filter_dedup.txt
DEEP_DEDUP_FPR=1e-4 DEEP_CASES=extreme python tools/deep_coverage_check.py 2> /dev/null >> $out/stress_deep_coverage_filter_dedup.txt
python tools/deep_long_reads_check.py 2> /dev/null > $out/stress_deep_long_reads.txt
tail -n 3 $out/feed.txt; tail -n 3 $out/stress_deep_coverage_filter_dedup.txt
