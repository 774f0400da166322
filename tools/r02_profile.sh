# round-2 profile recipe (run on the GPU box through gpurun): bench lines of every single-GPU configuration, rocprofv3 kernel
# trace + stats of the C3 bench, PMC passes (each counter group in its own run, counters only) for the two kernels with a
# roofline object, the instruction-rate microbenchmark, the host-feed measurement.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r02_final; mkdir -p $out
# (the box has no .git: tools/run_r02_profile.sh records HEAD and the clean state of the tree when it starts this script)
python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err
tail -c 300 $out/bench_c3.json
for wl in c3r c2 c5 c4; do
  python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-h2d > $out/bench_$wl.json 2> $out/bench_$wl.err
done
# (a) the default (pipelined) command under the tracer: per-kernel averages to set beside the line's HIP-event numbers
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_p -o c3 -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2d --no-verify --no-sequential-leg > $out/bench_prof.json 2> $out/prof.err
f=$(find $out/stats_p -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 10 --anchor reads_kernel --summary-only > $out/kernels_pipelined.md
rm -f $f
# (b) one step at a time: the dispatch sequence of a step with its gaps, every kernel alone on the GPU
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o c3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-h2d --no-verify --pipeline-depth 1 > $out/bench_prof_seq.json 2>> $out/prof.err
f=$(find $out/stats -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 5 --anchor reads_kernel > $out/step_timeline.md
find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -f $f
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-verify --no-kernel-timers --pipeline-depth 1"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex 'reads_kernel|probe_kernel' --output-format csv -d $out/pmc_$c -o s -- $B > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex 'reads_kernel|probe_kernel' --output-format csv -d $out/pmc_SQ -o s -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-include-regex 'reads_kernel|probe_kernel' --output-format csv -d $out/pmc_SQ2 -o s -- $B > /dev/null 2>&1
# the same counters for the ragged workload (lanes of a wavefront walk reads of different lengths)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-include-regex 'reads_kernel' --output-format csv -d $out/pmc_SQ_c3r -o s -- $B --workload c3r > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o /tmp/valu_rates 2> /dev/null && /tmp/valu_rates > $out/valu_rates.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/atomic_rates.hip -o /tmp/atomic_rates 2> /dev/null && /tmp/atomic_rates > $out/atomic_rates.txt 2>&1
python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err
python - <<'PY'
import csv, glob, collections, json
out = "gpurun_out/r02_final"
res = {}
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ", "pmc_SQ2", "pmc_SQ_c3r"):
    for f in glob.glob(f"{out}/{d}/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = "probe" if "probe_kernel" in r["Kernel_Name"] else "reads"
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            v = v[-3:]                                   # the timed steps (the first dispatches are settle / warm-up steps)
            res.setdefault(d, {})[f"{k}.{c}"] = sum(v) / len(v)
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
cat $out/valu_rates.txt; cat $out/feed.txt
