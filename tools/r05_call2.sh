# round 5, GPU call 2: the -m gpu suite after the deferred-count fix; the filter dedup's kernels alone on the GPU (trace of a
# sequential run with the filter on in every leg) and their traffic / instruction counters (one counter group per run)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05b; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
B="python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-packed-leg --no-files-leg --main-dedup-fpr 1e-4"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_f -o c3 -- $B > $out/bench_prof_filter.json 2> $out/prof.err
f=$(find $out/stats_f -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 5 --anchor reads_kernel > $out/step_timeline_filter.md
find $out/stats_f -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_filter.csv \;
rm -rf $out/stats_f
head -40 $out/step_timeline_filter.md
K='a10_|part_'
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "$K" --output-format csv -d $out/pmc_$c -o s -- $B --no-kernel-timers > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d $out/pmc_SQ -o s -- $B --no-kernel-timers > /dev/null 2>&1
python tools/pmc_by_kernel.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ > $out/pmc_a10.json
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ
head -c 3000 $out/pmc_a10.json
