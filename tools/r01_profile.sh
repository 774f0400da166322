# round-1 profile recipe: bench (with cpu baseline), rocprofv3 kernel trace + stats, PMC passes for the dominant kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/final; mkdir -p $out
python bench.py > $out/bench_c3.json 2> $out/bench_c3.err
tail -c 400 $out/bench_c3.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o c3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_prof.json 2> $out/prof.err
f=$(find $out/stats -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 5 --anchor reads_kernel > $out/step_timeline.md
find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -f $f
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'reads_kernel' --output-format csv -d $out/pmc_fetch -o s -- python tools/trace_run.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex 'reads_kernel' --output-format csv -d $out/pmc_write -o s -- python tools/trace_run.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex 'reads_kernel' --output-format csv -d $out/pmc_sq -o s -- python tools/trace_run.py > /dev/null 2>&1
head -60 $out/step_timeline.md
