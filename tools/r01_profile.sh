cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_c3.json 2> gpurun_out/final/bench_c3.err
tail -c 600 gpurun_out/final/bench_c3.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -o c3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_prof.json 2> gpurun_out/final/prof.err
f=$(find gpurun_out/final/stats -name '*kernel_trace.csv' | head -1)
python tools/kernel_trace_summary.py $f --tail-ms 22 > gpurun_out/final/kernel_summary.md
find gpurun_out/final/stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/final/kernel_stats.csv \;
rm -f $f
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'seeds_slots_kernel' --output-format csv -d gpurun_out/final/pmc_fetch -o s -- python tools/trace_run.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex 'seeds_slots_kernel' --output-format csv -d gpurun_out/final/pmc_write -o s -- python tools/trace_run.py > /dev/null 2>&1
head -12 gpurun_out/final/kernel_summary.md
find gpurun_out/final -name '*counter_collection.csv' | xargs -I{} sh -c 'echo {}; cat {} | cut -d, -f 9-20 | head -5'
