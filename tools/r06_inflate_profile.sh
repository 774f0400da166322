# Round 6: the device inflate alone under rocprofv3 — kernel trace + stats of tools/inflate_bench.py on bench.py's 0.5 Gbp mate file
# (gzip -1, ~198 MB; the file of the from-files leg) and on an Illumina-like file (gzip -6), then counter passes (one group per run, counters
# only) for decode_kernel: HBM bytes (FETCH_SIZE / WRITE_SIZE) and the SQ instruction mix.  Summaries -> gpurun_out/r06_inflate/ ->
# profiles/r06_inflate_*.  Started with: gpurun -- 'bash tools/r06_inflate_profile.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_inflate; mkdir -p $out
git rev-parse HEAD > $out/head.txt 2>/dev/null || cat profiles/.head_local > $out/head.txt 2>/dev/null
B="python tools/inflate_bench.py --mbp 500 --levels 1 --qual const --reps 3"
B2="python tools/inflate_bench.py --mbp 120 --levels 6 --qual binned --bgzf --reps 3"
SYLPH_HIP_INFLATE_STATS=1 $B --check > $out/bench_const.txt 2>&1
SYLPH_HIP_INFLATE_STATS=1 $B2 --check > $out/bench_binned.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > /dev/null 2>&1
python tools/kernel_trace_summary.py $(find $out/trace -name '*kernel_trace.csv' | head -1) > $out/kernel_stats_const.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace2 -o t -- $B2 > /dev/null 2>&1
python tools/kernel_trace_summary.py $(find $out/trace2 -name '*kernel_trace.csv' | head -1) > $out/kernel_stats_binned.txt 2>&1
K='decode_kernel'
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "$K" --output-format csv -d $out/pmc_$c -o s -- $B > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d $out/pmc_SQ -o s -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-include-regex "$K" --output-format csv -d $out/pmc_SQ2 -o s -- $B > /dev/null 2>&1
python tools/pmc_by_kernel.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ $out/pmc_SQ2 --last 3 > $out/pmc_decode.json 2>&1
cp $(find $out/trace -name '*kernel_stats.csv' | head -1) $out/rocprof_kernel_stats_const.csv 2>/dev/null
cp $(find $out/trace2 -name '*kernel_stats.csv' | head -1) $out/rocprof_kernel_stats_binned.csv 2>/dev/null
rm -rf $out/trace $out/trace2
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ $out/pmc_SQ2
ls -la $out; cat $out/kernel_stats_const.txt | head -30; cat $out/pmc_decode.json | head -60; cat $out/bench_const.txt | cut -c1-300
