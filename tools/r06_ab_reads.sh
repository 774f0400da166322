# The four builds of tools/r06_build_reads_variants.sh alternated on ONE box: the default C3 bench, both modes (ms per sample; the seeding
# kernel's ms alone on the GPU from the one-at-a-time leg), then SQ_INSTS_VALU of reads_kernel per build (counters only).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_ab_reads; mkdir -p $out
cp sylph_amd/libsylph_hip.so sylph_amd/libsylph_hip.so.keep
for i in 1 2 3; do
  for v in v0 v1 v2 v3; do
    cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
    python bench.py --steps 6 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-verify --no-files-leg --no-packed-leg --no-filter-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('pipelined',{}); s=d.get('one_step_at_a_time',{})
print('$v', 'pipelined Gbp/s', p.get('value'), 'ms/sample', p.get('ms_per_sample'), '| one at a time ms/sample', s.get('ms_per_sample'), 'seeds ms', (s.get('kernel_ms') or {}).get('seeds'))"
  done
done | tee $out/ab.txt
for v in v0 v1 v2 v3; do
  cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-include-regex 'reads_kernel' --output-format csv -d $out/pmc_$v -o s -- python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-files-leg --no-packed-leg --no-filter-leg --no-kernel-timers > /dev/null 2>&1
  echo "$v $(python tools/pmc_by_kernel.py $out/pmc_$v --last 3 | tr -d '\n ' | cut -c1-400)" | tee -a $out/pmc.txt
  rm -rf $out/pmc_$v
done
cp sylph_amd/libsylph_hip.so.keep sylph_amd/libsylph_hip.so
