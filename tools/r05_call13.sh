# round 5, GPU call 13: 32-byte bucket lines (SYLPH_LINE_SLOTS=4) against the default 64-byte lines, alternated on one box:
# the containment / reassignment / pipeline parity tests with each library, then the C3 bench line (probe alone, index size, both modes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05j; mkdir -p $out
cp sylph_amd/libsylph_hip.so /tmp/keep.so
cp /tmp/keep.so sylph_amd/libsylph_hip.so.line8
for v in line8 line4; do
  cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_c1_full.py tests/test_gpu_fullsize.py -m gpu -x -q -k "contain or reassign or pipeline or replica or sharded or end_to_end or hit_row or c1 or full" > $out/pytest_$v.txt 2>&1; echo "$v: $(grep -n 'passed\|failed' $out/pytest_$v.txt | tail -1)"
done
for i in 1 2; do for v in line8 line4 "line4 3"; do
  set -- $v
  cp sylph_amd/libsylph_hip.so.$1 sylph_amd/libsylph_hip.so
  SYLPH_BENCH_CTX_OPTIONS=${2:+index_lambda=$2} python bench.py --steps 6 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-packed-leg --no-filter-leg --no-files-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('pipelined',{}); s=d.get('one_step_at_a_time',{}); rp=d.get('roofline_profile',{}); a=rp.get('alone_on_gpu',{})
print('$v', 'index_gb', d['setup']['index_gb'], 'pipelined', p.get('value'), p.get('ms_per_sample'), 'sequential', s.get('ms_per_sample'), 'probe alone ms', a.get('avg_launch_ms'), 'frac8d', a.get('frac'), 'probe in mix', rp.get('avg_launch_ms'), 'verify', d.get('verify',{}).get('mismatches'), 'db_index_s', d['setup']['db_upload_index_s'])"
done; done | tee $out/ab_line.txt
cp sylph_amd/libsylph_hip.so.line4 sylph_amd/libsylph_hip.so
B="python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify --no-kernel-timers --no-packed-leg --no-filter-leg --no-files-leg"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "probe_kernel" --output-format csv -d $out/pmc_$c -o s -- $B > /dev/null 2>&1
done
python tools/pmc_by_kernel.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_probe_line4.json; rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE; cat $out/pmc_probe_line4.json
cp /tmp/keep.so sylph_amd/libsylph_hip.so
