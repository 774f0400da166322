cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03_p; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; grep -n "passed\|failed" $out/pytest.txt
AB_ROUNDS=3 bash tools/ab_lib_r03.sh
cp sylph_amd/libsylph_hip.so.new sylph_amd/libsylph_hip.so
for wl in c3r c2; do python bench.py --workload $wl --steps 4 --warmup 1 --min-seconds 0.5 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['value'], d['one_step_at_a_time'].get('kernel_ms'), d['verify']['mismatches'])"; done
