# round 5, GPU call 8: the feed with the early-return fork; the replay kernel's variants (marker tags x wavefronts per bucket), alternated on one box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05h; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest_cli.txt 2>&1; tail -3 $out/pytest_cli.txt
python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err; cat $out/feed.txt; tail -3 $out/feed.err
cp sylph_amd/libsylph_hip.so /tmp/keep.so
for v in base tags b64 tags64; do
  cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "read_sketch or replay or bucket or deep" > $out/pytest_$v.txt 2>&1; echo "$v: $(tail -1 $out/pytest_$v.txt)"
done
for i in 1 2; do for v in base tags b64 tags64; do
  cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
  python bench.py --steps 6 --warmup 2 --min-seconds 1.0 --no-cpu-baseline --no-h2d --no-verify --no-packed-leg --no-files-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('pipelined',{}); s=d.get('one_step_at_a_time',{}); f=d.get('default_pair_dedup',{})
print('$v', 'pipelined', p.get('value'), p.get('ms_per_sample'), 'sequential', s.get('ms_per_sample'), (s.get('kernel_ms') or {}).get('replay'), 'filter', f.get('pipelined',{}).get('value'), (f.get('one_step_at_a_time',{}).get('kernel_ms') or {}).get('replay'))"
done; done | tee $out/ab_replay.txt
for v in base tags tags64; do
  cp sylph_amd/libsylph_hip.so.$v sylph_amd/libsylph_hip.so
  echo "== $v"; python tools/deep_coverage_check.py 2>/dev/null | grep "paired=True"
done | tee $out/ab_replay_deep.txt
cp /tmp/keep.so sylph_amd/libsylph_hip.so
