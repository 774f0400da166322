# round 5, GPU call 24: (a) is the device briefly invisible to a NEW process right behind the CLI tests?  (b) the one-sample single-end command with
# SYLPH_HIP_FEED_DEVICE=1 (0.29 s against 0.23 in call 23: which route does it take?)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_fastq; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -n 3
  for i in 1 2 3 4; do date +%s.%N; python -c "
import ctypes,sys
h=ctypes.CDLL('libamdhip64.so'); n=ctypes.c_int(-1); rc=h.hipGetDeviceCount(ctypes.byref(n)); print('hipGetDeviceCount rc', rc, 'n', n.value)"; done ) > $out/probe_after_cli.txt 2>&1
cat $out/probe_after_cli.txt
d=/tmp/feed_bench
FEED_BENCH_ONLY=plain python tools/feed_bench.py 3333334 > /dev/null 2>&1
( for dev in 0 1 0 1; do echo "== single-end, SYLPH_HIP_FEED_DEVICE=$dev"; ( time env SYLPH_HIP_FEED_DEVICE=$dev SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -r $d/s_1.fq -d $d/outs$dev ) 2>&1 | grep -v "pgunzip\]" | head -60; done ) > $out/cli_single_trace.txt 2>&1
grep -n "==\|real\|device route\|timing\|index of" $out/cli_single_trace.txt | head -40
