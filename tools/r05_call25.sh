# round 5, GPU call 25: the device-side FASTQ route against the host feed on 8 plain pairs in one command (-t 1 and -t 3), and with the
# process confined to 4 CPUs (taskset) — the case the route is on by default for; CLI + FASTQ tests with the rebuilt host
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_fastq; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_fastq.py -m gpu -x -q > $out/pytest_cli2.txt 2>&1; tail -n 3 $out/pytest_cli2.txt
d=/tmp/feed_bench
FEED_BENCH_ONLY=plain python tools/feed_bench.py 3333334 > /dev/null 2>&1
for i in 4 5 6 7; do for m in 1 2; do ln -sf $d/s_$m.fq $d/p${i}_$m.fq; done; done
F1=$(for i in 0 1 2 3 4 5 6 7; do echo -n "$d/p${i}_1.fq "; done); F2=$(for i in 0 1 2 3 4 5 6 7; do echo -n "$d/p${i}_2.fq "; done)
run() {  # $1 = label, rest = command prefix
  local label=$1; shift
  for rep in 1 2; do for dev in 0 1; do
    s=$(date +%s.%N)
    env SYLPH_HIP_FEED_DEVICE=$dev "$@" > /dev/null 2> $out/err.txt
    e=$(date +%s.%N)
    echo "$label device_route=$dev command $(python -c "print(round($e-$s,3), 's =', round(8.0/($e-$s),2), 'Gbp/s')") samples $(grep -o 'written in [0-9.]* s' $out/err.txt | grep -o '[0-9.]*' | tr '\n' ' ')"
  done; done
}
{ run "8 pairs -t 1, 16 CPUs" sylph_amd/sylph-hip sketch -t 1 -1 $F1 -2 $F2 -d $d/o8
  run "8 pairs -t 3, 16 CPUs" sylph_amd/sylph-hip sketch -t 3 -1 $F1 -2 $F2 -d $d/o8
  run "8 pairs -t 1, 4 CPUs (taskset)" taskset -c 0-3 sylph_amd/sylph-hip sketch -t 1 -1 $F1 -2 $F2 -d $d/o8
  echo "default on 4 CPUs:"; ( time taskset -c 0-3 env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -t 1 -1 $d/p0_1.fq $d/p1_1.fq -2 $d/p0_2.fq $d/p1_2.fq -d $d/o8 ) 2>&1 | grep -c "device route: pushed"
} | tee $out/feed_device_route_ab.txt
