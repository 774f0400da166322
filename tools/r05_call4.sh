# round 5, GPU call 4: where the first sample of a `sylph-hip sketch` command spends its time (bring-up marks), and the whole-sample
# verify of the long-read (c5) and 8-tables-per-launch (c4) workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05d; mkdir -p $out
python - <<'PY'
import os, sys
sys.path.insert(0, "tools")
import numpy as np
import feed_bench as F
d="/tmp/feed_trace"; os.makedirs(d, exist_ok=True)
n_pairs=3333334; L=150
rng=np.random.default_rng(1)
genome=rng.choice(np.frombuffer(b"ACGT",dtype=np.uint8),size=20_000_000)
starts=rng.integers(0,len(genome)-400,size=n_pairs)
m1=genome[starts[:,None]+np.arange(L)[None,:]].reshape(-1)
F.write_fastq(f"{d}/s_1.fq",m1,L)
comp=np.zeros(256,dtype=np.uint8); comp[[65,67,71,84]]=[84,71,67,65]
m2=comp[genome[(starts[:,None]+399-np.arange(L)[None,:])]].reshape(-1)
F.write_fastq(f"{d}/s_2.fq",m2,L)
PY
d=/tmp/feed_trace
for rep in 1 2 3; do
  echo "== run $rep (default flags)"
  /usr/bin/time -f "wall %e s  user %U s  sys %S s" env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/s_1.fq -2 $d/s_2.fq -d $d/out 2>&1 | grep -v "^\[sylph_hip pgunzip\]" | head -60
done > $out/bringup_trace.txt 2>&1
echo "== dynamic linking" >> $out/bringup_trace.txt
LD_DEBUG=statistics sylph_amd/sylph-hip inspect /dev/null 2>&1 | grep -i "total startup\|relocation\|load" | head -8 >> $out/bringup_trace.txt
cat $out/bringup_trace.txt | head -90
timeout 900 python bench.py --workload c5 --steps 4 --warmup 1 --min-seconds 0.5 --no-h2d > $out/bench_c5.json 2> $out/bench_c5.err; tail -c 1500 $out/bench_c5.json | head -c 1500; echo
timeout 900 python bench.py --workload c4 --steps 4 --warmup 1 --min-seconds 0.5 --no-h2d --no-files-leg > $out/bench_c4.json 2> $out/bench_c4.err
python - <<PY
import json
for f in ("bench_c5", "bench_c4"):
    try:
        d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "verify", d.get("verify"), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("sample", "")[:160])
    except Exception as e:
        print(f, "unreadable:", e, open("$out/%s.err" % f).read()[-800:])
PY
