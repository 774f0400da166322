# command time of four 1 Gbp pairs (plain, single-member .gz) in one `sylph-hip sketch` for several parse-thread counts
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/feed_sweep; mkdir -p $out
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, "tools")
import numpy as np
import feed_bench as F
d="/tmp/feed_sweep"; os.makedirs(d, exist_ok=True)
n_pairs=3333334; L=150
rng=np.random.default_rng(1)
genome=rng.choice(np.frombuffer(b"ACGT",dtype=np.uint8),size=20_000_000)
starts=rng.integers(0,len(genome)-400,size=n_pairs)
m1=genome[starts[:,None]+np.arange(L)[None,:]].reshape(-1)
F.write_fastq(f"{d}/s_1.fq",m1,L)
comp=np.zeros(256,dtype=np.uint8); comp[[65,67,71,84]]=[84,71,67,65]
m2=comp[genome[(starts[:,None]+399-np.arange(L)[None,:])]].reshape(-1)
F.write_fastq(f"{d}/s_2.fq",m2,L)
ps=[subprocess.Popen(["gzip","-1","-k","-f",f"{d}/s_{m}.fq"]) for m in (1,2)]
[p.wait() for p in ps]
for i in range(4):
    for m in (1,2):
        for ext in ("fq", "fq.gz"):
            dst=f"{d}/p{i}_{m}.{ext}"
            if os.path.lexists(dst): os.remove(dst)
            os.symlink(f"{d}/s_{m}.{ext}", dst)
PY
d=/tmp/feed_sweep
for pt in ${SWEEP:-default 16 24 32 48}; do
  if [ $pt = default ]; then unset SYLPH_HIP_PARSE_THREADS; else export SYLPH_HIP_PARSE_THREADS=$pt; fi
  for kind in ${KINDS:-fq fq.gz}; do
    best=99999
    for rep in 1 2; do
      t0=$(date +%s%N)
      sylph_amd/sylph-hip sketch -1 $d/p0_1.$kind $d/p1_1.$kind $d/p2_1.$kind $d/p3_1.$kind -2 $d/p0_2.$kind $d/p1_2.$kind $d/p2_2.$kind $d/p3_2.$kind -d $d/out -t 1 > /dev/null 2>&1
      t1=$(date +%s%N)
      ms=$(( (t1 - t0) / 1000000 ))
      if [ $ms -lt $best ]; then best=$ms; fi
    done
    echo "parse threads $pt  $kind  ${best} ms"
  done
done | tee $out/sweep.txt
