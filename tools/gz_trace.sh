# where the time of a single-member .gz sample goes in `sylph-hip sketch` (host/pgunzip.cpp + feed): four paired 1 Gbp gz samples in
# one command (-t 1), SYLPH_HIP_FEED_TRACE=1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/gz_trace; mkdir -p $out
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, "tools")
import numpy as np
import feed_bench as F
d="/tmp/gz_trace"; os.makedirs(d, exist_ok=True)
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
        dst=f"{d}/m{i}_{m}.fq.gz"
        if os.path.lexists(dst): os.remove(dst)
        os.symlink(f"{d}/s_{m}.fq.gz", dst)
PY
d=/tmp/gz_trace
for rep in 1 2; do
  t0=$(date +%s.%N); env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/m0_1.fq.gz $d/m1_1.fq.gz $d/m2_1.fq.gz $d/m3_1.fq.gz -2 $d/m0_2.fq.gz $d/m1_2.fq.gz $d/m2_2.fq.gz $d/m3_2.fq.gz -d $d/out -t 1 > $out/run$rep.txt 2>&1
  t1=$(date +%s.%N); echo "run $rep: $(echo "$t1 - $t0" | bc) s"
  grep "timing" $out/run$rep.txt
done
grep -v "stretch" $out/run2.txt | head -150 > $out/run2_short.txt
