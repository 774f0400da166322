# where a warm sample's time goes in `sylph-hip sketch`: four paired 1 Gbp samples in one command (-t 1) with SYLPH_HIP_FEED_TRACE=1,
# for several settings of the parse-thread count (SYLPH_HIP_PARSE_THREADS; default: see parse_threads() in host/feed.cpp)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/feed_trace; mkdir -p $out
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
for i in range(4):
    for m in (1,2):
        dst=f"{d}/p{i}_{m}.fq"
        if os.path.lexists(dst): os.remove(dst)
        os.symlink(f"{d}/s_{m}.fq", dst)
PY
d=/tmp/feed_trace
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)" | head -8 | tee $out/lscpu.txt
for pt in default 16 32 128; do for rep in 1 2; do
  if [ $pt = default ]; then unset SYLPH_HIP_PARSE_THREADS; else export SYLPH_HIP_PARSE_THREADS=$pt; fi
  echo "== parse threads $pt"
  SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/p0_1.fq $d/p1_1.fq $d/p2_1.fq $d/p3_1.fq -2 $d/p0_2.fq $d/p1_2.fq $d/p2_2.fq $d/p3_2.fq -d $d/out -t 1 --fpr 0 2>&1 | grep "index of\|timing" | sed 's/.*index of .*fq *\([0-9.]* ms\).*/index \1/; s/.*sketched + written in \([0-9.]* s\).*/SAMPLE \1/' | tr '\n' ' '; echo
done; done | tee $out/threads.txt
