cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03h; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; grep -n "passed\|failed" $out/pytest.txt; grep -n "Error\|assert" $out/pytest.txt | head -20
# feed breakdown: 4 warm paired samples in one command, with the feed's lap times
python - <<'PY'
import os, sys, subprocess, time
sys.path.insert(0, "tools")
import numpy as np
import feed_bench as F
d="/tmp/feed_bench"; os.makedirs(d, exist_ok=True)
n_pairs=3333334; L=150
rng=np.random.default_rng(1)
genome=rng.choice(np.frombuffer(b"ACGT",dtype=np.uint8),size=20_000_000)
starts=rng.integers(0,len(genome)-400,size=n_pairs)
m1=genome[starts[:,None]+np.arange(L)[None,:]].reshape(-1)
comp=np.zeros(256,dtype=np.uint8); comp[[65,67,71,84]]=[84,71,67,65]
m2=comp[genome[(starts[:,None]+399-np.arange(L)[None,:])]].reshape(-1)
F.write_fastq(f"{d}/s_1.fq",m1,L); F.write_fastq(f"{d}/s_2.fq",m2,L)
for i in range(4):
    for m in (1,2):
        dst=f"{d}/p{i}_{m}.fq"
        if os.path.lexists(dst): os.remove(dst)
        os.symlink(f"{d}/s_{m}.fq",dst)
t=time.perf_counter()
p=subprocess.run([F.BIN,"sketch","-1",*[f"{d}/p{i}_1.fq" for i in range(4)],"-2",*[f"{d}/p{i}_2.fq" for i in range(4)],"-d",f"{d}/out","-t","1","--fpr","0"],capture_output=True,text=True,env=dict(os.environ,SYLPH_HIP_FEED_TRACE="1"))
print("command", time.perf_counter()-t)
open("gpurun_out/r03h/feed_trace.txt","w").write(p.stderr)
print(p.stderr[-3500:])
PY
