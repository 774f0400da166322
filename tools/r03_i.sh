cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03i; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q --durations=6 > $out/pytest.txt 2>&1; grep -n "passed\|failed" $out/pytest.txt; grep -n "Error\|assert" $out/pytest.txt | head -20; grep -A8 "slowest" $out/pytest.txt
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
for i in range(6):
    for m in (1,2):
        dst=f"{d}/p{i}_{m}.fq"
        if os.path.lexists(dst): os.remove(dst)
        os.symlink(f"{d}/s_{m}.fq",dst)
for t in ("1","2"):
    t0=time.perf_counter()
    p=subprocess.run([F.BIN,"sketch","-1",*[f"{d}/p{i}_1.fq" for i in range(6)],"-2",*[f"{d}/p{i}_2.fq" for i in range(6)],"-d",f"{d}/out","-t",t,"--fpr","0"],capture_output=True,text=True,env=dict(os.environ,SYLPH_HIP_FEED_TRACE="1" if t=="1" else "0"))
    dt=time.perf_counter()-t0
    per=[float(ln.split(" in ")[1].split(" s")[0]) for ln in p.stderr.split("\n") if "timing:" in ln]
    print("-t",t,"command",round(dt,3),"s for 6 Gbp =",round(6*2*n_pairs*L/1e9/dt,2),"Gbp/s; per sample s:",per, "rc", p.returncode)
    if t=="1": open("gpurun_out/r03i/feed_trace.txt","w").write(p.stderr); print(p.stderr[-1800:])
PY
