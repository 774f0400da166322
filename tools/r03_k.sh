cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03m; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; grep -n "passed\|failed" $out/pytest.txt; grep -n "Error\|assert" $out/pytest.txt | head -20
python bench.py --steps 10 --warmup 2 --min-seconds 1.5 --no-cpu-baseline --no-h2d > $out/bench_c3.json 2> $out/bench_c3.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03m/bench_c3.json").read().strip().splitlines()[-1])
for k in ("value","mode","ms_per_sample","timed_region_s","sample_interval_ms","calibration","kernel_ms"): print(k, d.get(k))
for k in ("one_step_at_a_time",): print(k, {x:d[k].get(x) for x in ("value","ms_per_sample","kernel_ms","sketch_ms","profile_ms")})
print("verify", d.get("verify")); print(d.get("roofline_profile"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o c3 -- python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify > $out/bench_prof_seq.json 2> $out/prof.err
f=$(find $out/stats -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f --steps 5 --anchor reads_kernel > $out/step_timeline.md
rm -rf $out/stats
head -36 $out/step_timeline.md
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
for t,env in (("1",{}),):
    t0=time.perf_counter()
    p=subprocess.run([F.BIN,"sketch","-1",*[f"{d}/p{i}_1.fq" for i in range(6)],"-2",*[f"{d}/p{i}_2.fq" for i in range(6)],"-d",f"{d}/out","-t",t,"--fpr","0"],capture_output=True,text=True,env=dict(os.environ,**env))
    dt=time.perf_counter()-t0
    per=[float(ln.split(" in ")[1].split(" s")[0]) for ln in p.stderr.split("\n") if "timing:" in ln]
    print("-t",t,env,"command",round(dt,3),"s for 6 Gbp =",round(6*2*n_pairs*L/1e9/dt,2),"Gbp/s; per sample s:",per, "rc", p.returncode)
    if "SYLPH_HIP_FEED_TRACE" in env: open("gpurun_out/r03m/feed_trace.txt","w").write(p.stderr); print(p.stderr[-1500:])
PY
