# round 5, GPU call 5: the database cut by genome inside the library (tests, the two-rank small bench), one process over two replicas,
# and where the first sample of a `sylph-hip sketch` command spends its time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sharded" > $out/pytest_sharded.txt 2>&1; tail -5 $out/pytest_sharded.txt
MASTER_PORT=29583 timeout 600 python bench.py --gpus 2 --workload small --db-mode genome --steps 3 --warmup 1 --min-seconds 0.3 --no-cpu-baseline --no-h2d > $out/bench_small_2ranks_genome.json 2> $out/bench_small_2ranks_genome.err; tail -c 400 $out/bench_small_2ranks_genome.json; echo
MASTER_PORT=29585 timeout 600 python bench.py --gpus 2 --workload small --steps 3 --warmup 1 --min-seconds 0.3 --no-cpu-baseline --no-h2d > $out/bench_small_2ranks_replicate.json 2> $out/bench_small_2ranks_replicate.err; tail -c 400 $out/bench_small_2ranks_replicate.json; echo
timeout 900 python tools/multi_gpu_pipeline_bench.py --gpus 2 --share > $out/multi_2replicas_one_gpu.json 2> $out/multi.err; cat $out/multi_2replicas_one_gpu.json; tail -3 $out/multi.err
timeout 900 python tools/multi_gpu_pipeline_bench.py --gpus 1 > $out/multi_1replica.json 2>> $out/multi.err; cat $out/multi_1replica.json
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
  ( time env SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/s_1.fq -2 $d/s_2.fq -d $d/out ) 2>&1 | grep -v "pgunzip\]" | head -70
done > $out/bringup_trace.txt 2>&1
cat $out/bringup_trace.txt | head -120
