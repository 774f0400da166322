# round 5, GPU call 12: why back-to-back database loads scatter (diagnostic variants), and random 64 / 32 / 16-byte reads at the probe's shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05i; mkdir -p $out
./tools/random_line_rates > $out/random_line_rates.txt 2>&1; cat $out/random_line_rates.txt
DBLOAD_DIAG=1 timeout 1200 python tools/db_load_bench.py > $out/db_load_diag.txt 2> $out/db_load_diag.err; python - <<'PY'
import ast
d = ast.literal_eval(open("gpurun_out/r05i/db_load_diag.txt").read().strip().splitlines()[-1])
for k, v in d.items():
    if isinstance(v, dict): print(k, v["command_s"], v["db_upload_index_s"])
PY
free -g | head -3; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null
