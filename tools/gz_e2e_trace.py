"""Where a `.fastq.gz` sample's time goes in `sylph-hip sketch` (round 6): bench.py's 1 Gbp pair (150 bp reads, constant qualities) written
as two `gzip -1` files, then the command with SYLPH_HIP_FEED_TRACE=1 SYLPH_HIP_TRACE=1 — device inflate (default) and host inflate
(SYLPH_HIP_INFLATE_DEVICE=0) — and the plain files for scale."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import feed_bench as FB  # noqa: E402

n_pairs = int(float(sys.argv[1]) * 1e9 / 300) if len(sys.argv) > 1 else 3_333_334
d = tempfile.mkdtemp(prefix="sylph_gz_trace_")
rng = np.random.default_rng(1)
for m in (1, 2):
    FB.write_fastq(f"{d}/s_{m}.fq", rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n_pairs * 150), 150)
ps = [subprocess.Popen(["gzip", "-1", "-k", "-f", f"{d}/s_{m}.fq"]) for m in (1, 2)]
for p in ps:
    p.wait()
exe = os.path.join(ROOT, "sylph_amd", "sylph-hip")
for what, files, env in (("plain", ("s_1.fq", "s_2.fq"), {}), ("gz, device inflate", ("s_1.fq.gz", "s_2.fq.gz"), {}),
                         ("gz, device inflate (again)", ("s_1.fq.gz", "s_2.fq.gz"), {}),
                         ("gz, host inflate", ("s_1.fq.gz", "s_2.fq.gz"), {"SYLPH_HIP_INFLATE_DEVICE": "0"}),
                         ("gz x2 samples, device inflate", ("s_1.fq.gz", "s_2.fq.gz", "x2"), {})):
    e = dict(os.environ, SYLPH_HIP_FEED_TRACE="1", SYLPH_HIP_TRACE="1", **env)
    if len(files) == 3:
        args = ["-1", f"{d}/{files[0]}", f"{d}/{files[0]}", "-2", f"{d}/{files[1]}", f"{d}/{files[1]}", "-t", "1", "-S", "a", "b"]
    else:
        args = ["-1", f"{d}/{files[0]}", "-2", f"{d}/{files[1]}"]
    t = time.perf_counter()
    p = subprocess.run([exe, "sketch", *args, "-d", f"{d}/out"], capture_output=True, text=True, env=e)
    dt = time.perf_counter() - t
    print(f"==== {what}: {dt:.3f} s, rc {p.returncode}")
    keep = [ln for ln in p.stderr.split("\n") if "inflate" in ln or "device route" in ln or "t+" in ln or "timing" in ln or "pool miss" in ln]
    print("\n".join(keep[:120]))
