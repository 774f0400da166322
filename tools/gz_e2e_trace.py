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
if os.environ.get("GZ_TRACE_READS", "random") == "genome":
    # reads drawn from a 100-genome community the way bench.py's are (abundances, 0.5 % errors, 2 % duplicate pairs): deflate finds its
    # matches a read's coverage away, not next door
    import torch
    sys.path.insert(0, ROOT)
    import synth
    dev = torch.device("cuda", 0)
    genomes = synth.random_genomes(100, 5_000_000, dev, seed=20250711)
    bases, _ = synth.paired_reads(genomes, n_pairs, seed=20250711 + 1000003)[:2]
    hb = bases[: n_pairs * 300].cpu().numpy().reshape(n_pairs, 2, 150)
    for m in (1, 2):
        FB.write_fastq(f"{d}/s_{m}.fq", np.ascontiguousarray(hb[:, m - 1, :]).reshape(-1), 150)
    del genomes, bases, hb
    torch.cuda.empty_cache()
else:
    for m in (1, 2):
        FB.write_fastq(f"{d}/s_{m}.fq", rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n_pairs * 150), 150)
ps = [subprocess.Popen(["gzip", "-1", "-k", "-f", f"{d}/s_{m}.fq"]) for m in (1, 2)]
for p in ps:
    p.wait()
hold = float(os.environ.get("GZ_TRACE_HOLD_GB", "0"))
if hold:                                   # bench.py's situation: the commands run beside a process that holds a context and HBM
    import torch
    held = torch.empty(int(hold * (1 << 30)), dtype=torch.uint8, device="cuda")
    held.fill_(1)
    torch.cuda.synchronize()
    print(f"(holding {hold} GB of HBM in this process while the commands run)")
exe = os.path.join(ROOT, "sylph_amd", "sylph-hip")
for what, files, env in (("plain", ("s_1.fq", "s_2.fq"), {}), ("gz, device inflate", ("s_1.fq.gz", "s_2.fq.gz"), {}),
                         ("gz, device inflate (again)", ("s_1.fq.gz", "s_2.fq.gz"), {}),
                         ("gz, host inflate", ("s_1.fq.gz", "s_2.fq.gz"), {"SYLPH_HIP_INFLATE_DEVICE": "0"}),
                         ("gz x4 samples, device inflate", ("s_1.fq.gz", "s_2.fq.gz", "x2"), {}),
                         ("gz x4 samples, device inflate (again)", ("s_1.fq.gz", "s_2.fq.gz", "x2"), {})):
    e = dict(os.environ, SYLPH_HIP_FEED_TRACE="1", SYLPH_HIP_TRACE="1", **env)
    if len(files) == 3:
        args = ["-1", *[f"{d}/{files[0]}"] * 4, "-2", *[f"{d}/{files[1]}"] * 4, "-t", "1", "-S", "a", "b", "c", "e"]
    else:
        args = ["-1", f"{d}/{files[0]}", "-2", f"{d}/{files[1]}"]
    t = time.perf_counter()
    p = subprocess.run([exe, "sketch", *args, "-d", f"{d}/out"], capture_output=True, text=True, env=e)
    dt = time.perf_counter() - t
    print(f"==== {what}: {dt:.3f} s, rc {p.returncode}")
    keep = [ln for ln in p.stderr.split("\n") if "inflate" in ln or "device route" in ln or "t+" in ln or "timing" in ln or "pool miss" in ln]
    print("\n".join(keep[:120]))
