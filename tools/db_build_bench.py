"""`sylph-hip sketch` of a directory of genome files (FASTA, gzip): wall time of the database build with -t 1 and -t 32 (files are
parsed / inflated on the -t threads, sketched on the GPU in batches of 1 Gbp).  GPU box: python tools/db_build_bench.py [n_files]"""
import gzip, os, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 160
glen = 4_000_000
d = tempfile.mkdtemp(prefix="sylph_db_")
rng = np.random.default_rng(3)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
t = time.time()
paths = []
for i in range(n_files):
    seq = acgt[rng.integers(0, 4, size=glen)]
    lines = [b">g%d contig_1 synthetic" % i] + [seq[j:j + 80].tobytes() for j in range(0, glen, 80)]
    p = os.path.join(d, f"g{i}.fa.gz" if i % 2 else f"g{i}.fa")
    data = b"\n".join(lines) + b"\n"
    if i % 2:
        with gzip.open(p, "wb", compresslevel=1) as f:
            f.write(data)
    else:
        open(p, "wb").write(data)
    paths.append(p)
lst = os.path.join(d, "genomes.txt")
open(lst, "w").write("\n".join(paths) + "\n")
print(f"{n_files} genome files of {glen / 1e6:.0f} Mbp (half of them gzip) written in {time.time() - t:.1f} s")
exe = os.path.join(ROOT, "sylph_amd", "sylph-hip")
res = {}
for threads in (1, 8, 32):
    out = os.path.join(d, f"db_t{threads}")
    t = time.time()
    r = subprocess.run([exe, "sketch", "-l", lst, "-o", out, "-t", str(threads)], capture_output=True, text=True)
    dt = time.time() - t
    assert r.returncode == 0, r.stderr[-2000:]
    res[threads] = open(out + ".syldb", "rb").read()
    print(f"-t {threads}: {dt:.2f} s = {n_files * glen / 1e9 / dt:.2f} Gbp/s (database {len(res[threads]) / 1e6:.1f} MB)")
assert res[1] == res[8] == res[32], "the database must not depend on -t"
print("databases identical for every -t")
