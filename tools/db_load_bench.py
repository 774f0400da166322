#!/usr/bin/env python3
"""`sylph-hip query / profile` on a GTDB-R220-scale .syldb FROM DISK: 113,104 genome sketches (1.8e9 k-mers, ~14 GB of
bincode with names and tracked k-mers) written to local disk in the reference's layout, one sample sketch, then the command itself:
whole-command wall clock and the database's load + index time the command logs (SYLPH_HIP_DEBUG), with the round-4 path (views of
the mapping gathered into page-locked upload chunks) and with SYLPH_HIP_DB_COPY_LOAD=1 (rounds 1-3: a vector per genome, one flat
copy, staged pageable upload).  The reference's own cost here is the single-threaded bincode read of contain.rs:492-500."""
import os
import struct
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth  # noqa: E402

BIN = os.path.join(ROOT, "sylph_amd", "sylph-hip")


def main():
    n_genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 113_104
    d = os.environ.get("SYLPH_DB_BENCH_DIR", "/tmp/db_load_bench")
    os.makedirs(d, exist_ok=True)
    dev = torch.device("cuda", 0)
    t0 = time.time()
    dk, doff = synth.decoy_sketches(n_genomes, c=200, device=dev, seed=20250718)
    k = dk.cpu().numpy().view(np.uint64)
    off = doff.cpu().numpy().astype(np.int64)
    del dk, doff
    # 13 % of every sketch as "tracked" k-mers (n_tracked = 0.13 / 0.87 of n_kept), taken from the end of the genome's own list
    path = f"{d}/gtdb_scale.syldb"
    with open(path, "wb", buffering=1 << 24) as f:
        f.write(struct.pack("<Q", n_genomes))
        for g in range(n_genomes):
            a, b = int(off[g]), int(off[g + 1])
            nt = (b - a) * 13 // 100
            f.write(struct.pack("<Q", b - a - nt))
            f.write(k[a:b - nt].tobytes())
            f.write(b"\x01" + struct.pack("<Q", nt))
            f.write(k[b - nt:b].tobytes())
            name = b"genomes/GCF_%09d.1_genomic.fna.gz" % g
            contig = b"NZ_CP%06d.1 Synthetic decoy %d chromosome, complete genome" % (g, g)
            f.write(struct.pack("<Q", len(name)) + name + struct.pack("<Q", len(contig)) + contig)
            f.write(struct.pack("<QQQQ", 200, 31, (b - a) * 200, 30))
    size = os.path.getsize(path)
    # one sample: 1.9 M entries, a few thousand of them k-mers of the database so that rows come out
    rng = np.random.default_rng(5)
    hit = np.concatenate([k[int(off[g]):int(off[g]) + 3000] for g in (5, 77, 1234)])
    sk = np.unique(np.concatenate([hit, rng.integers(0, (2**64 - 1) // 200, size=1_900_000, dtype=np.uint64)]))
    sc = rng.integers(1, 30, size=len(sk)).astype(np.uint32)
    rec = np.empty(len(sk), dtype=np.dtype([("k", "<u8"), ("c", "<u4")]))
    rec["k"], rec["c"] = sk, sc
    with open(f"{d}/sample.sylsp", "wb") as f:
        f.write(struct.pack("<Q", len(sk)) + rec.tobytes())
        name = b"sample.fq"
        f.write(struct.pack("<QQ", 200, 31) + struct.pack("<Q", len(name)) + name + b"\x00" + b"\x00" + struct.pack("<d", 150.0))
    del k
    res = {"genomes": n_genomes, "syldb_bytes": size, "write_s": round(time.time() - t0, 1), "host_threads": os.cpu_count()}
    for cmd in ("query", "profile"):
        variants = [("views", {}), ("views_again", {}), ("copy_load", {"SYLPH_HIP_DB_COPY_LOAD": "1"})]
        if os.environ.get("DBLOAD_DIAG"):       # back-to-back runs scatter: which knob, if any, matters
            variants += [("views_3", {}), ("views_nowarm", {"SYLPH_HIP_NO_WARMUP": "1"}), ("views_nowarm_again", {"SYLPH_HIP_NO_WARMUP": "1"}),
                         ("views_clean_exit", {"SYLPH_HIP_CLEAN_EXIT": "1"}), ("views_after_clean", {}), ("views_after_sleep", {"_SLEEP": "3"})]
        for label, env in variants:
            if env.get("_SLEEP"):
                time.sleep(float(env["_SLEEP"]))
                env = {}
            t = time.perf_counter()
            p = subprocess.run([BIN, cmd, path, f"{d}/sample.sylsp"], capture_output=True, text=True, env=dict(os.environ, SYLPH_HIP_DEBUG="1", **env))
            dt = time.perf_counter() - t
            assert p.returncode == 0, p.stderr[-2000:]
            db_s = [float(ln.split(" in ")[1].split(" s")[0]) for ln in p.stderr.split("\n") if "uploaded and indexed in" in ln]
            res[f"{cmd}_{label}"] = {"command_s": round(dt, 2), "db_upload_index_s": db_s[0] if db_s else None, "rows": len(p.stdout.strip().split("\n")) - 1,
                                     "stdout_md5": __import__("hashlib").md5(p.stdout.encode()).hexdigest()[:12]}
    print(res)
    os.remove(path)


if __name__ == "__main__":
    main()
