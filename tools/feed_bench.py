#!/usr/bin/env python3
"""End-to-end host-feed measurement: synthetic paired FASTQ files on local disk -> `sylph-hip sketch` -> .sylsp,
plain and gzip, wall clock of the whole command (process start, parsing, H2D, GPU, sketch file written)."""
import os
os.environ.pop("SYLPH_HIP_EXACT_DEDUP", None)         # default flags, as a user runs them: pairs behind the cuckoo filter (--fpr 1e-4)
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "sylph_amd", "sylph-hip")


def write_fastq(path, seqs, L):
    n = len(seqs) // L
    rec = np.empty((n, 12 + L + 3 + L + 1), dtype=np.uint8)
    ids = np.arange(n)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("r")
    for d in range(9):
        rec[:, 10 - d] = 48 + (ids // 10 ** d) % 10
    rec[:, 11] = 10
    rec[:, 12:12 + L] = seqs.reshape(n, L)
    rec[:, 12 + L:12 + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 12 + L + 3:12 + 2 * L + 3] = ord("I")
    rec[:, -1] = 10
    rec.tofile(path)
    return rec.nbytes


def _bgzf_range(args):
    import struct
    import zlib
    path, a, z = args
    with open(path, "rb") as f:
        f.seek(a)
        data = f.read(z - a)
    out = bytearray()
    for o in range(0, len(data), 65280):
        chunk = data[o:o + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
        out += body + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


def bgzf_file(src, dst):
    """what `bgzip -l 1` writes (blocked gzip: members of <= 64 KiB with their size in the header), compressed on all cores"""
    from multiprocessing import Pool
    size = os.path.getsize(src)
    step = 65280 * 512
    with Pool(min(32, os.cpu_count() or 1)) as pool, open(dst, "wb") as f:
        for part in pool.imap(_bgzf_range, [(src, a, min(size, a + step)) for a in range(0, size, step)]):
            f.write(part)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    L = 150
    d = "/tmp/feed_bench"
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(1)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=20_000_000)
    starts = rng.integers(0, len(genome) - 400, size=n_pairs)
    idx = starts[:, None] + np.arange(L)[None, :]
    m1 = genome[idx].reshape(-1)
    comp = np.zeros(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    m2 = comp[genome[(starts[:, None] + 399 - np.arange(L)[None, :])]].reshape(-1)
    b1 = write_fastq(f"{d}/s_1.fq", m1, L)
    write_fastq(f"{d}/s_2.fq", m2, L)
    gbp = 2 * n_pairs * L / 1e9
    plain_only = os.environ.get("FEED_BENCH_ONLY") == "plain"       # (the A/B of the device-side FASTQ route: nothing compressed)
    if not plain_only:
        subprocess.run(["gzip", "-1", "-k", "-f", f"{d}/s_1.fq", f"{d}/s_2.fq"], check=True)
        bgzf_file(f"{d}/s_1.fq", f"{d}/b_1.fq.gz")
        bgzf_file(f"{d}/s_2.fq", f"{d}/b_2.fq.gz")
    sys.path.insert(0, ROOT)
    import bench as B
    res = {"gbp": gbp, "fastq_bytes_per_file": b1, "host_threads": os.cpu_count(), "host_cpus_usable": B.effective_cpus()}
    for name, a in (("paired_plain", ["-1", f"{d}/s_1.fq", "-2", f"{d}/s_2.fq"]), ("paired_gz", ["-1", f"{d}/s_1.fq.gz", "-2", f"{d}/s_2.fq.gz"]),
                    ("paired_bgzf", ["-1", f"{d}/b_1.fq.gz", "-2", f"{d}/b_2.fq.gz"]),
                    ("single_plain", ["-r", f"{d}/s_1.fq"]), ("single_gz", ["-r", f"{d}/s_1.fq.gz"])):
        if plain_only and "plain" not in name:
            continue
        best, inner = 1e9, 1e9
        for _ in range(2):
            t = time.perf_counter()
            p = subprocess.run([BIN, "sketch", *a, "-d", f"{d}/out"], capture_output=True, text=True)
            dt = time.perf_counter() - t
            assert p.returncode == 0, p.stderr[-2000:]
            best = min(best, dt)
            for ln in p.stderr.split("\n"):          # "timing: <file> sketched + written in X s": the sample alone, without process start
                if "timing:" in ln:
                    inner = min(inner, float(ln.split(" in ")[1].split(" s")[0]))
        g = gbp if name.startswith("paired") else gbp / 2
        res[name] = {"command_seconds": round(best, 3), "command_gbp_per_s": round(g / best, 3), "sample_seconds": round(inner, 3),
                     "sample_gbp_per_s": round(g / inner, 3)}
    # the same paired sample through the reader-thread path (what .gz input uses), for comparison
    p = subprocess.run([BIN, "sketch", "-1", f"{d}/s_1.fq", "-2", f"{d}/s_2.fq", "-d", f"{d}/out"], capture_output=True, text=True,
                       env=dict(os.environ, SYLPH_HIP_SEQUENTIAL_FEED="1"))
    for ln in p.stderr.split("\n"):
        if "timing:" in ln:
            res["paired_plain_sequential_feed"] = {"sample_gbp_per_s": round(gbp / float(ln.split(" in ")[1].split(" s")[0]), 3)}
    # four uncompressed paired samples in ONE command (GPU bring-up paid once; -t 1: one sample after the other)
    for i in range(4):
        for m in (1, 2):
            dst = f"{d}/p{i}_{m}.fq"
            if os.path.lexists(dst):
                os.remove(dst)
            os.symlink(f"{d}/s_{m}.fq", dst)
    tm = time.perf_counter()
    p = subprocess.run([BIN, "sketch", "-1", *[f"{d}/p{i}_1.fq" for i in range(4)], "-2", *[f"{d}/p{i}_2.fq" for i in range(4)], "-d", f"{d}/out",
                        "-t", "1"], capture_output=True, text=True)
    dt = time.perf_counter() - tm
    assert p.returncode == 0, p.stderr[-2000:]
    per = sorted(float(ln.split(" in ")[1].split(" s")[0]) for ln in p.stderr.split("\n") if "timing:" in ln)
    res["four_paired_plain_samples_t1"] = {"command_seconds": round(dt, 3), "command_gbp_per_s": round(4 * gbp / dt, 3),
                                           "fastest_sample_gbp_per_s": round(gbp / per[0], 3) if per else None,
                                           "sample_seconds_sorted": [round(x, 4) for x in per],
                                           "note": "the first sample of a process includes GPU bring-up and the un-hidden index of its files; the later ones are warm"}
    if plain_only:
        print(res)
        return
    # several samples in one command: -t worker threads, one GPU context each
    for i in range(4):
        for m in (1, 2):
            dst = f"{d}/m{i}_{m}.fq.gz"
            if os.path.lexists(dst):
                os.remove(dst)
            os.symlink(f"{d}/s_{m}.fq.gz", dst)
    firsts = [f"{d}/m{i}_1.fq.gz" for i in range(4)]
    seconds = [f"{d}/m{i}_2.fq.gz" for i in range(4)]
    for t in (1, 4):
        tm = time.perf_counter()
        p = subprocess.run([BIN, "sketch", "-1", *firsts, "-2", *seconds, "-d", f"{d}/out", "-t", str(t)], capture_output=True, text=True)
        dt = time.perf_counter() - tm
        assert p.returncode == 0, p.stderr[-2000:]
        res[f"four_paired_gz_samples_t{t}"] = {"seconds": round(dt, 3), "gbp_per_s": round(4 * gbp / dt, 3)}
    print(res)


if __name__ == "__main__":
    main()
