#!/usr/bin/env python3
"""host/pgunzip.cpp alone: one `gzip -1` FASTQ file of N read pairs' mate 1 (default 3,333,334 x 150 bp = 1.05 GB inflated), inflated by
T threads; phases from SYLPH_HIP_FEED_TRACE.  Also zlib on one thread for scale."""
import ctypes as C
import os
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import feed_bench as FB  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_333_334
    d = "/tmp/pgz_bench"
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(1)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=20_000_000)
    starts = rng.integers(0, len(genome) - 400, size=n)
    m1 = genome[starts[:, None] + np.arange(150)[None, :]].reshape(-1)
    raw = FB.write_fastq(f"{d}/s.fq", m1, 150)
    subprocess.run(["gzip", "-1", "-k", "-f", f"{d}/s.fq"], check=True)
    gz = os.path.getsize(f"{d}/s.fq.gz")
    t = time.perf_counter()
    zlib.decompress(open(f"{d}/s.fq.gz", "rb").read(), 31)
    print(f"raw {raw} B, gz {gz} B; zlib on one thread: {time.perf_counter() - t:.2f} s", flush=True)
    L = C.CDLL(os.path.join(ROOT, "sylph_amd", "libsylph_host.so"))
    L.sylph_host_pgunzip.argtypes = [C.c_char_p, C.c_uint, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64]
    os.environ["SYLPH_HIP_FEED_TRACE"] = "1"
    import threading
    for rep in range(2):
        for thr in (16, 24, 32, 48, 64, 96, 128):
            nn, cc = C.c_uint64(0), C.c_uint32(0)
            t = time.perf_counter()
            rc = L.sylph_host_pgunzip(f"{d}/s.fq.gz".encode(), thr, C.byref(nn), C.byref(cc), None, 0)
            dt = time.perf_counter() - t
            print(f"threads {thr}: rc {rc}, {nn.value} B in {dt:.3f} s (incl. reading the file + a second CRC pass in the test hook)", flush=True)
    # two files at once (the two mates of a pair), each with `thr` threads
    for thr in (32, 48, 64):
        def one():
            nn, cc = C.c_uint64(0), C.c_uint32(0)
            L.sylph_host_pgunzip(f"{d}/s.fq.gz".encode(), thr, C.byref(nn), C.byref(cc), None, 0)
        t = time.perf_counter()
        ts = [threading.Thread(target=one) for _ in range(2)]
        [x.start() for x in ts]
        [x.join() for x in ts]
        print(f"TWO files at once, {thr} threads each: {time.perf_counter() - t:.3f} s", flush=True)


if __name__ == "__main__":
    main()
