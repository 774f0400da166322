"""Device inflate (csrc/inflate.hip) alone: a synthetic FASTQ file of --mbp bases — bench.py's shape (150 bp reads, constant
qualities) or Illumina-like (--qual binned: headers with coordinates, binned qualities in runs) — compressed with the box's `gzip -L`
(and, with --bgzf, as BGZF through Python's zlib), inflated through sylph_inflate: wall clock of the call (H2D of the compressed bytes
included), the kernel families' times, bytes checked against zlib.  One JSON line per (file, level)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sylph_amd as S  # noqa: E402

FAMILIES = ["inflate_scan", "inflate_decode", "inflate_windows", "inflate_translate", "inflate_crc"]


def make_text(mbp, qual, seed):
    rng = np.random.default_rng(seed)
    L = 150
    n = int(mbp * 1e6) // L
    if qual == "const":
        import feed_bench as FB
        d = tempfile.mkdtemp(prefix="sylph_inflate_")
        p = os.path.join(d, "s.fq")
        FB.write_fastq(p, rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n * L), L)
        return p
    d = tempfile.mkdtemp(prefix="sylph_inflate_")
    p = os.path.join(d, "s.fq")
    qa = np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8)
    with open(p, "wb") as f:
        for a in range(0, n, 200000):
            m = min(200000, n - a)
            ids = np.arange(a, a + m)
            hdr = np.char.add(np.char.add(b"@A00123:45:HXXXXXXXX:1:", (1101 + ids // 9000).astype("S")),
                              np.char.add(np.char.add(b":", (1000 + (ids * 37) % 30000).astype("S")), np.char.add(np.char.add(b":", ids.astype("S")), b" 1:N:0:ACGT")))
            seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(m, L))
            q = np.repeat(rng.choice(qa, size=(m, L // 6)), 6, axis=1)
            lines = []
            for i in range(m):
                lines.append(hdr[i] + b"\n" + seq[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")
            f.write(b"".join(lines))
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=250)
    ap.add_argument("--levels", default="1,6")
    ap.add_argument("--qual", default="const", choices=["const", "binned"])
    ap.add_argument("--bgzf", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    path = make_text(a.mbp, a.qual, 1)
    text_bytes = os.path.getsize(path)
    ctx = S.Context(0)
    ctx.profile(True)
    files = []
    for lv in a.levels.split(","):
        out = f"{path}.{lv}.gz"
        t = time.perf_counter()
        with open(out, "wb") as f:
            subprocess.check_call(["gzip", f"-{lv}", "-c", path], stdout=f)
        files.append((f"gzip -{lv}", out, time.perf_counter() - t))
    if a.bgzf:
        from feed_bench import _bgzf_range
        out = f"{path}.bgzf.gz"
        t = time.perf_counter()
        with open(out, "wb") as f:
            step = 65280 * 512
            for o in range(0, text_bytes, step):
                f.write(_bgzf_range((path, o, min(text_bytes, o + step))))
            f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
        files.append(("bgzf (zlib level 1)", out, time.perf_counter() - t))
    expect = open(path, "rb").read() if a.check else None
    for what, p, t_compress in files:
        gz = np.fromfile(p, dtype=np.uint8)
        best = None
        for rep in range(a.reps):
            before = {f: ctx.kernel_stats(f)[0] for f in FAMILIES}
            t = time.perf_counter()
            inf = S.Inflated(ctx, gz)
            dt = time.perf_counter() - t
            fam = {f: round(ctx.kernel_stats(f)[0] - before[f], 3) for f in FAMILIES}
            rec = dict(what=what, qual=a.qual, text_bytes=text_bytes, gz_bytes=int(len(gz)), ratio=round(text_bytes / len(gz), 2), call_ms=round(dt * 1e3, 2),
                       text_gb_per_s=round(text_bytes / dt / 1e9, 2), gbp_per_s_if_fastq=round(a.mbp / 1e3 / dt, 2), kernel_ms=fam, kernel_ms_sum=round(sum(fam.values()), 3),
                       blocks=inf.n_blocks, candidates=inf.n_candidates, members=inf.n_members, rep=rep)
            if a.check and rep == 0:
                rec["bytes_equal_zlib"] = bool(inf.read().tobytes() == expect)
            inf.close()
            if best is None or rec["call_ms"] < best["call_ms"]:
                best = rec
        # zlib on one thread, for scale
        if a.check:
            t = time.perf_counter()
            zlib.decompress(gz.tobytes(), 31)
            best["zlib_one_thread_ms"] = round((time.perf_counter() - t) * 1e3, 1)
        print(json.dumps(best), flush=True)


if __name__ == "__main__":
    main()
