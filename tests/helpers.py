import numpy as np


def xor_sum(a):
    x = 0
    for v in np.asarray(a).tolist():
        x ^= int(v)
    return x, int(np.sum(np.asarray(a, dtype=np.uint64), dtype=np.uint64))


def hist(c):
    u, n = np.unique(np.asarray(c), return_counts=True)
    return {int(a): int(b) for a, b in zip(u, n)}


ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_seq(rng, n, alphabet=ACGT):
    return rng.choice(alphabet, size=n).astype(np.uint8)


def revcomp(seq):
    comp = np.zeros(256, dtype=np.uint8)
    comp[:] = ord("N")
    for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
        comp[a] = b
    return comp[np.asarray(seq, dtype=np.uint8)][::-1].copy()


def concat(records):
    off = np.zeros(len(records) + 1, dtype=np.uint64)
    if records:
        off[1:] = np.cumsum([len(r) for r in records], dtype=np.uint64)
    bases = np.concatenate([np.asarray(r, dtype=np.uint8) for r in records]) if records else np.zeros(0, dtype=np.uint8)
    return bases.astype(np.uint8), off


def bgzf_compress(data, block=65280, level=6):
    """What `bgzip` writes: gzip members of <= 64 KiB, each with the 'BC' extra subfield holding its compressed size, and the
    empty end-of-file member."""
    import struct
    import zlib
    out = bytearray()
    for a in list(range(0, len(data), block)) + [None]:
        chunk = b"" if a is None else data[a:a + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += body + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)
