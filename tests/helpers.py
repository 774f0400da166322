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
