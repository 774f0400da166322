"""CPU: the two models of the filter dedup — oracle/sylph_oracle.cpp (victims drawn from an LCG) and oracle/pyref.py (round-robin victims) — on
400 random paired samples over c, --fpr and the filter's initial capacity, incl. capacities that fill the buckets to the brim.  Where insertions
cannot fail the two must agree count for count (what a cuckoo filter answers does not depend on where evictions left the fingerprints: the premise
of csrc/a10.hip); where they can (capacity 8 = 100 % load) they are expected to drift apart.  profiles/r04_a10_eviction_independence.txt"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O, pyref as P
rng = np.random.default_rng(2024)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
bad = 0
by_cap = {}
for it in range(400):
    g = rng.choice(acgt, size=int(rng.integers(300, 3000))).astype(np.uint8)
    n = int(rng.integers(20, 400))
    recs = []
    for _ in range(n):
        for m in range(2):
            L = int(rng.choice([20, 33, 34, 60, 100, 150, 250]))
            s = int(rng.integers(0, max(1, len(g) - L)))
            r = g[s:s + L].copy()
            recs.append(r.tobytes())
    for _ in range(n // 3):
        j = 2 * int(rng.integers(0, len(recs) // 2))
        recs += [recs[j], recs[j + 1]]
    c = int(rng.choice([1, 2, 5, 20]))
    fpr = float(rng.choice([1e-4, 0.01, 0.1, 0.3, 0.6]))
    cap = int(rng.choice([8, 40, 100, 600, 2500]))
    avx2 = bool(rng.integers(0, 2))
    b, off = O.concat(recs)
    try:
        e = O.sketch_reads_cuckoo_model(b, off, c=c, mode=O.MODE_AVX2_COMPAT if avx2 else O.MODE_SCALAR, fpr=fpr, initial_capacity=cap)
    except Exception as ex:
        print('oracle failed', it, ex); continue
    gq = P.sketch_pair_sequences(recs[0::2], recs[1::2], c, 31, avx2=avx2, dedup_fpr=fpr, initial_capacity=cap)
    ks = sorted(gq["kmer_counts"])
    same = ks == e["kmers"].tolist() and [gq["kmer_counts"][k] for k in ks] == e["counts"].tolist() and gq["dup_removed"] == e["dup_removed"]
    by_cap.setdefault(cap, [0, 0])[0] += 1
    if not same:
        by_cap[cap][1] += 1
        bad += 1
        print('MISMATCH', it, c, fpr, cap, avx2, gq["dup_removed"], e["dup_removed"])
print('done, mismatches', bad)
for cap in sorted(by_cap):
    nb = 1
    while nb * 4 < cap:
        nb <<= 1
    print(f'capacity {cap}: {nb} buckets, {100 * cap / (4 * nb):.0f} % load when full: {by_cap[cap][1]} of {by_cap[cap][0]} samples differ')
