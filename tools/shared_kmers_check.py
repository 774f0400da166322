"""Profile stage against a database in which many genomes share their k-mers (thousands of strains of one species, as in
undereplicated custom databases): a sample k-mer of that species hits every strain, index buckets carry long overflow runs, the
hit list grows from ~1 hit per probe to hundreds.  GPU box: python tools/shared_kmers_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import sylph_amd as S
from oracle import oracle as O

rng = np.random.default_rng(1)
thr = (2**64 - 1) // 200
ctx = S.Context(0)
base = np.unique(rng.integers(0, thr, size=25_000, dtype=np.uint64))           # one 5 Mbp species at c=200
decoys = [np.unique(rng.integers(0, thr, size=16_000, dtype=np.uint64)) for _ in range(5000)]
other = np.unique(rng.integers(0, thr, size=1_900_000, dtype=np.uint64))         # the rest of the sample
sk = np.unique(np.concatenate([base, other]))
sc = rng.integers(1, 6, size=len(sk)).astype(np.uint32)
for n_strains in (10, 100, 1000, 5000):
    strains = []
    for _ in range(n_strains):
        g = base.copy()
        rep = rng.random(len(g)) < 0.01
        g[rep] = rng.integers(0, thr, size=int(rep.sum()), dtype=np.uint64)
        strains.append(np.unique(g))
    gs = strains + decoys
    kmers = np.concatenate(gs)
    goff = np.zeros(len(gs) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g) for g in gs])
    t = time.perf_counter()
    db = S.Database(ctx, kmers, goff)
    t_up = time.perf_counter() - t
    ts = []
    for rep in range(3):
        ctx.profile(True)
        t = time.perf_counter()
        cc, off, covs = db.contain_view(sk, sc, packed=True)
        ts.append(time.perf_counter() - t)
        st = {f: tuple(round(x, 3) for x in ctx.kernel_stats(f)) for f in ("probe", "sort")}
        ctx.profile(False)
    n_hits = int(off[-1])
    ok = ""
    if True:                  # the oracle on the same inputs, for every case incl. the 5,000 strains (counts of all genomes, every 7th coverage vector)
        ecc, ecov, _ = O.contain(sk, sc, kmers, goff, n_threads=min(64, os.cpu_count() or 1))
        same = bool(np.array_equal(cc, ecc)) and all(np.array_equal(np.asarray(covs[int(off[g]):int(off[g + 1])]).astype(np.uint32), np.sort(ecov[g]))
                                                       for g in range(0, len(gs), 7))
        ok = f", oracle agrees: {same}"
    # profile mode: every strain passes; the winner table gives each shared k-mer to the strain with the highest ANI
    pg = np.arange(n_strains, dtype=np.uint32)
    pa = 0.95 + 0.05 * rng.random(n_strains)
    t = time.perf_counter()
    cc2, off2, covs2, lost2 = db.reassign_view(sk, sc, pg, pa)
    t_re = time.perf_counter() - t
    ok += f"; reassign {t_re * 1e3:.2f} ms ({int(off2[-1])} hits kept, {int(lost2.sum())} lost)"
    print(f"{n_strains} strains + {len(decoys)} decoys ({len(kmers) / 1e6:.0f} M postings): upload {t_up * 1e3:.0f} ms, contain {min(ts) * 1e3:.2f} ms "
          f"(host arrays in, results out), {n_hits} hits, kernels {st}{ok}", flush=True)
    db.close()
