#!/usr/bin/env python3
"""What replacing `dup_removal_lsh_full` (sketch.rs:733-769: the reference's DEFAULT paired-end dedup, an approximate cuckoo filter at
--fpr 1e-4) by the exact marker set (`dup_removal_lsh_full_exact`, :690-731 — what the GPU path implements) costs in output terms.
CPU only (oracle/): a 1 Gbp paired sample (3,333,334 pairs of 2 x 150 bp from a 100-genome community with log-normal abundances,
0.5 % errors, 2 % exact duplicate pairs — the shape of bench.py's C3 reads) is sketched both ways; the tables are compared
(sum |delta count|, occurrences removed), and both are profiled against the community's genomes: ANI / coverage / abundance deltas.

The filter is the one of oracle/sylph_oracle.cpp: the structure of Fan et al. 2014 with the crate's documented defaults — a MODEL of
scalable_cuckoo_filter 0.2.4 (not under /root/reference), so the numbers are an estimate of the default path's distance from the
exact one, not a parity statement.  Also printed: the same for fpr 1e-3 (the value the reference falls back to for fpr = 0 inside
the approximate branch, sketch.rs:797) and 1e-2, to show how the distance scales.

    python tools/a10_bound.py [n_pairs]        ->  profiles/r03_a10_bound.txt is this script's output"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
COMP[[65, 67, 71, 84]] = [84, 71, 67, 65]


def community_reads(n_pairs, n_genomes=100, glen=5_000_000, L=150, seed=20250711):
    rng = np.random.default_rng(seed)
    genomes = rng.choice(ACGT, size=(n_genomes, glen)).astype(np.uint8)
    ab = np.exp(rng.standard_normal(n_genomes))
    gid = rng.choice(n_genomes, size=n_pairs, p=ab / ab.sum())
    ins = np.clip(np.round(rng.normal(350, 30, size=n_pairs)).astype(np.int64), L, 1050)
    start = (rng.random(n_pairs) * (glen - ins)).astype(np.int64)
    out = np.empty((n_pairs, 2, L), dtype=np.uint8)
    ar = np.arange(L)
    flat = genomes.reshape(-1)
    step = 1 << 18
    for s in range(0, n_pairs, step):
        e = min(n_pairs, s + step)
        base = gid[s:e] * glen + start[s:e]
        fwd = flat[base[:, None] + ar[None, :]]
        tail = COMP[flat[(base + ins[s:e] - 1)[:, None] - ar[None, :]]]
        flip = rng.random(e - s) < 0.5
        m1 = np.where(flip[:, None], tail, fwd)
        m2 = np.where(flip[:, None], fwd, tail)
        pair = np.stack([m1, m2], axis=1)
        err = rng.random(pair.shape) < 0.005
        pair[err] = rng.choice(ACGT, size=int(err.sum()))
        out[s:e] = pair
    n_dup = int(n_pairs * 0.02)
    src, dst = rng.integers(0, n_pairs, n_dup), rng.integers(0, n_pairs, n_dup)
    out[dst] = out[src]
    bases = out.reshape(-1)
    off = np.arange(0, 2 * n_pairs + 1, dtype=np.uint64) * np.uint64(L)
    return genomes, bases, off


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3_333_334
    t0 = time.time()
    genomes, bases, off = community_reads(n_pairs)
    print(f"sample: {n_pairs} pairs of 2 x 150 bp ({2 * n_pairs * 150 / 1e9:.3f} Gbp) from 100 x 5 Mbp genomes, generated in {time.time() - t0:.0f} s")
    t0 = time.time()
    exact = O.sketch_reads(bases, off, c=200, k=31, paired=True)
    t_exact = time.time() - t0
    n_occ = int(exact["counts"].sum()) + exact["dup_removed"]
    print(f"exact set (--fpr 0, the GPU path's semantics): {len(exact['kmers'])} distinct k-mers, {int(exact['counts'].sum())} counted, "
          f"{exact['dup_removed']} removed of {n_occ} occurrences  [{t_exact:.1f} s on one core]")
    # genome sketches + containment of both tables
    gk = [O.sketch_genome(g, np.array([0, len(g)], dtype=np.uint64), c=200)["genome_kmers"] for g in genomes]
    goff = np.zeros(len(gk) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(x) for x in gk])
    db = np.concatenate(gk)

    def profile(t):
        cc, covs, _ = O.contain(t["kmers"], t["counts"], db, goff, n_threads=8)
        res = []
        for g in range(len(gk)):
            st = O.stats(covs[g], len(gk[g])) if cc[g] else None
            res.append(st)
        return cc, res

    cc_e, st_e = profile(exact)
    for fpr in (1e-4, 1e-3, 1e-2):
        t0 = time.time()
        approx = O.sketch_reads_cuckoo_model(bases, off, c=200, k=31, fpr=fpr)
        dt = time.time() - t0
        assert np.array_equal(approx["kmers"], exact["kmers"])            # the k-mer set never differs, only counts
        d = approx["counts"].astype(np.int64) - exact["counts"].astype(np.int64)
        cc_a, st_a = profile(approx)
        d_ani = d_cov = d_lam = 0.0
        flips = 0
        cov_e = np.array([s.final_est_cov if s is not None and s.final_est_ani >= 0.95 else 0.0 for s in st_e])
        cov_a = np.array([s.final_est_cov if s is not None and s.final_est_ani >= 0.95 else 0.0 for s in st_a])
        for a, b in zip(st_e, st_a):
            if a is None or b is None:
                continue
            d_ani = max(d_ani, abs(a.final_est_ani - b.final_est_ani))
            d_cov = max(d_cov, abs(a.final_est_cov - b.final_est_cov) / max(a.final_est_cov, 1e-12))
            flips += int((a.final_est_ani >= 0.95) != (b.final_est_ani >= 0.95))
        ab_e, ab_a = cov_e / cov_e.sum() * 100, cov_a / cov_a.sum() * 100
        print(f"cuckoo model fpr {fpr:g}: sum |delta count| = {int(np.abs(d).sum())} ({np.abs(d).sum() / n_occ:.2e} of the occurrences; "
              f"{int((d != 0).sum())} k-mers touched, max |delta| {int(np.abs(d).max())}), removed {approx['dup_removed']} vs {exact['dup_removed']}; "
              f"containment counts equal: {bool(np.array_equal(cc_e, cc_a))}; max |delta adjusted ANI| {d_ani:.2e}, max relative |delta eff. coverage| "
              f"{d_cov:.2e}, genomes crossing the 95 % threshold: {flips}, max |delta taxonomic abundance| {np.abs(ab_e - ab_a).max():.2e} points  [{dt:.1f} s]")


if __name__ == "__main__":
    main()
