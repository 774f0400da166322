#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ from the reference's own test data.

Run in the build container only (needs /root/reference/test_files, which does not exist on the GPU box):
    python tests/golden/make_golden.py

Outputs
  ecoli_slices.npz  — slices of the three E. coli FASTAs (inputs) + oracle genome sketches / seeds of the slices
  k12_reads.npz     — k12_R1/R2.fq and t1/t2.fq sequences (inputs) + oracle read sketches (single, paired, no-dedup,
                      duplicated-file variants from SURVEY.md Appendix A.2)
  full_genome_kat.json — oracle results on the FULL genomes/reads (lengths, xor, wrapping sums, containment,
                      statistics) for comparison with survey_kat.json (independent numpy restatement)
The reference binary cannot be built here (pure Rust, no toolchain), so these are oracle outputs, cross-checked
against the survey's independent restatement — "parity unpinned" with respect to a real sylph binary.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402

T = "/root/reference/test_files/"
GENOMES = ["e.coli-EC590.fasta.gz", "e.coli-K12.fasta.gz", "e.coli-o157.fasta.gz"]


def xs(a):
    x = 0
    for v in a.tolist():
        x ^= v
    return [int(len(a)), int(x), int(np.sum(a.astype(np.uint64), dtype=np.uint64))]


def hist(c):
    u, n = np.unique(c, return_counts=True)
    return {str(int(a)): int(b) for a, b in zip(u, n)}


def main():
    out = {}
    slices = {}
    full = {}
    # ---- genomes -------------------------------------------------------------------------------------
    gsk = []
    for gi, f in enumerate(GENOMES):
        recs = O.read_fastx(T + f)
        b, off = O.concat([r[1] for r in recs])
        for mode, mname in ((O.MODE_SCALAR, "scalar"), (O.MODE_AVX2_COMPAT, "avx2")):
            g = O.sketch_genome(b, off, mode=mode)
            full[f"{f}:{mname}"] = dict(contigs=len(recs), gn_size=g["gn_size"], raw=g["n_raw_seeds"], dup=g["n_dup_kmers"],
                                        genome_kmers=xs(g["genome_kmers"]), tracked=xs(g["tracked"]),
                                        first3=[int(x) for x in g["genome_kmers"][:3]],
                                        first_contig_name=recs[0][0].decode())
            if mode == O.MODE_AVX2_COMPAT:
                gsk.append(g)
        # slices: 300 kb of contig 0 (+ 60 kb of contig 1 for the two-contig O157 file, with a short 50 bp contig between)
        parts = [recs[0][1][:300_000]]
        if len(recs) > 1:
            parts += [recs[0][1][300_000:300_050], recs[1][1][:60_000]]
        sb, soff = O.concat(parts)
        key = f"g{gi}"
        slices[key + "_bases"], slices[key + "_off"] = sb, soff
        for mode, mname in ((O.MODE_SCALAR, "scalar"), (O.MODE_AVX2_COMPAT, "avx2")):
            g = O.sketch_genome(sb, soff, mode=mode)
            slices[f"{key}_{mname}_kmers"], slices[f"{key}_{mname}_tracked"] = g["genome_kmers"], g["tracked"]
    np.savez_compressed(os.path.join(HERE, "ecoli_slices.npz"), **slices)

    # ---- reads -----------------------------------------------------------------------------------------
    reads = {}
    r1 = [r[1] for r in O.read_fastx(T + "k12_R1.fq")]
    r2 = [r[1] for r in O.read_fastx(T + "k12_R2.fq")]
    t1 = [r[1] for r in O.read_fastx(T + "t1.fq")]
    t2 = [r[1] for r in O.read_fastx(T + "t2.fq")]
    for name, rr in (("r1", r1), ("r2", r2), ("t1", t1), ("t2", t2)):
        reads[name + "_bases"], reads[name + "_off"] = O.concat(rr)

    def inter(a, b):
        return [x for p in zip(a, b) for x in p]

    cases = {
        "k12_single": (r1, False, False), "k12_single_nodedup": (r1, False, True),
        "k12_single_x2": (r1 + r1, False, False), "k12_single_x2_nodedup": (r1 + r1, False, True),
        "k12_single_x6": (r1 * 6, False, False),
        "k12_paired": (inter(r1, r2), True, False), "k12_paired_nodedup": (inter(r1, r2), True, True),
        "k12_paired_x2": (inter(r1 + r1, r2 + r2), True, False),
        "k12_paired_x2_nodedup": (inter(r1 + r1, r2 + r2), True, True),
        "t_paired": (inter(t1, t2), True, False), "t1_single": (t1, False, False), "t2_single": (t2, False, False),
    }
    samples = {}
    for cname, (rr, paired, nd) in cases.items():
        b, off = O.concat(rr)
        for mode, mname in ((O.MODE_SCALAR, "scalar"), (O.MODE_AVX2_COMPAT, "avx2")):
            s = O.sketch_reads(b, off, mode=mode, paired=paired, no_dedup=nd)
            reads[f"{cname}_{mname}_kmers"], reads[f"{cname}_{mname}_counts"] = s["kmers"], s["counts"]
            full[f"{cname}:{mname}"] = dict(distinct=int(len(s["kmers"])), total=int(s["counts"].sum()), hist=hist(s["counts"]),
                                            keys=xs(s["kmers"]), dup_removed=s["dup_removed"], mean_read_length=s["mean_read_length"])
            if mode == O.MODE_AVX2_COMPAT:
                samples[cname] = s
    np.savez_compressed(os.path.join(HERE, "k12_reads.npz"), **reads)

    # ---- containment + statistics on the full genomes ------------------------------------------------
    db = np.concatenate([g["genome_kmers"] for g in gsk])
    goff = np.zeros(4, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g["genome_kmers"]) for g in gsk])
    for sname in ("k12_single", "k12_paired", "t_paired"):
        s = samples[sname]
        cc, covs, _ = O.contain(s["kmers"], s["counts"], db, goff)
        for gi, f in enumerate(GENOMES):
            st = O.stats(covs[gi], int(goff[gi + 1] - goff[gi]))
            full[f"contain:{sname}:{f}"] = dict(
                contain_count=int(cc[gi]), n_kmers=int(goff[gi + 1] - goff[gi]), cov_hist=hist(covs[gi]) if len(covs[gi]) else {},
                naive_ani=st.naive_ani, final_est_ani=st.final_est_ani, final_est_cov=st.final_est_cov, lambda_status=st.lambda_status,
                lam=st.lambda_, median_cov=st.median_cov, mean_cov_geq1=st.mean_cov)
    np.savez_compressed(os.path.join(HERE, "ecoli_full_sketches.npz"), db=db, goff=goff,
                        tracked=np.concatenate([g["tracked"] for g in gsk]),
                        toff=np.concatenate([[0], np.cumsum([len(g["tracked"]) for g in gsk])]).astype(np.uint64))
    with open(os.path.join(HERE, "full_genome_kat.json"), "w") as fh:
        json.dump(full, fh, indent=1, sort_keys=True)
    print("wrote fixtures; sizes:", {f: os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE)})


if __name__ == "__main__":
    main()
