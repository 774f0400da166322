"""CPU tests: oracle/pyref.py — the second, independent restatement (pure Python ints, dict/set, scipy Poisson tail,
written from the Rust sources) — (a) pinned to the survey's known answers, (b) diffed against the C++ oracle on random
inputs with hypothesis: order-dependent dedup, markers, the AVX2 tail drop, single-end cut-off, mate-2 skip, genome
duplicate/spacing rules, containment, ratio_lambda ties, the statistics at 1e-12.
Two restatements agreeing is not reference parity (no sylph binary can be built here or on the GPU box), but a mistake
now has to be made twice, independently, in two languages to survive."""
import json
import math
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import oracle as O
from oracle import pyref as P

from .helpers import hist, xor_sum

SET = settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "survey_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def reads(golden_dir):
    return np.load(os.path.join(golden_dir, "k12_reads.npz"))


def recs(z, name):
    b, o = z[f"{name}_bases"], z[f"{name}_off"]
    return [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1)]


def table(counts):
    ks = np.array(sorted(counts), dtype=np.uint64)
    return ks, np.array([counts[int(k)] for k in ks], dtype=np.uint32)


# ------------------------------------------------------------------------------------------ (a) survey known answers
def test_pyref_hash_and_threshold_kats(kat):
    for k, v in kat["mm_hash64"].items():
        assert P.mm_hash64(int(k)) == int(v) == O.mm_hash64(int(k))
    for c, t in kat["threshold"].items():
        assert P.threshold(int(c)) == int(t)
    assert [P.BYTE_TO_SEQ[b] for b in range(256)] == [O.lib().orc_byte_to_seq(b) for b in range(256)]


def test_pyref_per_read_seeds_and_read_sketches(kat, reads):
    t1, t2, r1, r2 = recs(reads, "t1"), recs(reads, "t2"), recs(reads, "r1"), recs(reads, "r2")
    for name, rr in (("t1", t1), ("t2", t2)):
        for idx, exp in kat["per_read_seeds"][name].items():
            got = sorted(P.fmh_seeds_positions(rr[int(idx) - 1], 200, 31))
            assert got == sorted((a, b) for a, b in exp)
    cases = {"k12_single": P.sketch_sequences_needle(r1, 200, 31), "k12_paired": P.sketch_pair_sequences(r1, r2, 200, 31),
             "t_paired": P.sketch_pair_sequences(t1, t2, 200, 31), "t1_single": P.sketch_sequences_needle(t1, 200, 31),
             "t2_single": P.sketch_sequences_needle(t2, 200, 31)}
    for name, sk in cases.items():
        e = kat["reads"][name]
        ks, cs = table(sk["kmer_counts"])
        assert len(ks) == e["distinct"] and int(cs.sum()) == e["total"]
        assert hist(cs) == {int(a): b for a, b in e["hist"].items()}
        assert xor_sum(ks) == (e["keys_xor"], e["keys_sum"])
        if "mean_read_length" in e:
            assert sk["mean_read_length"] == e["mean_read_length"]
    d = kat["dedup"]
    s = P.sketch_sequences_needle(r1 + r1, 200, 31)
    assert (len(s["kmer_counts"]), sum(s["kmer_counts"].values()), s["dup_removed"]) == (512, 515, d["k12_single_x2"]["dup_removed"])
    s = P.sketch_sequences_needle(r1 + r1, 200, 31, no_dedup=True)
    assert hist(list(s["kmer_counts"].values())) == {int(a): b for a, b in d["k12_single_x2_nodedup"]["hist"].items()}
    assert P.sketch_sequences_needle(r1 * 6, 200, 31)["dup_removed"] == d["k12_single_x6"]["dup_removed"]
    s = P.sketch_pair_sequences(r1 + r1, r2 + r2, 200, 31)
    assert (len(s["kmer_counts"]), sum(s["kmer_counts"].values()), s["dup_removed"]) == (994, 1002, 1002)
    s = P.sketch_pair_sequences(r1 + r1, r2 + r2, 200, 31, no_dedup=True)
    assert hist(list(s["kmer_counts"].values())) == {int(a): b for a, b in d["k12_paired_x2_nodedup"]["hist"].items()}


def test_pyref_order_dependence_vector():
    """three occurrences of one k-mer with markers R1=(A1,B1), R2=(A1,B2), R3=(A3,B2), paired rule:
    order R1,R2,R3 -> count 1 (2 removed); order R1,R3,R2 -> count 2 (1 removed)."""
    R1, R2, R3 = ((1, 11), (2, 12)), ((1, 11), (3, 13)), ((4, 14), (3, 13))
    for order, exp in (((R1, R2, R3), (1, 2)), ((R1, R3, R2), (2, 1))):
        d = P._Dedup()
        for pr in order:
            d.add(77, pr, False, None)
        assert (d.counts[77], d.removed) == exp


def test_pyref_poisson_cap_and_toy_stats(kat):
    for med, cap in kat["poisson_cap"].items():
        m = float(med)
        admitted = [x for x in range(int(m), 200) if P.poisson_cdf(m, float(x)) < P.CUTOFF_PVALUE]
        assert max(admitted) == cap and admitted == list(range(int(m), cap + 1))
        assert [x for x in range(int(m), 200) if O.poisson_cdf(m, x) < P.CUTOFF_PVALUE] == admitted
    t = kat["toy_stats"]
    full = [int(v) for v, n in t["full_covs"].items() for _ in range(n)]
    assert P.ratio_lambda(full, 3.0) == t["ratio_lambda"]
    nz = [x for x in full if x]
    s = P.stats(len(nz), nz, len(full))
    assert s["final_est_ani"] == pytest.approx(t["adjusted_ani"], abs=1e-10)
    assert s["naive_ani"] == pytest.approx(t["naive_ani"], abs=1e-10)


def test_pyref_golden_slices_and_containment(golden_dir, kat):
    """The committed E. coli slices (oracle outputs) reproduced by pyref; containment + statistics of the k12 reads against
    the full committed genome sketches reproduce the survey's counts, histograms, lambda and ANI."""
    z = np.load(os.path.join(golden_dir, "ecoli_slices.npz"))
    b, off = z["g2_bases"], z["g2_off"]       # the two-contig O157 slice
    contigs = [bytes(b[int(off[i]):int(off[i + 1])][:60000]) for i in range(len(off) - 1)]
    for avx2, mode in ((False, O.MODE_SCALAR), (True, O.MODE_AVX2_COMPAT)):
        cb, co = O.concat(contigs)
        e = O.sketch_genome(cb, co, mode=mode)
        g = P.sketch_genome(contigs, 200, 31, avx2=avx2)
        assert g["genome_kmers"] == e["genome_kmers"].tolist() and g["tracked"] == e["tracked"].tolist()
        assert (g["gn_size"], g["n_raw_seeds"], g["n_dup_kmers"]) == (e["gn_size"], e["n_raw_seeds"], e["n_dup_kmers"])
    fz = np.load(os.path.join(golden_dir, "ecoli_full_sketches.npz"))
    rz = np.load(os.path.join(golden_dir, "k12_reads.npz"))
    r1, r2 = recs(rz, "r1"), recs(rz, "r2")
    for sample, sk in (("k12_single", P.sketch_sequences_needle(r1, 200, 31)), ("k12_paired", P.sketch_pair_sequences(r1, r2, 200, 31))):
        for gi, f in enumerate(("e.coli-EC590.fasta.gz", "e.coli-K12.fasta.gz", "e.coli-o157.fasta.gz")):
            gk = fz["db"][int(fz["goff"][gi]):int(fz["goff"][gi + 1])].tolist()
            cc, covs, _ = P.probe(gk, sk["kmer_counts"])
            exp = kat["containment"][sample][f]
            assert (cc, len(gk)) == (exp[0], exp[1]) and hist(covs) == {int(a): b for a, b in exp[2].items()}
            s = P.stats(cc, covs, len(gk))
            assert s["naive_ani"] == pytest.approx(exp[3], abs=1e-9)
            if sample == "k12_paired":
                ps = kat["paired_stats"][f]
                assert s["lambda_"] == pytest.approx(ps["lambda"], abs=1e-9)
                assert s["final_est_ani"] == pytest.approx(ps["ani"], abs=1e-9)
                assert s["mean_cov"] == pytest.approx(ps["mean_cov_geq1"], abs=1e-9) and s["median_cov"] == ps["median"]


# ------------------------------------------------------------------------------------------ (b) hypothesis: pyref vs C++ oracle
ALPHABETS = [b"ACGT", b"ACGTN", b"acgtACGTUu", bytes(range(256)), b"AC"]


@st.composite
def sequences(draw, max_len=400):
    alpha = draw(st.sampled_from(ALPHABETS))
    n = draw(st.one_of(st.integers(0, 70), st.integers(0, max_len)))
    seed = draw(st.integers(0, 2**32 - 1))
    rng = np.random.default_rng(seed)
    return bytes(rng.choice(np.frombuffer(alpha, dtype=np.uint8), size=n).astype(np.uint8))


@SET
@given(sequences(max_len=600), st.sampled_from([1, 2, 7, 200]), st.sampled_from([21, 31]))
def test_fuzz_seeds(seq, c, k):
    for avx2, mode in ((False, O.MODE_SCALAR), (True, O.MODE_AVX2_COMPAT)):
        assert P.extract_markers(seq, c, k, avx2) == O.extract_markers(seq, c=c, k=k, mode=mode).tolist()   # emission order too
        pp, hh = O.extract_markers_positions(seq, c=c, k=k, mode=mode)
        assert P.extract_markers_positions(seq, c, k, avx2) == list(zip(pp.tolist(), hh.tolist()))
    if O.lib().orc_has_avx2():   # the real-intrinsics variant of the oracle (the CPU-baseline path) agrees as a multiset
        assert sorted(P.extract_markers(seq, c, k, True)) == sorted(O.extract_markers(seq, c=c, k=k, mode=O.MODE_AVX2_FAST).tolist())


@SET
@given(sequences(max_len=500), sequences(max_len=100))
def test_fuzz_markers(s1, s2):
    a = P.pair_kmer_single(s1)
    assert (None if a is None else (a[0][0], a[0][1], a[1][0], a[1][1])) == O.pair_kmer_single(s1)
    b = P.pair_kmer(s1, s2)
    assert (None if b is None else (b[0][0], b[0][1], b[1][0], b[1][1])) == O.pair_kmer(s1, s2)


@st.composite
def read_sets(draw):
    """Reads cut from a tiny genome (so that k-mers recur and the order-dependent dedup has work to do), lengths around
    every threshold (33 / 66 / 400), exact duplicates, reads sharing only one of the two markers, homopolymers."""
    seed = draw(st.integers(0, 2**32 - 1))
    rng = np.random.default_rng(seed)
    glen = draw(st.sampled_from([120, 300, 900]))
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=glen).astype(np.uint8)
    n = draw(st.integers(0, 40))
    out = []
    for _ in range(n):
        kind = rng.integers(0, 10)
        if kind == 0 and out:
            out.append(out[int(rng.integers(0, len(out)))])                    # exact duplicate
        elif kind == 1 and out:                                                  # same first half, different second half
            src = out[int(rng.integers(0, len(out)))]
            tail = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=len(src) - len(src) // 2).astype(np.uint8).tobytes()
            out.append(src[:len(src) // 2] + tail)
        elif kind == 2:
            out.append(bytes([int(rng.choice(list(b"ACGT")))]) * int(rng.integers(0, 120)))
        else:
            L = int(rng.choice([0, 20, 31, 32, 33, 34, 35, 65, 66, 67, 100, 150, 399, 400, 401, 450]))
            L = min(L, glen)
            s = int(rng.integers(0, glen - L + 1))
            r = genome[s:s + L].copy()
            if L and rng.random() < 0.3:
                r[int(rng.integers(0, L))] = ord("N")
            out.append(r.tobytes())
    return out


@SET
@given(read_sets(), st.sampled_from([1, 3, 50]), st.booleans(), st.booleans())
def test_fuzz_read_sketch_single(recs_, c, no_dedup, avx2):
    mode = O.MODE_AVX2_COMPAT if avx2 else O.MODE_SCALAR
    b, off = O.concat(recs_)
    e = O.sketch_reads(b, off, c=c, mode=mode, no_dedup=no_dedup)
    g = P.sketch_sequences_needle(recs_, c, 31, no_dedup=no_dedup, avx2=avx2)
    ks, cs = table(g["kmer_counts"])
    assert ks.tolist() == e["kmers"].tolist() and cs.tolist() == e["counts"].tolist()
    assert g["dup_removed"] == e["dup_removed"]
    assert g["mean_read_length"] == e["mean_read_length"]     # same sequence of f64 operations: bit-equal


@SET
@given(read_sets(), st.sampled_from([1, 3, 50]), st.booleans(), st.booleans())
def test_fuzz_read_sketch_paired(recs_, c, no_dedup, avx2):
    mode = O.MODE_AVX2_COMPAT if avx2 else O.MODE_SCALAR
    if len(recs_) % 2:
        recs_ = recs_[:-1]
    b, off = O.concat(recs_)
    e = O.sketch_reads(b, off, c=c, mode=mode, paired=True, no_dedup=no_dedup)
    g = P.sketch_pair_sequences(recs_[0::2], recs_[1::2], c, 31, no_dedup=no_dedup, avx2=avx2)
    ks, cs = table(g["kmer_counts"])
    assert ks.tolist() == e["kmers"].tolist() and cs.tolist() == e["counts"].tolist()
    assert g["dup_removed"] == e["dup_removed"]
    assert g["mean_read_length"] == e["mean_read_length"]


@SET
@given(read_sets(), st.sampled_from([1, 3, 50]), st.booleans(), st.sampled_from([(1e-4, 10_000_000), (0.05, 2500), (0.3, 600), (0.3, 40)]))
def test_fuzz_read_sketch_paired_behind_the_filter(recs_, c, avx2, filt):
    """a10: the default pair dedup.  The C++ model of the filter (what csrc/a10.hip is held against) and this file's — written from the
    same description, with another eviction policy — must agree on every count: what a cuckoo filter answers does not depend on
    where an eviction left a fingerprint (the premise of a10.hip's formulation)."""
    fpr, cap = filt
    mode = O.MODE_AVX2_COMPAT if avx2 else O.MODE_SCALAR
    if len(recs_) % 2:
        recs_ = recs_[:-1]
    b, off = O.concat(recs_)
    e = O.sketch_reads_cuckoo_model(b, off, c=c, mode=mode, fpr=fpr, initial_capacity=cap)
    g = P.sketch_pair_sequences(recs_[0::2], recs_[1::2], c, 31, avx2=avx2, dedup_fpr=fpr, initial_capacity=cap)
    ks, cs = table(g["kmer_counts"])
    assert ks.tolist() == e["kmers"].tolist() and cs.tolist() == e["counts"].tolist()
    assert g["dup_removed"] == e["dup_removed"]


@st.composite
def genomes(draw):
    seed = draw(st.integers(0, 2**32 - 1))
    rng = np.random.default_rng(seed)
    n_contigs = draw(st.integers(0, 5))
    contigs = []
    for _ in range(n_contigs):
        L = int(rng.choice([0, 30, 61, 62, 63, 200, 1500, 4000]))
        contigs.append(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).astype(np.uint8))
    if len(contigs) >= 2 and len(contigs[0]) >= 1500 and len(contigs[-1]) >= 1500:
        contigs[-1][100:900] = contigs[0][300:1100]          # a repeat across contigs: genome-wide duplicate rule
    return [c.tobytes() for c in contigs]


@SET
@given(genomes(), st.sampled_from([2, 10, 50]), st.sampled_from([0, 5, 30]), st.booleans(), st.booleans())
def test_fuzz_genome_sketch(contigs, c, spacing, pseudotax, avx2):
    mode = O.MODE_AVX2_COMPAT if avx2 else O.MODE_SCALAR
    b, off = O.concat(contigs)
    e = O.sketch_genome(b, off, c=c, mode=mode, min_spacing=spacing, pseudotax=pseudotax)
    g = P.sketch_genome(contigs, c, 31, min_spacing=spacing, pseudotax=pseudotax, avx2=avx2)
    assert g["genome_kmers"] == e["genome_kmers"].tolist() and g["tracked"] == e["tracked"].tolist()
    assert (g["gn_size"], g["n_raw_seeds"], g["n_dup_kmers"]) == (e["gn_size"], e["n_raw_seeds"], e["n_dup_kmers"])
    if len(contigs) == 1:   # --individual-records flavour (sketch.rs:481-548) of a one-contig genome is the same sketch
        gi = P.sketch_genome_individual(contigs[0], c, 31, min_spacing=spacing, pseudotax=pseudotax, avx2=avx2)
        assert gi["genome_kmers"] == g["genome_kmers"] and gi["tracked"] == g["tracked"]


@SET
@given(st.integers(0, 2**32 - 1), st.sampled_from([0.0, 3.0, 50.0]))
def test_fuzz_containment(seed, min_kmers):
    rng = np.random.default_rng(seed)
    universe = rng.integers(0, 2**63, size=400, dtype=np.uint64)
    sample_k = np.unique(rng.choice(universe, size=int(rng.integers(0, 300))))
    sample_c = rng.choice([0, 1, 1, 2, 3, 9, 3_000_000_000], size=len(sample_k)).astype(np.uint32)
    genomes_ = [rng.choice(universe, size=int(rng.integers(0, 120))) for _ in range(int(rng.integers(1, 6)))]   # duplicates inside a genome allowed
    goff = np.zeros(len(genomes_) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g) for g in genomes_])
    dbk = np.concatenate(genomes_).astype(np.uint64) if genomes_ else np.zeros(0, np.uint64)
    cc, covs, _ = O.contain(sample_k, sample_c, dbk, goff, min_number_kmers=min_kmers)
    counts = {int(k): int(c) for k, c in zip(sample_k, sample_c)}
    for gi, g in enumerate(genomes_):
        r = P.probe(g.tolist(), counts, min_kmers)
        if r is None:
            assert cc[gi] == 0
        else:
            assert r[0] == cc[gi] and r[1] == covs[gi].tolist()


@SET
@given(st.integers(0, 2**32 - 1))
def test_fuzz_ratio_lambda_and_stats(seed):
    """Coverage vectors around every branch: few hits (< 25), single distinct value, ties for the mode (the reference
    breaks them by (count, value) descending, inference.rs:228-230), mode+1 absent, counts below min_count_correct,
    medians above 2 (HIGH) and above 15/30, outliers above the Poisson cap."""
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 6))
    n_hits = int(rng.choice([1, 5, 24, 25, 26, 60, 300]))
    if kind == 0:
        covs = rng.choice([1, 2], size=n_hits, p=[0.8, 0.2])
    elif kind == 1:
        covs = rng.choice([1, 2, 3, 4], size=n_hits)                    # near-ties for the mode
    elif kind == 2:
        covs = rng.poisson(float(rng.choice([0.3, 1.0, 2.5, 8.0, 20.0, 40.0])), size=n_hits) + 1
    elif kind == 3:
        covs = np.full(n_hits, int(rng.integers(1, 5)))                   # one distinct value
    elif kind == 4:
        covs = np.concatenate([rng.choice([1, 3], size=n_hits), [2] * int(rng.integers(0, 4))])   # mode+1 absent / rare
    else:
        covs = np.concatenate([rng.poisson(1.0, size=n_hits) + 1, rng.integers(50, 100000, size=3)])   # outliers
    covs = covs.astype(np.uint32)
    L = len(covs) + int(rng.integers(0, 3000))
    mcc = float(rng.choice([1.0, 3.0, 10.0]))
    e = O.stats(covs, L, min_count_correct=mcc)
    g = P.stats(len(covs), covs.tolist(), L, min_count_correct=mcc)
    assert g["lambda_status"] == {0: "LOW", 1: "HIGH", 2: "LAMBDA"}[e.lambda_status]
    for name in ("naive_ani", "final_est_ani", "final_est_cov", "mean_cov", "median_cov"):
        a, b = g[name], getattr(e, name)
        assert a == pytest.approx(b, rel=1e-12, abs=0), (name, a, b)
    assert g["n_full"] == e.n_full
    assert (g["max_cov"] == math.inf and e.max_cov > 1e300) or g["max_cov"] == e.max_cov
    if g["lambda_"] is not None:
        assert g["lambda_"] == pytest.approx(e.lambda_, rel=1e-12)
    full = [0] * (L - len(covs)) + [int(x) for x in np.sort(covs) if x <= g["max_cov"]]
    assert P.ratio_lambda(full, mcc) == O.ratio_lambda(np.array(full, dtype=np.uint32), mcc)
