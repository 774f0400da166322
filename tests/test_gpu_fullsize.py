"""Full-size parity (BASELINE configs[1], C2): 1 Gbp of synthetic 2x150 bp reads (3,333,334 pairs, duplicates, errors) sketched
on the GPU through every seeding / finishing flavour and compared — bit for bit — with the CPU oracle on the same bytes (the
oracle needs a few seconds for 1 Gbp), then profiled against sequence-backed genomes + decoy sketches, again against the
oracle.  Plus the size-independent properties: ascending distinct k-mers below the threshold, occurrences conserved,
batch-split invariance."""
import numpy as np
import pytest

import sylph_amd as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_one_gbp_sample_against_the_oracle(ctx):
    import torch
    import synth
    dev = torch.device("cuda", 0)
    c, k, n_pairs, read_len = 200, 31, 3_333_334, 150
    genomes = synth.random_genomes(24, 2_000_000, dev, 3, mutated_frac=0.0)
    bases, off = synth.paired_reads(genomes, n_pairs, seed=11)
    torch.cuda.synchronize()
    n_rec = 2 * n_pairs
    n_bases = n_rec * read_len                      # (the generator leaves a few bytes of slack behind the last read)
    assert int(off[-1].item()) == n_bases and int(bases.numel()) >= n_bases

    def sketch(finish, seeds, batches=1):
        ctx.set_option("finish", finish)
        ctx.set_option("seeds", seeds)
        try:
            sk = S.ReadSketcher(ctx, c=c, k=k, paired=True)
            step = ((n_rec // batches + 1) // 2) * 2
            keep = []
            for a in range(0, n_rec, step):
                z = min(n_rec, a + step)
                o = (off[a:z + 1] - off[a]).contiguous()
                keep.append(o)
                torch.cuda.synchronize()
                sk.push_device(bases.data_ptr() + a * read_len, o.data_ptr(), z - a, (z - a) * read_len)
            r = sk.finish()
            sk.close()
            return r
        finally:
            ctx.set_option("finish", "auto")
            ctx.set_option("seeds", "auto")

    g = sketch("auto", "auto")
    # size-independent properties
    thr = np.uint64((2**64 - 1) // c)
    assert np.all(np.diff(g["kmers"].astype(np.uint64)) > 0) and g["kmers"][-1] < thr and g["counts"].min() >= 1
    # the other flavours and a three-batch split of the same sample
    for finish, seeds, batches in (("generic", "slots", 1), ("auto", "unordered", 1), ("auto", "auto", 3)):
        o = sketch(finish, seeds, batches)
        assert np.array_equal(o["kmers"], g["kmers"]) and np.array_equal(o["counts"], g["counts"]), (finish, seeds, batches)
        assert o["dup_removed"] == g["dup_removed"]
    # the oracle on the same bytes
    hb = bases[:n_bases].cpu().numpy()
    ho = off.cpu().numpy().astype(np.uint64)
    e = O.sketch_reads(hb, ho, c=c, k=k, paired=True)
    assert np.array_equal(g["kmers"], e["kmers"]) and np.array_equal(g["counts"], e["counts"])
    assert g["dup_removed"] == e["dup_removed"] and e["dup_removed"] > 10000
    # containment: the 24 source genomes (sketched on the GPU in one batch) + 3,000 decoy sketches
    gb = genomes.reshape(-1).cpu().numpy()
    coff = np.arange(25, dtype=np.uint64) * np.uint64(2_000_000)
    km, koff, _, _ = ctx.sketch_genomes(gb, coff, np.arange(25, dtype=np.uint64), c=c, k=k)
    for i in (0, 23):
        eg = O.sketch_genome(gb[i * 2_000_000:(i + 1) * 2_000_000], np.array([0, 2_000_000], dtype=np.uint64), c=c, k=k)
        assert np.array_equal(km[int(koff[i]):int(koff[i + 1])], eg["genome_kmers"])
    dk, doff = synth.decoy_sketches(3000, c=c, device=dev, seed=5)
    dk, doff = dk.cpu().numpy().view(np.uint64), doff.cpu().numpy().astype(np.uint64)
    db_k = np.concatenate([km, dk])
    db_off = np.concatenate([koff, doff[1:] + koff[-1]])
    db = S.Database(ctx, db_k, db_off)
    cc, coff2, covs = db.contain_view(g["kmers"], g["counts"], packed=True)
    cc, coff2, covs = cc.copy(), coff2.copy(), covs.astype(np.uint32)
    db.close()
    ecc, ecov, _ = O.contain(g["kmers"], g["counts"], db_k, db_off, n_threads=8)
    assert np.array_equal(cc, ecc) and cc[:24].min() > 1000
    for gi in list(range(24)) + list(np.nonzero(ecc[24:])[0][:50] + 24):
        assert np.array_equal(covs[int(coff2[gi]):int(coff2[gi + 1])], np.sort(ecov[gi]))


def test_ragged_reads_with_n_against_the_oracle(ctx):
    """The c3r shape at size: 1.5 M pairs trimmed to 35-151 bp each (0.28 Gbp), 0.1 % N.  No two lanes of a wavefront walk the
    same number of k-mers (records are dealt to the lanes by length), blocks hold a varying number of records (some need a
    second pass), mates of 33 bp and more carry markers and shorter ones do not, N takes the exact ASCII path: bit for bit
    against the oracle, paired and single-end (single-end reads the same bytes as 3 M records with the cut-off at 4), through
    the read-per-lane kernel and the position kernel, in one push and in three."""
    import torch
    import synth
    dev = torch.device("cuda", 0)
    c, k, n_pairs = 200, 31, 1_500_000
    genomes = synth.random_genomes(12, 1_000_000, dev, 5, mutated_frac=0.0)
    bases, off = synth.ragged_paired_reads(genomes, n_pairs, min_len=31, seed=23)     # 31/32: below / at the seeding minimum
    torch.cuda.synchronize()
    n_rec = 2 * n_pairs
    n_bases = int(off[-1].item())
    hb = bases[:n_bases].cpu().numpy()
    ho = off.cpu().numpy().astype(np.uint64)
    assert (hb == ord("N")).sum() > 100_000

    def sketch(paired, seeds, batches=1):
        ctx.set_option("seeds", seeds)
        try:
            sk = S.ReadSketcher(ctx, c=c, k=k, paired=paired)
            step = ((n_rec // batches + 1) // 2) * 2
            keep = []
            for a in range(0, n_rec, step):
                z = min(n_rec, a + step)
                o = (off[a:z + 1] - off[a]).contiguous()
                keep.append(o)
                torch.cuda.synchronize()
                sk.push_device(bases.data_ptr() + int(off[a].item()), o.data_ptr(), z - a, int(o[-1].item()))
            r = sk.finish()
            sk.close()
            return r
        finally:
            ctx.set_option("seeds", "auto")

    for paired in (True, False):
        e = O.sketch_reads(hb, ho, c=c, k=k, paired=paired)
        assert e["dup_removed"] > 100
        for seeds, batches in (("auto", 1), ("slots", 1), ("auto", 3)):
            g = sketch(paired, seeds, batches)
            assert np.array_equal(g["kmers"], e["kmers"]) and np.array_equal(g["counts"], e["counts"]), (paired, seeds, batches)
            assert g["dup_removed"] == e["dup_removed"], (paired, seeds, batches)


def test_long_read_sample_in_two_pushes_at_c100_against_a_c200_database(ctx):
    """BASELINE configs[4] (C5) in small: ONT-like reads (log-normal lengths, N50 10 kb, 5 % errors as substitutions : insertions : deletions = 2 : 1 : 1 — SURVEY 8d; 1.2 Gbp — the oracle
    sketches that in seconds) sketched at c = 100 in TWO pushes of whole reads (a push holds < 2^32 bases: the real 5 Gbp sample
    needs two as well; the second push starts at an unaligned device address), through the position kernel (no record fits the
    read-per-lane kernel), compared bit for bit with the oracle; then profiled against a database sketched at c = 200 — reads may
    be denser than the database (contain.rs:562-568, :616-623): half of the sample's k-mers lie above the database's threshold
    and must simply find nothing."""
    import torch
    import synth
    dev = torch.device("cuda", 0)
    k = 31
    genomes = synth.random_genomes(12, 3_000_000, dev, 21, mutated_frac=0.0)
    bases, off = synth.long_reads(genomes, 1_200_000_000, seed=5)
    torch.cuda.synchronize()
    n_rec = off.numel() - 1
    n_bases = int(off[-1].item())
    cut = int(torch.searchsorted(off, off[-1] // 2).item())
    sk = S.ReadSketcher(ctx, c=100, k=k, paired=False)
    for a, z in ((0, cut), (cut, n_rec)):
        o = (off[a:z + 1] - off[a]).contiguous()
        torch.cuda.synchronize()
        sk.push_device(bases.data_ptr() + int(off[a].item()), o.data_ptr(), z - a, int(o[-1].item()))
    g = sk.finish()
    sk.close()
    hb = bases[:n_bases].cpu().numpy()
    ho = off.cpu().numpy().astype(np.uint64)
    e = O.sketch_reads(hb, ho, c=100, k=k, paired=False)
    assert np.array_equal(g["kmers"], e["kmers"]) and np.array_equal(g["counts"], e["counts"]) and g["dup_removed"] == e["dup_removed"] == 0
    thr200 = (2**64 - 1) // 200
    assert 0.4 < float((g["kmers"] >= thr200).mean()) < 0.6
    # database at c = 200: the source genomes + decoys
    gb = genomes.reshape(-1).cpu().numpy()
    coff = np.arange(13, dtype=np.uint64) * np.uint64(3_000_000)
    km, koff, _, _ = ctx.sketch_genomes(gb, coff, np.arange(13, dtype=np.uint64), c=200, k=k)
    dk, doff = synth.decoy_sketches(2000, c=200, device=dev, seed=9)
    dk, doff = dk.cpu().numpy().view(np.uint64), doff.cpu().numpy().astype(np.uint64)
    db_k = np.concatenate([km, dk])
    db_off = np.concatenate([koff, doff[1:] + koff[-1]])
    db = S.Database(ctx, db_k, db_off)
    cc, coff2, covs = db.contain_batch([(g["kmers"], g["counts"])])
    cc, coff2, covs = cc.copy(), coff2.copy(), covs.astype(np.uint32)
    db.close()
    ecc, ecov, _ = O.contain(g["kmers"], g["counts"], db_k, db_off, n_threads=8)
    assert np.array_equal(cc, ecc) and cc[:12].min() > 1000
    for gi in list(range(12)) + list(np.nonzero(ecc[12:])[0][:50] + 12):
        assert np.array_equal(covs[int(coff2[gi]):int(coff2[gi + 1])], np.sort(ecov[gi]))


def test_filter_dedup_at_its_real_capacity_with_growth(ctx):
    """a10 at full size: 1.5 Gbp of pairs = 12 M filter operations against the reference's own parameters (--fpr 1e-4, initial
    capacity 10^7, sketch.rs:796-804): the first filter fills up and a second one opens in mid-sample.  csrc/a10.hip (two phases:
    class table, count of the inserting operations, the cut, closed-filter lookups) against the oracle's walk of the same filter,
    whole table and duplicate count; in one push and in three."""
    import torch
    import synth
    dev = torch.device("cuda", 0)
    c, k, n_pairs, read_len = 200, 31, 5_000_000, 150
    genomes = synth.random_genomes(24, 2_000_000, dev, 3, mutated_frac=0.0)
    bases, off = synth.paired_reads(genomes, n_pairs, seed=13)
    torch.cuda.synchronize()
    n_rec = 2 * n_pairs
    n_bases = n_rec * read_len
    hb = bases[:n_bases].cpu().numpy()
    ho = off.cpu().numpy().astype(np.uint64)
    e = O.sketch_reads_cuckoo_model(hb, ho, c=c, k=k, fpr=1e-4)
    x = O.sketch_reads(hb, ho, c=c, k=k, paired=True)
    assert int(e["counts"].sum()) + e["dup_removed"] > 5_100_000            # more than 10^7 operations: the filter had to grow
    assert e["dup_removed"] >= x["dup_removed"]
    for batches in (1, 3):
        sk = S.ReadSketcher(ctx, c=c, k=k, paired=True, dedup_fpr=1e-4)
        step = ((n_rec // batches + 1) // 2) * 2
        keep = []
        for a in range(0, n_rec, step):
            z = min(n_rec, a + step)
            o = (off[a:z + 1] - off[a]).contiguous()
            keep.append(o)
            torch.cuda.synchronize()
            sk.push_device(bases.data_ptr() + a * read_len, o.data_ptr(), z - a, (z - a) * read_len)
        g = sk.finish()
        sk.close()
        assert np.array_equal(g["kmers"], e["kmers"]) and np.array_equal(g["counts"], e["counts"]), batches
        assert g["dup_removed"] == e["dup_removed"]
