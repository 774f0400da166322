"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, through the C ABI, against the CPU oracle on the
same inputs — bit-exact for hashes, seeds, (k-mer, count) tables, genome sketches, containment counts and coverage
multisets — plus the committed golden fixtures."""
import os

import numpy as np
import pytest

import sylph_amd as S
from oracle import oracle as O

from .helpers import ACGT, concat, random_seq, revcomp

pytestmark = pytest.mark.gpu

MODES = [(S.SEED_SCALAR, O.MODE_SCALAR), (S.SEED_AVX2_COMPAT, O.MODE_AVX2_COMPAT)]


# ---------------------------------------------------------------------------------------------- seeds
@pytest.mark.parametrize("k", [21, 31])
@pytest.mark.parametrize("c", [1, 7, 200])
def test_extract_markers_random(ctx, k, c):
    rng = np.random.default_rng(100 * k + c)
    for L in [0, 1, 20, 21, 22, 30, 31, 32, 33, 34, 35, 61, 62, 63, 150, 151, 1000, 16383, 16384, 16385, 16414, 16415,
              40000]:
        seq = random_seq(rng, L)
        for gm, om in MODES:
            got = ctx.extract_markers(seq, c=c, k=k, seed_mode=gm)
            exp = O.extract_markers(seq, c=c, k=k, mode=om)
            # order: ABI returns ascending start position; reference's lane-interleaved order is not observable
            pe, he = O.extract_markers_positions(seq, c=c, k=k, mode=om) if (om == O.MODE_SCALAR or L >= 2 * k) else (None, None)
            assert sorted(got.tolist()) == sorted(exp.tolist()), (L, k, c, gm)
            if pe is not None:
                order = np.argsort(pe, kind="stable")
                assert got.tolist() == he[order].tolist()


def test_extract_markers_alphabet_exact(ctx):
    """BYTE_TO_SEQ semantics for every byte value: N/IUPAC/gaps -> A, lower case, U/u, raw bytes 1,2,3 (types.rs:50-59)."""
    rng = np.random.default_rng(5)
    seq = rng.integers(0, 256, size=30000, dtype=np.uint8)
    for gm, om in MODES:
        got = ctx.extract_markers(seq, c=3, seed_mode=gm)
        exp = O.extract_markers(seq, c=3, mode=om)
        assert sorted(got.tolist()) == sorted(exp.tolist())
    for alphabet in (b"ACGTN", b"acgtn", b"ACGU", b"\x01\x02\x03A", b"AC-GT.RYKM"):
        seq = random_seq(rng, 5000, np.frombuffer(alphabet, dtype=np.uint8))
        got = ctx.extract_markers(seq, c=2)
        exp = O.extract_markers(seq, c=2)
        assert sorted(got.tolist()) == sorted(exp.tolist())


def test_extract_markers_rejects_bad_k(ctx):
    with pytest.raises(S.SylphHipError):
        ctx.extract_markers(b"ACGT" * 100, k=25)
    with pytest.raises(S.SylphHipError):
        ctx.extract_markers(b"ACGT" * 100, c=0)


def test_golden_genome_slices(ctx, golden_dir):
    z = np.load(os.path.join(golden_dir, "ecoli_slices.npz"))
    for gi in range(3):
        b, off = z[f"g{gi}_bases"], z[f"g{gi}_off"]
        for (gm, om), name in zip(MODES, ("scalar", "avx2")):
            g = ctx.sketch_genome(b, off, seed_mode=gm)
            assert np.array_equal(g["genome_kmers"], z[f"g{gi}_{name}_kmers"])
            assert np.array_equal(g["tracked"], z[f"g{gi}_{name}_tracked"])
            assert g["gn_size"] == int(off[-1])
    # survey known answers on the EC590 slice: first seeds (end_pos, hash)
    c, p, h = ctx.extract_markers_positions(z["g0_bases"][:2000], np.array([0, 2000], dtype=np.uint64))
    assert list(zip(p.tolist(), h.tolist()))[:5] == [(149, 4659887629048781), (183, 83502980970892378),
                                                     (186, 7394584420440650), (640, 54932856311185093),
                                                     (1114, 35186119693294796)]
    assert set(c.tolist()) == {0}


def test_genome_sketch_synthetic_multicontig(ctx):
    rng = np.random.default_rng(11)
    contigs = [random_seq(rng, n) for n in (5000, 0, 61, 62, 63, 30, 200000, 1, 77777)]
    # plant an exact repeat across contigs so the genome-wide duplicate rule (sketch.rs:594-605) fires
    contigs[8][1000:6000] = contigs[6][500:5500]
    b, off = concat(contigs)
    for gm, om in MODES:
        for c, spacing in ((50, 30), (200, 30), (10, 0), (10, 5)):
            g = ctx.sketch_genome(b, off, c=c, seed_mode=gm, min_spacing=spacing)
            e = O.sketch_genome(b, off, c=c, mode=om, min_spacing=spacing)
            assert e["n_dup_kmers"] > 0
            assert np.array_equal(g["genome_kmers"], e["genome_kmers"])
            assert np.array_equal(g["tracked"], e["tracked"])
            cc, pp, hh = ctx.extract_markers_positions(b, off, c=c, seed_mode=gm)
            exp = []
            for ci in range(len(contigs)):
                p, h = O.extract_markers_positions(contigs[ci], c=c, mode=om)
                exp += sorted((ci, int(a), int(x)) for a, x in zip(p, h))
            assert list(zip(cc.tolist(), pp.tolist(), hh.tolist())) == exp


def test_genome_batch_matches_per_genome_oracle(ctx):
    """sylph_sketch_genomes (database build on the device): every genome of a batch must come out exactly as
    sketch_genome (sketch.rs:550-622) gives it alone — duplicates are per genome even when other genomes of the batch share
    the k-mer, spacing restarts per genome/contig, empty genomes and contigs are legal."""
    import torch
    rng = np.random.default_rng(41)
    base = random_seq(rng, 120000)
    genomes = []
    genomes.append([base[:50000].copy(), base[60000:90000].copy()])
    g1 = [base[:50000].copy(), random_seq(rng, 20000)]            # shares 50 kb with genome 0 (cross-genome equal hashes)
    g1[1][2000:7000] = g1[0][10000:15000]                          # and repeats 5 kb inside itself (dup rule fires)
    genomes.append(g1)
    genomes.append([])                                             # a genome without contigs
    genomes.append([random_seq(rng, n) for n in (0, 30, 61, 62, 63, 5000)])
    g4 = [random_seq(rng, 40000)]
    g4[0][20000:30000] = g4[0][5000:15000]                         # tandem-ish repeat: both copies dropped entirely
    genomes.append(g4)
    genomes.append([base.copy()])                                  # superset of genome 0
    genomes.append([np.frombuffer(b"ACGTNNNNacgtuURYK" * 600, dtype=np.uint8).copy()])   # low complexity + non-ACGT
    contigs = [c for g in genomes for c in g]
    b, off = concat(contigs) if contigs else (np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    goff = np.zeros(len(genomes) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g) for g in genomes])
    dev = torch.from_numpy(np.concatenate([np.zeros(5, np.uint8), b, np.zeros(64, np.uint8)])).cuda()
    torch.cuda.synchronize()
    for gm, om in MODES:
        for c, spacing, pseudotax in ((200, 30, True), (50, 30, False), (7, 30, True), (7, 0, True), (20, 1000, True)):
            runs = [ctx.sketch_genomes(b, off, goff, c=c, seed_mode=gm, min_spacing=spacing, pseudotax=pseudotax),
                    ctx.sketch_genomes(None, off, goff, c=c, seed_mode=gm, min_spacing=spacing, pseudotax=pseudotax,
                                       device_ptr=dev.data_ptr() + 5)]     # device-resident, deliberately misaligned
            for km, koff, tr, toff in runs:
                assert koff[0] == 0 and toff[0] == 0 and len(km) == koff[-1] and len(tr) == toff[-1]
                any_dup = 0
                for gi, g in enumerate(genomes):
                    gb, go = concat(g) if g else (np.zeros(0, np.uint8), np.zeros(1, np.uint64))
                    e = O.sketch_genome(gb, go, c=c, mode=om, min_spacing=spacing, pseudotax=pseudotax)
                    any_dup += e["n_dup_kmers"]
                    assert np.array_equal(km[int(koff[gi]):int(koff[gi + 1])], e["genome_kmers"]), (gi, c, spacing)
                    if pseudotax:
                        assert np.array_equal(tr[int(toff[gi]):int(toff[gi + 1])], e["tracked"]), (gi, c, spacing)
                    else:
                        assert toff[gi + 1] == 0
                assert any_dup > 0
    # degenerate batches
    km, koff, tr, toff = ctx.sketch_genomes(np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(1, np.uint64))
    assert len(km) == 0 and list(koff) == [0]
    with pytest.raises(S.SylphHipError):
        ctx.sketch_genomes(b, off, goff[:-1])                       # genome offsets do not cover all contigs


# ---------------------------------------------------------------------------------------------- read sketches
FINISH_MODES = ("auto", "generic")   # bucket + in-LDS replay (with its fallback) and the device-wide sort path


def sketch_gpu(ctx, bases, off, paired=False, no_dedup=False, seed_mode=S.SEED_AVX2_COMPAT, c=200, k=31, batches=1, **dedup):
    """Sketches with BOTH finish paths and insists that they agree before returning the result."""
    res = []
    # both finish paths x the three seeding flavours (read-per-lane kernel, position kernel with ordered slots / unordered)
    for mode, seeds in (("auto", "auto"), ("generic", "unordered"), ("auto", "slots")):
        ctx.set_option("finish", mode)
        ctx.set_option("seeds", seeds)
        try:
            res.append(_sketch_gpu_once(ctx, bases, off, paired, no_dedup, seed_mode, c, k, batches, **dedup))
        finally:
            ctx.set_option("finish", "auto")
            ctx.set_option("seeds", "auto")
    for other in res[1:]:
        assert np.array_equal(res[0]["kmers"], other["kmers"]) and np.array_equal(res[0]["counts"], other["counts"])
        assert res[0]["dup_removed"] == other["dup_removed"]
    return res[0]


def _sketch_gpu_once(ctx, bases, off, paired, no_dedup, seed_mode, c, k, batches, **dedup):
    sk = S.ReadSketcher(ctx, c=c, k=k, paired=paired, no_dedup=no_dedup, seed_mode=seed_mode, **dedup)
    n = len(off) - 1
    step = max(2, ((n // batches + 1) // 2) * 2)
    for s in range(0, max(n, 1), step):
        e = min(n, s + step)
        lo, hi = int(off[s]), int(off[e])
        sk.push(bases[lo:hi], off[s:e + 1] - off[s])
    r = sk.finish()
    sk.close()
    return r


def assert_same_sketch(g, e, tag=None):
    assert np.array_equal(g["kmers"], e["kmers"]), tag
    assert np.array_equal(g["counts"], e["counts"]), tag
    assert g["dup_removed"] == e["dup_removed"], tag


def test_golden_read_sketches(ctx, golden_dir):
    z = np.load(os.path.join(golden_dir, "k12_reads.npz"))
    r1, o1, r2, o2 = z["r1_bases"], z["r1_off"], z["r2_bases"], z["r2_off"]

    def rec(b, o):
        return [b[int(o[i]):int(o[i + 1])] for i in range(len(o) - 1)]
    R1, R2 = rec(r1, o1), rec(r2, o2)
    T1, T2 = rec(z["t1_bases"], z["t1_off"]), rec(z["t2_bases"], z["t2_off"])
    inter = lambda a, b: [x for p in zip(a, b) for x in p]
    cases = {"k12_single": (R1, False, False), "k12_single_nodedup": (R1, False, True), "k12_single_x2": (R1 + R1, False, False),
             "k12_single_x2_nodedup": (R1 + R1, False, True), "k12_single_x6": (R1 * 6, False, False),
             "k12_paired": (inter(R1, R2), True, False), "k12_paired_nodedup": (inter(R1, R2), True, True),
             "k12_paired_x2": (inter(R1 + R1, R2 + R2), True, False),
             "k12_paired_x2_nodedup": (inter(R1 + R1, R2 + R2), True, True), "t_paired": (inter(T1, T2), True, False),
             "t1_single": (T1, False, False), "t2_single": (T2, False, False)}
    for name, (rr, paired, nd) in cases.items():
        b, off = concat(rr)
        for (gm, om), mname in zip(MODES, ("scalar", "avx2")):
            for batches in (1, 3):
                g = sketch_gpu(ctx, b, off, paired=paired, no_dedup=nd, seed_mode=gm, batches=batches)
                assert np.array_equal(g["kmers"], z[f"{name}_{mname}_kmers"]), (name, mname)
                assert np.array_equal(g["counts"], z[f"{name}_{mname}_counts"]), (name, mname)
    # survey dedup known answers (A.2)
    b, off = concat(R1 + R1)
    g = sketch_gpu(ctx, b, off)
    assert len(g["kmers"]) == 512 and int(g["counts"].sum()) == 515 and g["dup_removed"] == 515
    b, off = concat(R1 * 6)
    assert sketch_gpu(ctx, b, off)["dup_removed"] == 2575
    b, off = concat(inter(R1 + R1, R2 + R2))
    g = sketch_gpu(ctx, b, off, paired=True)
    assert len(g["kmers"]) == 994 and int(g["counts"].sum()) == 1002 and g["dup_removed"] == 1002


def make_reads(rng, genome, n, L, err=0.005, dup_frac=0.1, paired=False, insert=350, ragged=True):
    recs = []
    for _ in range(n):
        if paired:
            ins = int(rng.integers(max(L, insert - 60), insert + 60))
            s = int(rng.integers(0, len(genome) - ins))
            frag = genome[s:s + ins]
            l1 = L if not ragged else int(rng.integers(20, L + 1))
            l2 = L if not ragged else int(rng.integers(20, L + 1))
            m1, m2 = frag[:l1].copy(), revcomp(frag)[:l2].copy()
            for m in (m1, m2):
                e = rng.random(len(m)) < err
                m[e] = rng.choice(ACGT, size=int(e.sum()))
            recs.append((m1, m2))
        else:
            l = L if not ragged else int(rng.integers(10, L + 1))
            s = int(rng.integers(0, len(genome) - l))
            m = genome[s:s + l].copy()
            if rng.random() < 0.5:
                m = revcomp(m)
            e = rng.random(len(m)) < err
            m[e] = rng.choice(ACGT, size=int(e.sum()))
            recs.append(m)
    # exact duplicates (PCR) appended at random positions, exercises dup_removal_lsh_full_exact
    for _ in range(int(n * dup_frac)):
        recs.insert(int(rng.integers(0, len(recs))), recs[int(rng.integers(0, len(recs)))])
    if paired:
        return [x for p in recs for x in p]
    return recs


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("no_dedup", [False, True])
def test_read_sketch_synthetic(ctx, paired, no_dedup):
    rng = np.random.default_rng(42 + paired * 2 + no_dedup)
    genome = random_seq(rng, 30000)
    for c, n, L in ((20, 4000, 150), (5, 1500, 100), (200, 3000, 250)):
        recs = make_reads(rng, genome, n, L, paired=paired, dup_frac=0.3)
        b, off = concat(recs)
        for gm, om in MODES:
            e = O.sketch_reads(b, off, c=c, mode=om, paired=paired, no_dedup=no_dedup)
            for batches in (1, 4):
                g = sketch_gpu(ctx, b, off, paired=paired, no_dedup=no_dedup, seed_mode=gm, c=c, batches=batches)
                assert_same_sketch(g, e)
            if not no_dedup:
                assert e["dup_removed"] > 0


@pytest.mark.parametrize("fpr,capacity", [(1e-4, None), (0.02, None), (0.05, 2500), (0.3, 600)])
def test_read_sketch_paired_filter_dedup(ctx, fpr, capacity):
    """a10, the reference's DEFAULT for pairs (sketch.rs:733-769 over a scalable cuckoo filter): csrc/a10.hip against the oracle's model
    of the filter walked pair by pair — bit for bit, through both finish paths, every seeding flavour, one batch and several, with
    the filter growing (small capacities) and with false positives that really happen (large --fpr)."""
    rng = np.random.default_rng(77)
    genome = random_seq(rng, 30000)
    cap = {} if capacity is None else dict(dedup_capacity=capacity)
    ocap = {} if capacity is None else dict(initial_capacity=capacity)
    differs = 0
    for c, n, L in ((20, 4000, 150), (5, 1500, 100), (200, 3000, 250)):
        recs = make_reads(rng, genome, n, L, paired=True, dup_frac=0.3)
        b, off = concat(recs)
        for gm, om in MODES:
            e = O.sketch_reads_cuckoo_model(b, off, c=c, mode=om, fpr=fpr, **ocap)
            x = O.sketch_reads(b, off, c=c, mode=om, paired=True)
            differs += int(e["dup_removed"] != x["dup_removed"])
            for batches in (1, 4):
                for a10 in ("walk", "part"):     # the phase walk over a class table / the partitioned pass (one filter only: else it IS the walk)
                    g = sketch_gpu(ctx, b, off, paired=True, seed_mode=gm, c=c, batches=batches, dedup_fpr=fpr, a10=a10, **cap)
                    assert_same_sketch(g, e)
            assert e["dup_removed"] > 0
    if fpr >= 0.02:
        assert differs > 0          # the filter's false positives show in the result (else this test checks the exact path twice)
    # a capacity that would fill the filter's buckets to the brim is refused (insertions fail there; the answers turn order-dependent)
    with pytest.raises(S.SylphHipError):
        S.ReadSketcher(ctx, c=200, paired=True, dedup_fpr=fpr, dedup_capacity=4096)
    # --no-dedup and single-end sessions ignore the option (sketch.rs:744, :897)
    g = sketch_gpu(ctx, b, off, paired=True, no_dedup=True, c=200, dedup_fpr=fpr, **cap)
    assert_same_sketch(g, O.sketch_reads(b, off, c=200, paired=True, no_dedup=True))


def test_filter_dedup_seeds_repeated_inside_one_read(ctx):
    """The filter's answers depend on the ORDER of the walk (sketch.rs:806-867 hands a record's seeds over in the order
    extract_markers emitted them: lane-interleaved for the AVX2 routine), and so does `*c > 0` (:749): reads cut from a tandem
    repeat of period 40 carry the same k-mer two or three times, and the occurrence that comes first by position is often the one the
    walk sees last.  Exact set and filter, both seed modes, both passes, both finish paths."""
    rng = np.random.default_rng(4040)
    unit = random_seq(rng, 40)
    rep = np.tile(unit, 60)                                    # 2,400 bases of period 40
    genome = np.concatenate([random_seq(rng, 3000), rep, random_seq(rng, 3000)])
    recs = make_reads(rng, genome, 1500, 150, err=0.002, dup_frac=0.2, paired=True, insert=300, ragged=False)
    b, off = concat(recs)
    for c in (3, 11):
        for gm, om in MODES:
            e = O.sketch_reads_cuckoo_model(b, off, c=c, mode=om, fpr=1e-4)
            x = O.sketch_reads(b, off, c=c, mode=om, paired=True)
            assert e["counts"].max() > 20                       # the repeat's k-mers are deep, and repeated inside single reads
            for a10 in ("walk", "part"):
                assert_same_sketch(sketch_gpu(ctx, b, off, paired=True, seed_mode=gm, c=c, dedup_fpr=1e-4, a10=a10), e)
            assert_same_sketch(sketch_gpu(ctx, b, off, paired=True, seed_mode=gm, c=c), x)


def test_filter_dedup_partitioned_pass_and_its_verdict(ctx):
    """Round 5: where ONE filter takes every operation of the sample the marks come from the partitioned pass (csrc/a10.hip: operations
    sorted by class, no table), which cannot know by itself that one filter was enough: its verdict — a bucket with more copies of one
    class than a workgroup takes, more operations than the capacity — is read with the finish's tail block, and a bad one sends the
    sample through the phase walk.  Slots of a deferred device batch, slots of a checked push, dense arrays; every table against
    the model, the road read from the context's counters."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(555)
    genome = random_seq(rng, 400_000)

    def pairs(n, L=150):
        st = rng.integers(0, len(genome) - 500, size=n)
        recs = []
        for s in st:
            recs.append(genome[s:s + L])
            recs.append(revcomp(genome[s + 200:s + 200 + L]))
        return recs

    def run(recs_list, borrow=True, host=False, c=200, **dedup):
        sk = S.ReadSketcher(ctx, c=c, k=31, paired=True, dedup_fpr=1e-4, **dedup)
        if borrow:
            sk.set_option("borrow_until_finish", 1)
        keep = []
        for recs in recs_list:
            b, o = concat(recs)
            if host:
                sk.push(b, o)
            else:
                tb = torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).to(dev)
                to = torch.from_numpy(o.astype(np.int64)).to(dev)
                torch.cuda.synchronize()
                keep.append((tb, to))
                sk.push_device(tb.data_ptr(), to.data_ptr(), len(recs), int(o[-1]))
        g = sk.finish()
        sk.close()
        b, o = concat([r for recs in recs_list for r in recs])
        ocap = dict(initial_capacity=dedup["dedup_capacity"]) if "dedup_capacity" in dedup else {}
        assert_same_sketch(g, O.sketch_reads_cuckoo_model(b, o, c=c, mode=O.MODE_AVX2_COMPAT, fpr=1e-4, **ocap))
        return g

    def roads(fn):
        ctx.profile(True)
        fn()
        r = tuple(int(ctx.kernel_stats(f)[1]) for f in ("deferred", "deferred_redo", "a10_part", "a10_redo"))
        ctx.profile(False)
        return r

    normal = pairs(6000) + pairs(50) * 3                                   # ~1.8 Mbp, some exact duplicate pairs
    assert roads(lambda: run([normal])) == (1, 0, 1, 0)                    # deferred slots, partitioned pass, good verdict
    assert roads(lambda: run([normal], a10="walk")) == (1, 0, 0, 0)
    assert roads(lambda: run([normal], borrow=False)) == (0, 0, 1, 0)      # slots of a checked push
    assert roads(lambda: run([normal], host=True)) == (0, 0, 1, 0)         # dense arrays
    assert roads(lambda: run([normal, pairs(4000)])) == (1, 0, 1, 0)       # two batches: dense arrays by the time of the finish
    long_rec = pairs(3000) + [genome[1000:1600], revcomp(genome[1200:1350])] + pairs(3000)
    assert roads(lambda: run([long_rec])) == (1, 1, 2, 0)                  # the seeding verdict was bad: batch redone, marked again
    one = None
    for s in range(0, 5000, 7):                                            # a pair whose mate 1 carries >= 3 seeds at c = 200
        if len(O.extract_markers(genome[s:s + 150], c=200, k=31)) >= 3:
            one = [genome[s:s + 150], revcomp(genome[s + 200:s + 350])]
            break
    assert one is not None
    # 400 copies of one pair spread among ordinary ones: round 5's two-level pass (SYLPH_HIP_A10_LEVELS=2) finds a class bucket fuller than
    # a workgroup takes -> the walk; round 6's one-level pass cuts the range into slices and resolves them in place.  3,000 copies of
    # one item are more than a slice's lists take: the walk in both.
    two_levels = os.environ.get("SYLPH_HIP_A10_LEVELS") == "2"
    spread = []
    for _ in range(400):
        spread += pairs(9) + one
    assert roads(lambda: run([spread])) == ((1, 0, 1, 1) if two_levels else (1, 0, 1, 0))
    spread = []
    for _ in range(3000):
        spread += pairs(2) + one
    assert roads(lambda: run([spread])) == (1, 0, 1, 1)
    # capacities around the sample's number of operations: below the estimate the walk is chosen at once; between the estimate and
    # the true count the partitioned pass runs and its verdict sends the sample to the walk (which then grows a second filter)
    b, o = concat(normal)
    n_ops = 2 * sum(len(O.extract_markers(b[int(o[i]):int(o[i + 1])], c=200, k=31)) for i in range(len(o) - 1))
    seen = set()
    for cap in sorted({int(n_ops * f) for f in (0.55, 0.8, 0.9, 0.95, 0.99, 1.05, 1.2)}):
        nb = 1
        while nb * 4 < cap:
            nb *= 2
        if cap > 0.8 * 4 * nb:
            continue                                                       # (refused: would fill its buckets to the brim)
        seen.add(roads(lambda: run([normal], dedup_capacity=cap))[2:])
    assert (1, 0) in seen and ((1, 1) in seen or (0, 0) in seen)


def test_context_stream_priority_option():
    """Round 6: a context that owns its stream can have it made again at another priority (ctx option "stream_priority", csrc/capi.hip;
    the A/B of VERDICT r05 #5).  Results do not depend on it; a context on the caller's stream refuses."""
    import torch
    rng = np.random.default_rng(31)
    genome = random_seq(rng, 50_000)
    recs = [genome[s:s + 150] for s in rng.integers(0, len(genome) - 150, size=3000)]
    b, off = concat(recs)
    e = O.sketch_reads(b, off, c=50, mode=O.MODE_AVX2_COMPAT)
    own = S.Context(0)
    for prio in ("high", "low", "normal"):
        own.set_option("stream_priority", prio)
        assert_same_sketch(sketch_gpu(own, b, off, c=50), e)
    with pytest.raises(S.SylphHipError):
        own.set_option("stream_priority", "urgent")
    st = torch.cuda.Stream(device=torch.device("cuda", 0))
    borrowed = S.Context(0, stream=st.cuda_stream)
    with pytest.raises(S.SylphHipError):
        borrowed.set_option("stream_priority", "high")


def test_read_sketch_deep_coverage_and_cutoff(ctx):
    """Tiny genome, very deep coverage: long per-k-mer occurrence lists, single-end cut-off at 4 (sketch.rs:706,937),
    partial marker overlaps."""
    rng = np.random.default_rng(9)
    genome = random_seq(rng, 600)
    recs = make_reads(rng, genome, 6000, 120, err=0.0, dup_frac=0.5, ragged=True)
    b, off = concat(recs)
    for c in (3, 50):
        e = O.sketch_reads(b, off, c=c, mode=O.MODE_AVX2_COMPAT)
        assert_same_sketch(sketch_gpu(ctx, b, off, c=c), e)
    recs = make_reads(rng, genome, 3000, 100, err=0.0, dup_frac=0.5, paired=True, insert=200, ragged=False)
    b, off = concat(recs)
    e = O.sketch_reads(b, off, c=10, mode=O.MODE_AVX2_COMPAT, paired=True)
    assert_same_sketch(sketch_gpu(ctx, b, off, c=10, paired=True), e)


def test_read_sketch_long_reads_and_edge_lengths(ctx):
    rng = np.random.default_rng(13)
    genome = random_seq(rng, 200000)
    recs = []
    for L in (0, 1, 30, 31, 32, 33, 34, 35, 65, 66, 67, 399, 400, 401, 5000, 20000, 0, 70000):
        s = int(rng.integers(0, len(genome) - L)) if L else 0
        recs.append(genome[s:s + L].copy())
        recs.append(genome[s:s + L].copy())   # duplicate: dedup only applies for 66 <= L <= 400
    b, off = concat(recs)
    for gm, om in MODES:
        for c in (100, 4):
            e = O.sketch_reads(b, off, c=c, mode=om)
            assert_same_sketch(sketch_gpu(ctx, b, off, c=c, seed_mode=gm, batches=2), e)
    # homopolymer / doubled reads: both markers equal (self-hit rule)
    dbl = [np.repeat(random_seq(rng, 60), 2) for _ in range(6)]
    recs = [dbl[0], dbl[1], dbl[0], dbl[2], dbl[1], dbl[1]] + [np.full(100, ord("A"), dtype=np.uint8)] * 3
    b, off = concat(recs)
    e = O.sketch_reads(b, off, c=1, mode=O.MODE_SCALAR)
    assert_same_sketch(sketch_gpu(ctx, b, off, c=1, seed_mode=S.SEED_SCALAR), e)


def test_bucket_path_is_really_used(ctx):
    """finish=bucket forbids the fallback: an ordinary sample must go through the in-LDS replay; a sample with a k-mer of
    thousands of occurrences must overflow it (and then works through the fallback under finish=auto)."""
    rng = np.random.default_rng(17)
    genome = random_seq(rng, 300000)
    b, off = concat(make_reads(rng, genome, 20000, 150, dup_frac=0.1))
    e = O.sketch_reads(b, off, c=20)
    ctx.set_option("finish", "bucket")
    try:
        assert_same_sketch(_sketch_gpu_once(ctx, b, off, False, False, S.SEED_AVX2_COMPAT, 20, 31, 1), e)
        deep = concat([genome[:200]] * 3000)
        with pytest.raises(S.SylphHipError):
            _sketch_gpu_once(ctx, deep[0], deep[1], False, False, S.SEED_AVX2_COMPAT, 3, 31, 1)
    finally:
        ctx.set_option("finish", "auto")
    assert_same_sketch(_sketch_gpu_once(ctx, deep[0], deep[1], False, False, S.SEED_AVX2_COMPAT, 3, 31, 1),
                       O.sketch_reads(deep[0], deep[1], c=3))
    # an ordinary sample with a few k-mers at thousands of occurrences: only the buckets that hold them leave the in-LDS
    # replay (their occurrences go through the device-wide path as a small sample of their own), single-end and paired
    recs = make_reads(rng, genome, 20000, 150, dup_frac=0.1) + [genome[1000:1200].copy() for _ in range(2500)]
    order = rng.permutation(len(recs))
    mixed = concat([recs[i] for i in order])
    ctx.profile(True)
    try:
        for paired in (False, True):
            e = O.sketch_reads(mixed[0], mixed[1], c=20, paired=paired)
            assert e["counts"].max() > 500
            assert_same_sketch(_sketch_gpu_once(ctx, mixed[0], mixed[1], paired, False, S.SEED_AVX2_COMPAT, 20, 31, 1), e)
        assert ctx.kernel_stats("replay_overflow")[1] >= 1
    finally:
        ctx.profile(False)


def test_bucket_path_large_buckets(ctx):
    """k-mers with a few hundred occurrences each: their buckets exceed the 256-slot configuration of the in-LDS replay and
    must be picked up by its 1024-slot second launch — still without the device-wide fallback (finish=bucket)."""
    rng = np.random.default_rng(23)
    genome = random_seq(rng, 350)
    filler = random_seq(rng, 60000)
    ctx.set_option("finish", "bucket")
    try:
        recs = make_reads(rng, genome, 1000, 100, err=0.0, dup_frac=0.3, ragged=False) + make_reads(rng, filler, 6000, 100, dup_frac=0.1)
        order = rng.permutation(len(recs))
        b, off = concat([recs[i] for i in order])
        e = O.sketch_reads(b, off, c=10)
        assert 256 < e["counts"].max() < 600
        assert_same_sketch(_sketch_gpu_once(ctx, b, off, False, False, S.SEED_AVX2_COMPAT, 10, 31, 1), e)
        recs = make_reads(rng, genome, 600, 100, err=0.0, dup_frac=0.3, paired=True, insert=200, ragged=False) + \
            make_reads(rng, filler, 3000, 100, dup_frac=0.1, paired=True, insert=300, ragged=False)
        b, off = concat(recs)
        e = O.sketch_reads(b, off, c=10, paired=True)
        assert e["counts"].max() > 256
        assert_same_sketch(_sketch_gpu_once(ctx, b, off, True, False, S.SEED_AVX2_COMPAT, 10, 31, 1), e)
    finally:
        ctx.set_option("finish", "auto")


@pytest.mark.parametrize("depth", [70, 130, 300, 700, 1500])
def test_replay_configurations_by_kmer_depth(ctx, depth):
    """The three in-LDS configurations of the dedup/count stage and what lies beyond them, by k-mer depth: below 96 occurrences
    per k-mer the quadratic marker test of the 256-slot configuration, from 96 on the hash-table marker test of the 512- and
    1024-slot configurations, above 1024 the device-wide path.  Pairs with exact duplicates (PCR), pairs that share only one
    mate's start (partial marker overlap), mates on the same k-mer (mate-2 skip), reads that carry no markers, single-end with
    its cut-off at 4 — bucket-only finish up to depth 300 (no silent fall back to the device-wide path), in one push
    and in three, against the oracle."""
    rng = np.random.default_rng(depth)
    genome = random_seq(rng, 400)                  # ~370 k-mers, all at `depth`
    filler = random_seq(rng, 40000)
    n_pairs = depth * 400 // 160
    recs = make_reads(rng, genome, n_pairs, 100, err=0.002, dup_frac=0.2, paired=True, insert=180, ragged=False)
    # pairs that share mate 1 with an earlier pair but not mate 2 (one marker half equal), and short mates without markers
    for _ in range(n_pairs // 10):
        j = 2 * int(rng.integers(0, len(recs) // 2))
        s0 = int(rng.integers(0, len(genome) - 100))
        recs += [recs[j].copy(), revcomp(genome[s0:s0 + 100])]
    for _ in range(n_pairs // 10):
        s0 = int(rng.integers(0, len(genome) - 100))
        recs += [genome[s0:s0 + 100].copy(), genome[s0 + 5:s0 + 5 + 31].copy()]      # mate 2 of 31 bases: no markers, same k-mers
    recs += make_reads(rng, filler, 2000, 100, dup_frac=0.1, paired=True, insert=300, ragged=False)
    pairs = [(recs[i], recs[i + 1]) for i in range(0, len(recs), 2)]
    order = rng.permutation(len(pairs))
    b, off = concat([m for i in order for m in pairs[i]])
    ctx.set_option("finish", "bucket" if depth <= 300 else "auto")   # (deeper: some bucket passes 1024 occurrences)
    try:
        for paired in (True, False):
            e = O.sketch_reads(b, off, c=7, paired=paired)
            assert e["dup_removed"] > 50 and (paired or e["counts"].max() >= depth // 2)
            for batches in (1, 3):
                assert_same_sketch(_sketch_gpu_once(ctx, b, off, paired, False, S.SEED_AVX2_COMPAT, 7, 31, batches), e)
        # the same buckets with the filter's answers in place of the marker comparisons (a10: every configuration reads the bit a10.hip
        # left in the records, the device-wide path too); a small, leaky filter so that it grows and reports false positives
        ef = O.sketch_reads_cuckoo_model(b, off, c=7, fpr=0.05, initial_capacity=2500)
        assert ef["dup_removed"] != O.sketch_reads(b, off, c=7, paired=True)["dup_removed"]
        for batches in (1, 3):
            assert_same_sketch(_sketch_gpu_once(ctx, b, off, True, False, S.SEED_AVX2_COMPAT, 7, 31, batches, dedup_fpr=0.05, dedup_capacity=2500), ef)
    finally:
        ctx.set_option("finish", "auto")
    assert_same_sketch(sketch_gpu(ctx, b, off, paired=True, c=7), O.sketch_reads(b, off, c=7, paired=True))   # all three flavours agree


def test_deep_kmers_of_distinct_reads_are_not_quadratic(ctx):
    """100 k-mers with 6,000 occurrences each, every occurrence in a DIFFERENT read (random flanks => distinct markers): the
    buckets overflow the in-LDS replay and go through the device-wide path, whose marker test used to scan all earlier
    occurrences of the k-mer (quadratic: ~2e9 dependent loads here).  The sort-based formulation must agree with the oracle,
    single-end (cut-off at 4) and paired (no cut-off), within a time bound."""
    import time
    rng = np.random.default_rng(77)
    probe = random_seq(rng, 60000)
    pos, _ = O.extract_markers_positions(probe, c=200)
    cores = np.stack([probe[int(p) - 30:int(p) + 1] for p in pos[:100]])       # k-mers whose hash passes the c=200 threshold
    assert cores.shape == (100, 31)
    n = 100 * 6000
    reads = rng.choice(ACGT, size=(n, 150)).astype(np.uint8)
    reads[:, 60:91] = cores[rng.integers(0, 100, size=n)]
    reads[rng.integers(0, n, size=n // 50)] = reads[rng.integers(0, n, size=n // 50)]   # some exact duplicates too
    b = reads.reshape(-1)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)
    for paired in (False, True):
        e = O.sketch_reads(b, off, c=200, paired=paired)
        assert e["counts"].max() > 2500 and e["dup_removed"] > 0
        for finish in ("auto", "generic"):
            ctx.set_option("finish", finish)
            try:
                t0 = time.perf_counter()
                g = _sketch_gpu_once(ctx, b, off, paired, False, S.SEED_AVX2_COMPAT, 200, 31, 1)
                dt = time.perf_counter() - t0
            finally:
                ctx.set_option("finish", "auto")
            assert_same_sketch(g, e)
            assert dt < 5.0, (paired, finish, dt)   # ~0.1 s including the host->device copy of 90 MB


def test_tandem_repeats_overflow_the_tile_slots(ctx):
    """A short-period tandem repeat whose k-mer passes the threshold yields thousands of survivors per 16 KiB tile: the
    ordered K1 must detect the slot overflow and fall back, the unordered K1 must spill past its LDS stage."""
    rng = np.random.default_rng(4)
    for _ in range(200):
        unit = random_seq(rng, 13)
        rep = np.tile(unit, 6000)[:70000]
        if len(O.extract_markers(rep[:200], c=200)) > 0:
            break
    else:
        pytest.skip("no passing repeat unit found")
    recs = [rep, random_seq(rng, 5000), rep[:30000], rep[7:20000]]
    b, off = concat(recs)
    e = O.sketch_reads(b, off, c=200)
    assert e["counts"].max() > 1000
    assert_same_sketch(sketch_gpu(ctx, b, off, c=200), e)
    g = ctx.sketch_genome(b, off, c=200)
    eg = O.sketch_genome(b, off, c=200)
    assert np.array_equal(g["genome_kmers"], eg["genome_kmers"]) and np.array_equal(g["tracked"], eg["tracked"])


def test_low_complexity_reads_spill_locally(ctx):
    """Runs of low-complexity reads whose k-mer passes the threshold (thousands of survivors in one 16 KiB tile) inside an
    ordinary sample: the ordered K1 redoes only those tiles into spill regions (no whole-batch fallback); results equal the
    oracle's.  (ACC)n yields 40 survivors per 150 bp read at c = 20."""
    rng = np.random.default_rng(29)
    lowc = np.tile(np.frombuffer(b"ACC", dtype=np.uint8), 50)
    assert len(O.extract_markers(lowc, c=20)) == 40
    genome = random_seq(rng, 400000)
    recs = make_reads(rng, genome, 3000, 150, dup_frac=0.05, ragged=False)
    for pos in (200, 1700, 1701 + 60):
        for j in range(60):
            recs.insert(pos, np.roll(lowc, j % 3).copy())
    b, off = concat(recs)
    e = O.sketch_reads(b, off, c=20)
    assert e["dup_removed"] > 5000                     # 180 reads x 40 identical survivors, nearly all removed as duplicates
    ctx.profile(True)
    try:
        assert_same_sketch(_sketch_gpu_once(ctx, b, off, False, False, S.SEED_AVX2_COMPAT, 20, 31, 1), e)
        assert ctx.kernel_stats("seeds_spill")[1] == 1   # the overflowing tiles were redone locally ...
        assert ctx.kernel_stats("seeds")[1] == 2         # ... by the same kernel: no unordered fallback, no radix sort
    finally:
        ctx.profile(False)
    assert_same_sketch(sketch_gpu(ctx, b, off, c=20), e)
    recs2 = [r for r in recs for _ in (0, 1)]          # as mate pairs (mate 2 = copy of mate 1: the :852 skip rule fires)
    b, off = concat(recs2)
    e = O.sketch_reads(b, off, c=20, paired=True)
    assert_same_sketch(sketch_gpu(ctx, b, off, c=20, paired=True), e)


def test_read_sketch_k21(ctx):
    rng = np.random.default_rng(21)
    genome = random_seq(rng, 40000)
    for paired in (False, True):
        b, off = concat(make_reads(rng, genome, 2500, 120, paired=paired, dup_frac=0.2, insert=260))
        for gm, om in MODES:
            e = O.sketch_reads(b, off, c=30, k=21, mode=om, paired=paired)
            assert_same_sketch(sketch_gpu(ctx, b, off, c=30, k=21, paired=paired, seed_mode=gm), e)
    g = ctx.sketch_genome(genome, np.array([0, len(genome)], dtype=np.uint64), c=30, k=21)
    e = O.sketch_genome(genome, np.array([0, len(genome)], dtype=np.uint64), c=30, k=21)
    assert np.array_equal(g["genome_kmers"], e["genome_kmers"]) and np.array_equal(g["tracked"], e["tracked"])


def test_read_sketch_empty(ctx):
    sk = S.ReadSketcher(ctx)
    r = sk.finish()
    assert len(r["kmers"]) == 0 and r["dup_removed"] == 0
    sk.close()
    sk = S.ReadSketcher(ctx)
    sk.push(np.zeros(0, dtype=np.uint8), np.zeros(4, dtype=np.uint64))   # three empty records
    sk.push(np.frombuffer(b"ACGT", dtype=np.uint8), np.array([0, 4], dtype=np.uint64))
    r = sk.finish()
    assert len(r["kmers"]) == 0
    with pytest.raises(S.SylphHipError):
        sk.push(np.frombuffer(b"ACGT", dtype=np.uint8), np.array([0, 4], dtype=np.uint64))
    sk.close()
    sk = S.ReadSketcher(ctx, paired=True)
    with pytest.raises(S.SylphHipError):
        sk.push(np.frombuffer(b"ACGT", dtype=np.uint8), np.array([0, 4], dtype=np.uint64))   # odd record count
    sk.close()


# ---------------------------------------------------------------------------------------------- containment
def check_contain(ctx, db_kmers, goff, sk, sc, min_kmers=50.0):
    db = S.Database(ctx, db_kmers, goff)
    cc, off, covs = db.contain(sk, sc, min_number_kmers=min_kmers)
    vcc, voff, vcovs = db.contain_view(sk, sc, min_number_kmers=min_kmers)   # borrowed pinned views: same answer
    assert np.array_equal(vcc, cc) and np.array_equal(voff, off) and np.array_equal(vcovs, covs)
    pcc, poff, pcovs = db.contain_view(sk, sc, min_number_kmers=min_kmers, packed=True)   # narrowest width that fits
    big = int(covs.max()) if len(covs) else 0                    # the width follows the largest count among the HITS
    assert len(pcovs) == 0 or pcovs.dtype == np.uint32 or \
        pcovs.dtype == (np.uint8 if big < 256 else np.uint16 if big < 65536 else np.uint32)   # (64-bit hit keys report u32)
    assert np.array_equal(pcc, cc) and np.array_equal(poff, off) and np.array_equal(pcovs.astype(np.uint32), covs)
    db.close()
    ecc, ecov, _ = O.contain(sk, sc, db_kmers, goff, min_number_kmers=min_kmers)
    assert np.array_equal(cc, ecc)
    assert int(off[-1]) == int(ecc.sum()) and off[0] == 0
    assert np.array_equal(np.diff(off.astype(np.int64)), ecc.astype(np.int64))
    for g in range(len(goff) - 1):
        got = covs[int(off[g]):int(off[g + 1])]
        assert np.array_equal(got, np.sort(ecov[g])), g   # contain.rs:661: the reference sorts covs before use
    return cc


def test_contain_golden(ctx, golden_dir):
    z = np.load(os.path.join(golden_dir, "ecoli_full_sketches.npz"))
    r = np.load(os.path.join(golden_dir, "k12_reads.npz"))
    cc = check_contain(ctx, z["db"], z["goff"], r["k12_single_avx2_kmers"], r["k12_single_avx2_counts"])
    assert cc.tolist() == [200, 220, 131]
    cc = check_contain(ctx, z["db"], z["goff"], r["k12_paired_avx2_kmers"], r["k12_paired_avx2_counts"])
    assert cc.tolist() == [387, 439, 265]
    cc = check_contain(ctx, z["db"], z["goff"], r["t_paired_avx2_kmers"], r["t_paired_avx2_counts"])
    assert cc.tolist() == [0, 0, 0]


def test_contain_synthetic(ctx):
    rng = np.random.default_rng(21)
    thr = O.threshold(200)
    pool = np.unique(rng.integers(0, thr, size=200000, dtype=np.uint64))
    lens = [0, 10, 49, 50, 51, 300, 5000, 20000, 1, 12345, 0, 777]
    genomes = [rng.choice(pool, size=n, replace=False) for n in lens]
    genomes[5][:100] = genomes[6][:100]           # shared k-mers between genomes
    genomes[11] = np.concatenate([genomes[11][:700], genomes[11][:77]])   # duplicate k-mers inside one genome
    db = np.concatenate(genomes)
    goff = np.zeros(len(genomes) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g) for g in genomes])
    sk = np.sort(rng.choice(pool, size=60000, replace=False))
    sc = rng.integers(0, 40, size=len(sk)).astype(np.uint32)   # includes zero counts (contain.rs:634)
    sc[rng.random(len(sk)) < 0.01] = 3_000_000_000              # large counts survive the 32-bit packing
    for mk in (50.0, 0.0, 1000.5):
        check_contain(ctx, db, goff, sk, sc, min_kmers=mk)
    check_contain(ctx, db, goff, sk, np.minimum(sc, 200).astype(np.uint32))      # counts fit one byte
    check_contain(ctx, db, goff, sk, np.minimum(sc, 40000).astype(np.uint32))    # counts fit two bytes
    check_contain(ctx, db, goff, sk[:0], sc[:0])                # empty sample
    check_contain(ctx, db[:0], np.zeros(4, dtype=np.uint64), sk, sc)   # three empty genomes
    check_contain(ctx, db, goff, np.array([5, thr - 1, 2**64 - 1], dtype=np.uint64), np.array([1, 2, 3], dtype=np.uint32))


def crowded_db(rng, n_genomes=300):
    """A database with crowded index buckets: k-mers shared by 2 .. all genomes (overflow runs behind the 64-byte line), k-mers
    repeated inside a genome, neighbours that differ only in the low bits (same bucket, different remainder), the extreme
    values 0 and 2^64 - 1."""
    thr = O.threshold(200)
    pool = np.unique(rng.integers(0, thr, size=60000, dtype=np.uint64))
    shared = pool[:400]
    cluster = (pool[1000] + np.arange(64, dtype=np.uint64))           # 64 consecutive values: one or two buckets
    genomes = []
    for g in range(n_genomes):
        n = int(rng.integers(60, 400))
        own = rng.choice(pool[2000:], size=n, replace=False)
        sh = shared[:int(rng.integers(0, 400))] if g % 3 else shared[:8]
        parts = [own, sh, cluster[rng.random(64) < 0.5]]
        if g == 7:
            parts.append(np.array([0, 2**64 - 1, 2**64 - 1, thr - 1], dtype=np.uint64))
        if g == 9:
            parts.append(own[:30])                                    # repeated inside the genome: counted twice (contain.rs:632)
        genomes.append(np.concatenate(parts))
    return pool, shared, cluster, genomes


def flat_db(genomes):
    goff = np.zeros(len(genomes) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g) for g in genomes])
    return np.concatenate(genomes), goff


def test_contain_crowded_buckets_and_index_shapes(ctx):
    """The line index under every shape the build can take: 1..8 postings per bucket aimed for, one pass and several
    (index_pass_max lowered so that the filter -> sort -> lines passes really split the bucket range), overflow runs."""
    rng = np.random.default_rng(31)
    pool, shared, cluster, genomes = crowded_db(rng)
    db, goff = flat_db(genomes)
    sk = np.unique(np.concatenate([rng.choice(pool, size=20000, replace=False), shared[:300], cluster[::2],
                                   np.array([0, 2**64 - 1, 1], dtype=np.uint64)]))
    sc = rng.integers(0, 300, size=len(sk)).astype(np.uint32)
    try:
        for lam, pass_max in ((3, 1 << 30), (1, 1 << 30), (8, 1 << 30), (3, 5000), (2, 777)):
            ctx.set_option("index_lambda", str(lam))
            ctx.set_option("index_pass_max", str(pass_max))
            cc = check_contain(ctx, db, goff, sk, sc, min_kmers=0.0)
            assert cc.max() > 300            # the shared k-mers hit
            check_contain(ctx, db, goff, sk, sc, min_kmers=100.0)
    finally:
        ctx.set_option("index_lambda", "4")
        ctx.set_option("index_pass_max", str(1 << 30))


def test_contain_kmers_shared_by_thousands_of_genomes(ctx):
    """Strains of one species in an undereplicated database: 1,500 genomes hold (almost) the same k-mers, so the overflow run
    behind a bucket's line is 1,500 postings long and one sample k-mer yields 1,500 hits.  Such runs are walked by the whole
    wavefront (64 entries per step, matches counted, space taken once, hits written from the lanes): runs that end inside the
    first step, runs of many steps, several long runs per wavefront, neighbouring sample k-mers whose hits overflow the
    workgroup's stage inside one chunk (the rest goes straight to the hit array), genomes below min_number_kmers filtered inside
    the run, a batch of samples — counts and every coverage vector against the oracle."""
    rng = np.random.default_rng(77)
    thr = O.threshold(200)
    core = np.unique(rng.integers(0, thr, size=300, dtype=np.uint64))
    core = np.concatenate([core, core[10] + np.arange(1, 40, dtype=np.uint64)])      # 40 consecutive values: one bucket, many remainders
    genomes = []
    for g in range(1500):
        keep = rng.random(len(core)) < (0.97 if g % 50 else 0.1)                     # every 50th strain is a fragment (short genome)
        own = rng.integers(0, thr, size=int(rng.integers(0, 60)), dtype=np.uint64)
        genomes.append(np.concatenate([core[keep], own]))
    # a 60-strain and a 100-strain species: shorter runs (one or two steps)
    sp60 = rng.integers(0, thr, size=150, dtype=np.uint64)
    sp100 = rng.integers(0, thr, size=150, dtype=np.uint64)
    genomes += [np.concatenate([sp60[rng.random(150) < 0.9], rng.integers(0, thr, size=20, dtype=np.uint64)]) for _ in range(60)]
    genomes += [np.concatenate([sp100[rng.random(150) < 0.9], rng.integers(0, thr, size=20, dtype=np.uint64)]) for _ in range(100)]
    genomes += [rng.integers(0, thr, size=int(rng.integers(50, 400)), dtype=np.uint64) for _ in range(300)]
    db, goff = flat_db(genomes)
    sk = np.unique(np.concatenate([core, sp60[:100], sp100[::2], rng.integers(0, thr, size=5000, dtype=np.uint64)]))
    sc = rng.integers(0, 40, size=len(sk)).astype(np.uint32)
    for min_kmers in (0.0, 100.0):                                                    # 100: the fragment strains drop out inside the runs
        cc = check_contain(ctx, db, goff, sk, sc, min_kmers=min_kmers)
        assert cc[:1500].max() > 250 and int(cc.sum()) > 300_000
    # batch: the same table three times (one of them thinned) in one launch
    dbh = S.Database(ctx, db, goff)
    thin = rng.random(len(sk)) < 0.5
    tables = [(sk, sc), (sk[thin], sc[thin]), (sk, sc)]
    bcc, boff, bcovs = dbh.contain_batch(tables)
    G = len(genomes)
    for s_i, (k_i, c_i) in enumerate(tables):
        ecc, ecov, _ = O.contain(k_i, c_i, db, goff)
        assert np.array_equal(bcc[s_i * G:(s_i + 1) * G], ecc)
        for g in range(0, G, 13):
            r = s_i * G + g
            assert np.array_equal(np.asarray(bcovs[int(boff[r]):int(boff[r + 1])]).astype(np.uint32), np.sort(ecov[g]))
    dbh.close()


def test_contain_batch_matches_single_samples(ctx):
    """sylph_db_contain_batch: S tables in one probe launch / sort / copy — row s * G + g must equal what the single-sample
    call gives for sample s (and the oracle), with empty tables, zero counts and wide counts in the batch."""
    import torch
    rng = np.random.default_rng(32)
    pool, shared, cluster, genomes = crowded_db(rng, n_genomes=120)
    db, goff = flat_db(genomes)
    G = len(genomes)
    samples = []
    for s in range(7):
        n = [5000, 0, 12000, 1, 3000, 0, 800][s]
        k = np.unique(np.concatenate([rng.choice(pool, size=n, replace=False), shared[:50 * s]])) if n else np.zeros(0, np.uint64)
        c = rng.integers(0, 50, size=len(k)).astype(np.uint32)
        if s == 4:
            c[::7] = 70000                                             # forces 4-byte coverage values for the whole batch
        samples.append((k, c))
    d = S.Database(ctx, db, goff)
    for subset in (samples, samples[:1], samples[1:2], samples[:4], []):
        for mk in (50.0, 0.0):
            cc, off, covs = d.contain_batch(subset, min_number_kmers=mk)
            cc, off, covs = cc.copy(), off.copy(), covs.astype(np.uint32)
            assert len(cc) == len(subset) * G and len(off) == len(subset) * G + 1
            for s, (k, c) in enumerate(subset):
                ecc, ecov, _ = O.contain(k, c, db, goff, min_number_kmers=mk)
                assert np.array_equal(cc[s * G:(s + 1) * G], ecc), s
                for g in range(G):
                    assert np.array_equal(covs[int(off[s * G + g]):int(off[s * G + g + 1])], np.sort(ecov[g])), (s, g)
    # device-resident tables (what sylph_sketch_finish_device hands over)
    tk = [torch.from_numpy(k.view(np.int64)).cuda() for k, _ in samples]
    tc = [torch.from_numpy(c.view(np.int32)).cuda() for _, c in samples]
    torch.cuda.synchronize()
    cc2, off2, covs2 = d.contain_batch([(a.data_ptr() if a.numel() else 0, b.data_ptr() if b.numel() else 0, a.numel()) for a, b in zip(tk, tc)],
                                       device_ptrs=True)
    cc, off, covs = d.contain_batch(samples)
    assert np.array_equal(cc2, cc) and np.array_equal(off2, off) and np.array_equal(covs2, covs)
    d.close()


def test_sharded_database_one_rank_over_rccl(ctx):
    """sylph_db_contain_batch_sharded with a real RCCL communicator (world size 1: the only one a 1-GPU box can form — RCCL
    refuses two ranks on one device): librccl is resolved with dlopen, ncclCommInitRank / ncclAllGather / the grouped
    send-recv path execute on the GPU, and the sharded entry point must return exactly what the unsharded batch call returns."""
    rng = np.random.default_rng(33)
    pool, shared, cluster, genomes = crowded_db(rng, n_genomes=90)
    db, goff = flat_db(genomes)
    samples = [(np.sort(rng.choice(pool, size=n, replace=False)), rng.integers(0, 9, size=n).astype(np.uint32)) for n in (4000, 0, 9000)]
    comm = S.Comm(0, 1, ctx=ctx, rccl_id=S.Comm.rccl_unique_id())
    bounds = S.shard_bounds(int(db.max()), 1)
    d1 = S.Database(ctx, db, goff, shard=(bounds, 1, 0))
    d0 = S.Database(ctx, db, goff)
    for subset in (samples, samples[:1], []):
        a = d1.contain_batch_sharded(comm, subset)
        a = [x.copy() for x in a]
        b = d0.contain_batch(subset)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    with pytest.raises(S.SylphHipError):
        d0.contain_batch_sharded(comm, samples)            # not a shard
    # round 5: the one-rank "genome shard" (every genome) through the same call, the same RCCL communicator
    from sylph_amd import shard as SH
    gb = SH.genome_shard_bounds(goff, 1)
    assert list(gb) == [0, len(goff) - 1]
    d2 = S.Database(ctx, db, goff, genome_shard=(gb, 1, 0))
    for subset in (samples, samples[1:]):
        a = [x.copy() for x in d2.contain_batch_sharded(comm, subset)]
        assert all(np.array_equal(x, y) for x, y in zip(a, d0.contain_batch(subset)))
    gb3 = SH.genome_shard_bounds(goff, 3)
    per = [int(goff[int(gb3[i + 1])] - goff[int(gb3[i])]) for i in range(3)]
    assert gb3[0] == 0 and gb3[3] == len(goff) - 1 and max(per) - min(per) <= 2 * int(np.diff(goff.astype(np.int64)).max())
    d2.close()
    d1.close(); d0.close(); comm.close()


def test_sharded_database_through_torch_distributed_rccl():
    """The same call with the collectives served by torch.distributed's nccl (= RCCL) process group on the device buffers, and
    with the library's communicator bootstrapped through that process group — the two ways bench.py --gpus N forms its
    communicator (tests/rccl_torch_worker.py, one rank: all a 1-GPU box can form)."""
    import subprocess
    import sys
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(here, "rccl_torch_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_TORCH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---------------------------------------------------------------------------------------------- end to end
def test_end_to_end_sketch_then_contain_device_resident(ctx):
    """Reads -> device-resident table -> containment without a host round trip; checked against the oracle end to end,
    including the host statistics the reference derives from these integers (contain.rs:657-813)."""
    import torch
    rng = np.random.default_rng(77)
    genomes = [random_seq(rng, 60000) for _ in range(6)]
    genomes[3] = genomes[0].copy()
    mut = rng.random(len(genomes[3])) < 0.02
    genomes[3][mut] = rng.choice(ACGT, size=int(mut.sum()))
    gk, lens = [], []
    for g in genomes:
        r = ctx.sketch_genome(g, np.array([0, len(g)], dtype=np.uint64), c=50)
        e = O.sketch_genome(g, np.array([0, len(g)], dtype=np.uint64), c=50)
        assert np.array_equal(r["genome_kmers"], e["genome_kmers"])
        gk.append(r["genome_kmers"])
        lens.append(len(r["genome_kmers"]))
    db_k = np.concatenate(gk)
    goff = np.zeros(len(gk) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum(lens)
    recs = make_reads(rng, genomes[0], 3000, 150, paired=True, dup_frac=0.05, ragged=False) + \
        make_reads(rng, genomes[1], 800, 150, paired=True, dup_frac=0.0, ragged=False)
    b, off = concat(recs)
    tb = torch.from_numpy(np.concatenate([b, np.zeros(64, dtype=np.uint8)])).cuda()
    toff = torch.from_numpy(off.astype(np.int64)).cuda()
    sk = S.ReadSketcher(ctx, c=50, paired=True)
    sk.push_device(tb.data_ptr(), toff.data_ptr(), len(off) - 1)
    dk, dc, n, dup = sk.finish_device()
    e = O.sketch_reads(b, off, c=50, paired=True)
    assert n == len(e["kmers"]) and dup == e["dup_removed"]
    db = S.Database(ctx, db_k, goff)
    cc, coff, covs = db.contain(dk, dc, device_ptrs=True, n=n)
    ecc, ecov, _ = O.contain(e["kmers"], e["counts"], db_k, goff)
    assert np.array_equal(cc, ecc)
    for g in range(len(gk)):
        mine = covs[int(coff[g]):int(coff[g + 1])]
        assert np.array_equal(mine, np.sort(ecov[g]))
        a, bb = O.stats(mine, lens[g]), O.stats(ecov[g], lens[g])
        for f in ("naive_ani", "final_est_ani", "final_est_cov", "mean_cov", "median_cov"):
            assert abs(getattr(a, f) - getattr(bb, f)) <= 1e-6   # north_star float tolerance
    assert cc[0] > cc[3] > cc[2]
    db.close()
    sk.close()


def test_device_push_unaligned_batches(ctx):
    """Device-resident batches that start at arbitrary byte offsets of one big buffer (second batch of a sample)."""
    import torch
    rng = np.random.default_rng(31)
    genome = random_seq(rng, 50000)
    recs = make_reads(rng, genome, 3000, 137, dup_frac=0.2, ragged=True)
    b, off = concat(recs)
    e = O.sketch_reads(b, off, c=20)
    tb = torch.from_numpy(np.concatenate([np.zeros(3, dtype=np.uint8), b, np.zeros(80, dtype=np.uint8)])).cuda()   # +3: misaligned
    sk = S.ReadSketcher(ctx, c=20)
    n = len(off) - 1
    cuts = [0, 1, 700, 701, 1999, n]
    keep = []
    for a, z in zip(cuts[:-1], cuts[1:]):
        o = torch.from_numpy((off[a:z + 1] - off[a]).astype(np.int64)).cuda()
        keep.append(o)
        torch.cuda.synchronize()
        sk.push_device(tb.data_ptr() + 3 + int(off[a]), o.data_ptr(), z - a)
    g = sk.finish()
    sk.close()
    assert_same_sketch(g, e)


def test_push_from_pinned_host_memory(ctx):
    """SYLPH_MEM_HOST_PINNED (host feed): batches parsed in place into page-locked buffers, reused between
    pushes, give the same table as ordinary host memory."""
    rng = np.random.default_rng(37)
    genome = random_seq(rng, 80000)
    recs = make_reads(rng, genome, 4000, 150, dup_frac=0.2, paired=True, ragged=True)
    b, off = concat(recs)
    e = O.sketch_reads(b, off, c=20, paired=True)
    pb, po = S.PinnedBuffer(1 << 20), S.PinnedBuffer(8 * 4001)
    sk = S.ReadSketcher(ctx, c=20, paired=True)
    n = len(off) - 1
    for s0 in range(0, n, 2000):                     # 4 batches through the same two pinned buffers
        s1 = min(n, s0 + 2000)
        lo, hi = int(off[s0]), int(off[s1])
        pb.array[:hi - lo] = b[lo:hi]
        ov = po.array[:8 * (s1 - s0 + 1)].view(np.uint64)
        ov[:] = off[s0:s1 + 1] - off[s0]
        sk.push_pinned(pb, hi - lo, ov)
    g = sk.finish()
    sk.close()
    pb.close(); po.close()
    assert_same_sketch(g, e)


# ---------------------------------------------------------------------------------------------- multi-process
def test_packed_input_and_chunked_host_pipeline(ctx):
    """sylph_sketch_push_enc: SYLPH_ENC_2BIT input (host-packed with sylph_pack_2bit) must give the same sketch as the ASCII
    bytes it was packed from, from pageable, page-locked and device memory; host batches are cut into chunks that travel on the
    copy stream (push_chunk_bytes lowered so that a small batch has dozens of chunks, chunk starts at every phase inside a
    packed byte, pairs never split); long records send a packed batch through the unpack + position-kernel path."""
    import torch
    from sylph_amd.binding import ENC_2BIT, ENC_ASCII, MEM_DEVICE, MEM_HOST, MEM_HOST_PINNED
    rng = np.random.default_rng(55)
    genome = random_seq(rng, 80000)
    short_single = make_reads(rng, genome, 3000, 151, dup_frac=0.2, ragged=True)
    for r in short_single[::17]:
        if len(r) > 5:
            r[int(rng.integers(0, len(r)))] = ord("N")
    short_paired = make_reads(rng, genome, 2500, 150, paired=True, dup_frac=0.2, ragged=True)
    with_long = short_single[:500] + [genome[1000:9000].copy(), genome[20000:20401].copy()] + short_single[500:900]
    cases = [(short_single, False, 20), (short_paired, True, 20), (with_long, False, 50), ([], False, 20),
             ([genome[:50].copy(), np.zeros(0, np.uint8), genome[100:131].copy()], False, 3)]
    try:
        for recs, paired, c in cases:
            b, off = concat(recs)
            e = O.sketch_reads(b, off, c=c, paired=paired)
            packed = S.pack_2bit(b)
            n_bases, n_rec = int(off[-1]), len(off) - 1
            pin_a, pin_p, pin_o = S.PinnedBuffer(len(b) + 64), S.PinnedBuffer(len(packed) + 64), S.PinnedBuffer(len(off) * 8)
            pin_a.array[:len(b)] = b
            pin_p.array[:len(packed)] = packed
            pin_o.array.view(np.uint64)[:len(off)] = off
            dev_a = torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).cuda()
            dev_p = torch.from_numpy(packed).cuda()
            dev_o = torch.from_numpy(off.astype(np.int64)).cuda()
            torch.cuda.synchronize()
            for chunk in (str(64 << 20), "4099", "1000"):
                ctx.set_option("push_chunk_bytes", chunk)
                for enc, host, pin, dev in ((ENC_ASCII, b, pin_a, dev_a), (ENC_2BIT, packed, pin_p, dev_p)):
                    for mem in (MEM_HOST, MEM_HOST_PINNED, MEM_DEVICE):
                        if mem == MEM_DEVICE and chunk != "1000":
                            continue
                        sk = S.ReadSketcher(ctx, c=c, paired=paired)
                        if n_rec:
                            if mem == MEM_HOST:
                                sk.push_enc(host, off, n_bases, mem, enc)
                            elif mem == MEM_HOST_PINNED:
                                sk.push_enc(pin.ptr, pin_o.ptr, n_bases, mem, enc, n_records=n_rec)
                            else:
                                sk.push_enc(dev.data_ptr(), dev_o.data_ptr(), n_bases, mem, enc, n_records=n_rec)
                        g = sk.finish()
                        sk.close()
                        assert_same_sketch(g, e)
            for p_ in (pin_a, pin_p, pin_o):
                p_.close()
    finally:
        ctx.set_option("push_chunk_bytes", str(64 << 20))


@pytest.mark.parametrize("shard_by", ["kmer", "genome", "genome-allgather"])
def test_sharded_containment_two_ranks_one_gpu(shard_by):
    """Two ranks (gloo rendezvous, both on cuda:0) run the library's sharded exchange (csrc/shard.hip) with the collectives routed
    through torch.distributed callbacks, each with its own shard resident on the GPU — cut by k-mer range, or (round 5) by genome:
    sylph_db_upload_genome_shard, every table probed whole by every rank — and check their own samples against the oracle over the
    whole database; a rank that fails between the collectives takes every rank out of the call.  "genome-allgather" (round 6): north_star's
    literal shape — whole genomes per rank and the answers reduced by ONE all-gather (ctx option "shard_reduce" = "allgather")."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", {"kmer": "29533", "genome": "29537"}.get(shard_by, "29541"), os.path.join(root, "tests", "dist_gpu_worker.py")]
    env = dict(os.environ, SYLPH_TEST_SHARD_BY=shard_by.split("-")[0], SYLPH_TEST_SHARD_REDUCE="allgather" if shard_by.endswith("allgather") else "alltoall")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "DIST_GPU_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_reassign_on_device_matches_reference_semantics(ctx):
    """sylph_db_reassign_view vs a direct restatement of winner_table (contain.rs:410-430) + the winner pass of get_stats
    (contain.rs:637-646): shared k-mers, tracked k-mers that steal ownership, ANI ties (first in the passing list wins),
    non-passing genomes, duplicate postings."""
    rng = np.random.default_rng(99)
    thr = O.threshold(200)
    pool = np.unique(rng.integers(0, thr, size=40000, dtype=np.uint64))
    G = 14
    genomes = [rng.choice(pool, size=int(n), replace=False) for n in rng.integers(300, 3000, size=G)]
    tracked = [rng.choice(pool, size=int(n), replace=False) for n in rng.integers(0, 400, size=G)]
    genomes[2][:500] = genomes[1][:500]                 # shared genome_kmers
    genomes[5][:300] = genomes[1][200:500]
    tracked[7][:100] = genomes[1][:100]                 # a tracked k-mer set that can steal from genome 1
    tracked[3] = np.zeros(0, dtype=np.uint64)
    genomes[9] = np.concatenate([genomes[9], genomes[9][:50]])   # duplicate postings inside one genome
    db_k = np.concatenate(genomes)
    goff = np.zeros(G + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g) for g in genomes])
    t_k = np.concatenate(tracked)
    toff = np.zeros(G + 1, dtype=np.uint64)
    toff[1:] = np.cumsum([len(t) for t in tracked])
    sk = np.sort(rng.choice(pool, size=25000, replace=False))
    sc = rng.integers(0, 6, size=len(sk)).astype(np.uint32)
    db = S.Database(ctx, db_k, goff)
    db.attach_tracked(t_k, toff)
    smap = dict(zip(sk.tolist(), sc.tolist()))
    for trial in range(4):
        passing = rng.permutation(G)[: int(rng.integers(1, G + 1))]
        ani = rng.choice([0.95, 0.96, 0.97, 0.97, 0.99], size=len(passing))          # plenty of ties
        winner = {}
        for r, g in enumerate(passing):
            for km in list(genomes[g].tolist()) + list(tracked[g].tolist()):
                if km not in winner or ani[r] > winner[km][0]:
                    winner[km] = (ani[r], int(g))
        cc, off, covs, lost = db.reassign_view(sk, sc, passing, ani)
        for g in range(G):
            got = covs[int(off[g]):int(off[g + 1])]
            if g not in passing:
                assert cc[g] == 0 and len(got) == 0 and lost[g] == 0
                continue
            exp, exp_lost = [], 0
            for km in genomes[g].tolist():
                c = smap.get(km, 0)
                if c == 0:
                    continue
                if winner[km][1] != g:
                    exp_lost += 1
                else:
                    exp.append(c)
            assert cc[g] == len(exp) and lost[g] == exp_lost, (trial, g)
            assert np.array_equal(got, np.sort(np.array(exp, dtype=np.uint32)))
    cc, off, covs, lost = db.reassign_view(sk, sc, np.zeros(0, dtype=np.uint32), np.zeros(0))
    assert cc.sum() == 0 and lost.sum() == 0 and len(covs) == 0
    with pytest.raises(S.SylphHipError):
        db.reassign_view(sk, sc, np.array([1, 1], dtype=np.uint32), np.array([0.9, 0.9]))
    db.close()


def test_reassign_with_kmers_shared_by_hundreds_of_genomes(ctx):
    """The winner pass when 400 strains hold (almost) the same k-mers in genome_kmers AND in their tracked sets: overflow runs of
    hundreds of postings in both indexes, walked by the whole wavefront; winner = highest first-pass ANI, ties to the first in
    the passing list; strains outside the passing list neither win nor lose; every genome's kept coverages and kmers_lost
    against a direct restatement of winner_table + the winner pass (contain.rs:410-430, :637-646)."""
    rng = np.random.default_rng(123)
    thr = O.threshold(200)
    core = np.unique(rng.integers(0, thr, size=250, dtype=np.uint64))
    tcore = np.unique(rng.integers(0, thr, size=120, dtype=np.uint64))
    G = 460
    genomes, tracked = [], []
    for g in range(G):
        if g < 400:
            genomes.append(np.concatenate([core[rng.random(len(core)) < 0.95], rng.integers(0, thr, size=20, dtype=np.uint64)]))
            # tracked sets: the shared tracked core + some of the OTHER strains' genome k-mers (they can steal ownership)
            tracked.append(np.concatenate([tcore[rng.random(len(tcore)) < 0.9], core[rng.random(len(core)) < 0.05]]))
        else:
            genomes.append(np.concatenate([rng.integers(0, thr, size=200, dtype=np.uint64), tcore[:30], core[:5]]))
            tracked.append(np.zeros(0, dtype=np.uint64))
    db_k, goff = flat_db(genomes)
    t_k, toff = flat_db(tracked)
    sk = np.unique(np.concatenate([core, tcore, rng.integers(0, thr, size=3000, dtype=np.uint64)] + [g[-20:] for g in genomes[:50]]))
    sc = rng.integers(0, 5, size=len(sk)).astype(np.uint32)
    smap = dict(zip(sk.tolist(), sc.tolist()))
    db = S.Database(ctx, db_k, goff)
    db.attach_tracked(t_k, toff)
    for trial in range(3):
        passing = rng.permutation(G)[: int(rng.integers(250, G + 1))]
        ani = rng.choice([0.95, 0.96, 0.97, 0.97, 0.99], size=len(passing)) if trial else np.full(len(passing), 0.97)   # trial 0: all tied
        winner = {}
        for r, g in enumerate(passing):
            for km in list(genomes[g].tolist()) + list(tracked[g].tolist()):
                if km not in winner or ani[r] > winner[km][0]:
                    winner[km] = (ani[r], int(g))
        cc, off, covs, lost = db.reassign_view(sk, sc, passing, ani)
        pset = set(int(x) for x in passing)
        for g in range(G):
            got = covs[int(off[g]):int(off[g + 1])]
            if g not in pset:
                assert cc[g] == 0 and len(got) == 0 and lost[g] == 0
                continue
            exp, exp_lost = [], 0
            for km in genomes[g].tolist():
                c = smap.get(km, 0)
                if c == 0:
                    continue
                if winner[km][1] != g:
                    exp_lost += 1
                else:
                    exp.append(c)
            assert cc[g] == len(exp) and lost[g] == exp_lost, (trial, g)
            assert np.array_equal(got, np.sort(np.array(exp, dtype=np.uint32)))
        assert int(lost.sum()) > 10000
    db.close()


# ---------------------------------------------------------------------------------------------- fuzz over read shapes
@pytest.mark.parametrize("seed", range(int(os.environ.get("SYLPH_FUZZ_SEEDS", "6"))))   # (a longer campaign: SYLPH_FUZZ_SEEDS=200)
def test_read_shapes_fuzz(ctx, seed):
    """Random mixtures of record lengths around every threshold of the seeding kernels (k, k+1, 33, 66, 400, 401, tiny reads
    that put more than 256 records into one block of the read-per-lane kernel, long reads that make it decline), random
    alphabets, pairs and singles, 1-3 batches, through all seeding/finishing flavours against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    genome = random_seq(rng, 30000)
    shapes = [
        lambda: int(rng.integers(0, 40)),                      # tiny: hundreds of records per block, most without k-mers
        lambda: int(rng.choice([30, 31, 32, 33, 34, 35, 65, 66, 67])),
        lambda: int(rng.integers(100, 160)),
        lambda: int(rng.choice([398, 399, 400])),
        lambda: int(rng.integers(240, 310)),
    ]
    if seed % 3 == 2:
        shapes.append(lambda: int(rng.choice([401, 402, 1500])))    # beyond the read-per-lane kernel: whole batch falls back
    n_pairs = int(rng.integers(300, 1500))
    pick = rng.integers(0, len(shapes), size=2)
    recs = []
    for _ in range(n_pairs):
        for m in range(2):
            L = shapes[pick[m] if rng.random() < 0.8 else int(rng.integers(0, len(shapes)))]()
            s = int(rng.integers(0, len(genome) - L)) if L else 0
            r = genome[s:s + L].copy()
            if rng.random() < 0.5:
                r = revcomp(r)
            if L and rng.random() < 0.05:
                r[rng.integers(0, L, size=max(1, L // 20))] = rng.choice(np.frombuffer(b"NnRYacgtu", dtype=np.uint8))
            recs.append(r)
    for _ in range(n_pairs // 5):                                 # exact duplicate pairs
        j = 2 * int(rng.integers(0, len(recs) // 2))
        recs += [recs[j].copy(), recs[j + 1].copy()]
    b, off = concat(recs)
    c = int(rng.choice([1, 3, 20, 200]))
    for paired in (False, True):
        for gm, om in MODES:
            e = O.sketch_reads(b, off, c=c, mode=om, paired=paired)
            g = sketch_gpu(ctx, b, off, paired=paired, seed_mode=gm, c=c, batches=int(rng.integers(1, 4)))
            assert_same_sketch(g, e)
            if paired:       # the same shapes behind the filter (a10): a leaky one that has to grow, against the oracle's walk of it
                cap = int(rng.choice([600, 2500, 40000]))
                fpr = float(rng.choice([1e-4, 0.02, 0.3]))
                ef = O.sketch_reads_cuckoo_model(b, off, c=c, mode=om, fpr=fpr, initial_capacity=cap)
                n_ops = 2 * (int(ef["counts"].sum()) + ef["dup_removed"])
                if n_ops <= cap * 255:                             # (a10.hip stops at 8 filters: c = 1 with the smallest capacity may need more)
                    gf = sketch_gpu(ctx, b, off, paired=True, seed_mode=gm, c=c, batches=int(rng.integers(1, 4)), dedup_fpr=fpr, dedup_capacity=cap)
                    assert_same_sketch(gf, ef)


def test_short_reads_over_the_whole_byte_alphabet(ctx):
    """BYTE_TO_SEQ (types.rs:50-59) inside the short-read kernel: reads salted with N / n, IUPAC letters, U / u (code 3), the
    raw values 0-3 (codes 0-3) and arbitrary bytes, at rates from "one odd byte per wavefront" to "every byte odd", plus reads
    written entirely in raw codes.  The ASCII -> 2-bit pack clears the codes of bytes that map to 0 in place and sends only U,
    u and raw 0-3 through its exact path: both must agree with the oracle, in sessions (read-per-lane and position kernel)."""
    rng = np.random.default_rng(4242)
    genome = random_seq(rng, 20000)
    pools = [np.frombuffer(b"Nn", dtype=np.uint8), np.frombuffer(b"RYKMSWBDHVrykmswbdhv-.*", dtype=np.uint8), np.frombuffer(b"Uu", dtype=np.uint8),
             np.array([0, 1, 2, 3], dtype=np.uint8), np.arange(256, dtype=np.uint8)]
    recs = []
    for i in range(4000):
        L = int(rng.integers(60, 152))
        s0 = int(rng.integers(0, len(genome) - L))
        r = genome[s0:s0 + L].copy()
        if rng.random() < 0.3:
            r = np.char.lower(r.view("S1")).view(np.uint8).copy()
        rate = float(rng.choice([0.0, 0.001, 0.01, 0.2, 1.0], p=[0.2, 0.3, 0.3, 0.15, 0.05]))
        hit = rng.random(L) < rate
        pool = pools[int(rng.integers(0, len(pools)))]
        r[hit] = rng.choice(pool, size=int(hit.sum()))
        if i % 50 == 0:                                     # a read in raw 2-bit codes: A/C/G/T as 0/1/2/3
            r = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), genome[s0:s0 + L]).astype(np.uint8)
        recs.append(r)
    b, off = concat(recs)
    for paired in (False, True):
        for gm, om in MODES:
            e = O.sketch_reads(b, off, c=5, mode=om, paired=paired)
            assert len(e["kmers"]) > 1000
            assert_same_sketch(sketch_gpu(ctx, b, off, paired=paired, seed_mode=gm, c=5), e)


def test_tiny_reads_many_records_per_block(ctx):
    """Reads of 0-45 bases: a block of the read-per-lane kernel holds more records than lanes (several passes per block), most
    records have no k-mer at all, mates shorter than 33 bases carry no marker (sketch.rs:661)."""
    rng = np.random.default_rng(77)
    genome = random_seq(rng, 5000)
    recs = []
    for _ in range(6000):
        L = int(rng.integers(0, 46))
        s = int(rng.integers(0, len(genome) - 46))
        recs.append(genome[s:s + L].copy())
    recs[3000:3000] = [genome[:0].copy() for _ in range(3000)]   # thousands of empty records inside one block
    b, off = concat(recs)
    for paired in (False, True):
        for c in (1, 5):
            e = O.sketch_reads(b, off, c=c, paired=paired)
            assert len(e["kmers"]) > 50
            assert_same_sketch(sketch_gpu(ctx, b, off, paired=paired, c=c), e)


@pytest.mark.parametrize("k", [21, 31])
def test_reads_kernel_hash_spellings_and_the_pruning_road(ctx, k):
    """The read-per-lane kernel's k-mer loop in its three spellings (ctx option "reads_hash": 0 the compiler's lowering, 1 the
    hand-scheduled 64-bit hash + exact test, 2 the last hash step and the test on the HIGH word only).  Spelling 2 marks a superset of
    the seeds — about 3 k-mers in 2^32 too many — which the survivors' pass strikes from the hit masks before it redoes its bookkeeping:
    "reads_slack" widens the superset (by a third, and to five times the seeds, where the slots and the survivors' list overflow while
    the wrong candidates are still in) so that this road is taken in nearly every pass.  Ragged pairs, tiny reads (several passes per
    block), odd bytes, packed 2-bit input, c from 2 (c = 1 cannot use spelling 2: falls back to 1) to 1000; same tables as the oracle."""
    from sylph_amd.binding import ENC_2BIT, MEM_HOST
    rng = np.random.default_rng(9100 + k)
    genome = random_seq(rng, 60000)
    ragged = make_reads(rng, genome, 3000, 150, paired=True, dup_frac=0.2, ragged=True)
    for r in ragged[::13]:
        if len(r) > 5:
            r[int(rng.integers(0, len(r)))] = ord("N")
    tiny = [genome[s:s + int(rng.integers(0, 60))].copy() for s in rng.integers(0, 50000, size=8000)]
    equal = make_reads(rng, genome, 2500, 150, paired=True, dup_frac=0.1, ragged=False)
    cases = [(ragged, True), (tiny, False), (tiny, True), (equal, True), (equal, False)]
    try:
        for c in (1, 2, 3, 20, 200, 1000):
            th = (0xFFFFFFFFFFFFFFFF // c) >> 32
            slacks = [0] if c == 1 else [0, th // 3, min(4 * th, 0xFFFFFFFE - th - 1)]
            for recs, paired in cases:
                b, off = concat(recs)
                e = O.sketch_reads(b, off, c=c, k=k, paired=paired)
                packed = S.pack_2bit(b)
                for hv, slack in [(0, 0), (1, 0)] + [(2, s_) for s_ in slacks]:
                    ctx.set_option("reads_hash", str(hv))
                    ctx.set_option("reads_slack", str(slack))
                    tag = (c, paired, hv, slack)
                    g = sketch_gpu(ctx, b, off, paired=paired, c=c, k=k, batches=1 + (hv + c) % 2)
                    assert_same_sketch(g, e, tag)
                    if hv == 2:
                        sk = S.ReadSketcher(ctx, c=c, k=k, paired=paired)
                        sk.push_enc(packed, off, int(off[-1]), MEM_HOST, ENC_2BIT)
                        g = sk.finish()
                        sk.close()
                        assert_same_sketch(g, e, tag)
                        if paired and c >= 2:        # the pruned slots behind the default pair dedup (deferred verdict, partitioned pass)
                            ef = O.sketch_reads_cuckoo_model(b, off, c=c, k=k, fpr=1e-4)
                            assert_same_sketch(sketch_gpu(ctx, b, off, paired=True, c=c, k=k, dedup_fpr=1e-4), ef, tag)
        # a device batch with a deferred verdict (what the pipeline pushes): the pruned slots are partitioned where they lie
        import torch
        big = make_reads(rng, genome, 12000, 150, paired=True, dup_frac=0.1, ragged=False)
        b, off = concat(big)
        tb = torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).cuda()
        to = torch.from_numpy(off.astype(np.int64)).cuda()
        torch.cuda.synchronize()
        for c in (20, 200):
            th = (0xFFFFFFFFFFFFFFFF // c) >> 32
            e = O.sketch_reads(b, off, c=c, k=k, paired=True)
            ef = O.sketch_reads_cuckoo_model(b, off, c=c, k=k, fpr=1e-4)
            for slack in (0, th // 3, 4 * th):
                ctx.set_option("reads_hash", "2")
                ctx.set_option("reads_slack", str(slack))
                for fpr, want in ((None, e), (1e-4, ef)):
                    ctx.profile(True)
                    sk = S.ReadSketcher(ctx, c=c, k=k, paired=True, **({} if fpr is None else {"dedup_fpr": fpr}))
                    sk.set_option("borrow_until_finish", 1)
                    sk.push_device(tb.data_ptr(), to.data_ptr(), len(off) - 1, int(off[-1]))
                    g = sk.finish()
                    sk.close()
                    deferred = ctx.kernel_stats("deferred")[1]
                    ctx.profile(False)
                    assert_same_sketch(g, want, ("deferred", c, slack, fpr))
                    assert deferred == 1, (c, slack, fpr)
    finally:
        ctx.set_option("reads_hash", "-1")
        ctx.set_option("reads_slack", "0")


def _sketch_in_batches(ctx, batches, c, no_dedup=False, plain="1"):
    ctx.set_option("plain_records", plain)
    try:
        sk = S.ReadSketcher(ctx, c=c, paired=False, no_dedup=no_dedup)
        for recs in batches:
            b, off = concat(recs)
            sk.push(b, off)
        r = sk.finish()
        sk.close()
        return r
    finally:
        ctx.set_option("plain_records", "1")


def test_marker_less_single_end_samples_are_counted_without_records(ctx):
    """Single-end batches whose records carry no dedup marker (reads above 400 bases, sketch.rs:922-927; --no-dedup) keep only
    their hashes and are counted by bucket_count_kernel.  Same tables as with occurrence records (`plain_records` = 0) and as the
    oracle's: (a) long reads only, two batches; (b) a first batch of long reads, then a batch with 66-400 base reads and exact
    duplicates among them (the records of the first batch are written after the fact, dedup applies to the second);
    (c) --no-dedup over a mixture of lengths; (d) a k-mer 3,000 deep — beyond the largest count configuration, through the
    overflow path; (e) reads of 401+ bases that are short on average (the short-read kernel declines, the position kernel runs)."""
    rng = np.random.default_rng(77)
    genome = random_seq(rng, 300000)

    def cut(n, lo, hi):
        out = []
        for _ in range(n):
            L = int(rng.integers(lo, hi))
            s = int(rng.integers(0, len(genome) - L))
            out.append(genome[s:s + L].copy())
        return out
    long_a, long_b = cut(300, 401, 9000), cut(200, 2000, 30000)
    short = cut(400, 66, 401)
    short = short + short[:150] + [short[0]] * 5                                  # exact duplicates: dedup must see them
    deep = [genome[1000:1700].copy()] * 3000
    cases = {
        "long_only": ([long_a, long_b], False),
        "long_then_short": ([long_a, short, long_b], False),
        "short_then_long": ([short, long_a], False),
        "no_dedup_mixture": ([long_a + short, long_b + cut(100, 0, 66)], True),
        "deep_kmers": ([long_a + deep, long_b], False),
        "just_above_400": ([cut(2000, 401, 420)], False),
    }
    for name, (batches, nd) in cases.items():
        flat = [r for bt in batches for r in bt]
        b, off = concat(flat)
        for c in (20, 100):
            e = O.sketch_reads(b, off, c=c, no_dedup=nd)
            for plain in ("1", "0"):
                g = _sketch_in_batches(ctx, batches, c, no_dedup=nd, plain=plain)
                assert np.array_equal(g["kmers"], e["kmers"]), (name, c, plain)
                assert np.array_equal(g["counts"], e["counts"]), (name, c, plain)
                assert g["dup_removed"] == e["dup_removed"], (name, c, plain)
        if name in ("long_then_short", "short_then_long"):
            assert e["dup_removed"] > 0


@pytest.mark.gpu
def test_hit_row_assembly_paths(ctx):
    """The result block is assembled per row since round 4 (csrc/hits.hip: replicated row counters, scan, scatter, per-row sort;
    sizes from device memory) instead of radix-sorting the hit list.  Every branch against the oracle: rows of 1..64 hits (ranked
    inside a wavefront), long rows (LDS histogram over the values), a batch whose largest count does not fit the histogram (4096
    and above: the sorted path of rounds 1-3 takes it, decided from the device word), a hit list that overflows its first buffer
    (probe + assembly run twice), row spaces of different sizes one after the other on the same database (batches of 1, 3, 1
    tables: the counters and tickets must come back clean every time), and the same calls with SYLPH_HIP_HIT_SORT-style sorting
    giving identical blocks is covered by every other contain test having passed on the old path in round 3."""
    rng = np.random.default_rng(404)
    thr = O.threshold(200)
    big = [np.unique(rng.integers(0, thr, size=n, dtype=np.uint64)) for n in (30_000, 12_000, 5_000)]
    small = [rng.integers(0, thr, size=int(rng.integers(50, 90)), dtype=np.uint64) for _ in range(400)]
    clones = [big[0].copy() for _ in range(45)]                     # 45 x 30k shared k-mers: 1.35 M hits > the first hit buffer (1 M)
    genomes = big + small + clones
    db_k, goff = flat_db(genomes)
    sk = np.unique(np.concatenate([big[0], big[1][::2], big[2][::3]] + [g[: int(rng.integers(1, len(g)))] for g in small[::3]] +
                                  [rng.integers(0, thr, size=4000, dtype=np.uint64)]))
    counts_small = rng.integers(1, 200, size=len(sk)).astype(np.uint32)      # u8 values
    counts_mid = counts_small.copy()
    counts_mid[::7] = rng.integers(256, 4000, size=len(counts_mid[::7]))     # u16 values, still inside the histogram
    counts_huge = counts_small.copy()
    counts_huge[5] = 70_000                                                  # beyond the histogram: sorted path, u32 values
    counts_zero = counts_small.copy()
    counts_zero[::2] = 0                                                     # contain.rs:634: zero counts are no hits
    for sc in (counts_small, counts_mid, counts_huge, counts_zero):
        cc = check_contain(ctx, db_k, goff, sk, sc, min_kmers=0.0)
        assert cc[0] == len(big[0]) or sc is counts_zero
        assert int(cc.sum()) > 1_000_000 or sc is counts_zero
    # more distinct rows in one workgroup's stretch of the hit list than its LDS table takes (20,000 genomes, every one hit ~60
    # times by a table that holds the whole pool): the table freezes and the rest of the rows go straight to the global counters
    pool = np.unique(rng.integers(0, thr, size=50_000, dtype=np.uint64))
    many = [rng.choice(pool, size=int(rng.integers(40, 80)), replace=False) for _ in range(20_000)]
    mk, moff = flat_db(many)
    cc = check_contain(ctx, mk, moff, pool, rng.integers(1, 250, size=len(pool)).astype(np.uint32), min_kmers=0.0)
    assert int((cc > 0).sum()) == 20_000
    # the same database object through row spaces of different sizes, and a thinned table whose hits fit the first buffer
    db = S.Database(ctx, db_k, goff)
    G = len(goff) - 1
    thin = sk[::9]
    tables = [(sk, counts_small), (thin, counts_mid[::9]), (sk, counts_huge), (thin, counts_small[::9])]
    exp = {}
    for i, (k_, c_) in enumerate(tables):
        ecc, ecov, _ = O.contain(k_, c_, db_k, goff, min_number_kmers=0.0)
        exp[i] = (ecc, [np.sort(x) for x in ecov])
    for batch in ([1], [1, 3, 0], [3], [0, 1, 2, 3], [2], [1]):
        cc, off, covs = db.contain_batch([tables[i] for i in batch], min_number_kmers=0.0)
        cc, off, covs = np.array(cc), np.array(off), np.array(covs).astype(np.uint32)
        for s, i in enumerate(batch):
            ecc, ecov = exp[i]
            assert np.array_equal(cc[s * G:(s + 1) * G], ecc), (batch, s)
            for g in list(range(3)) + list(range(3, G, 37)) + list(range(G - 45, G, 11)):
                assert np.array_equal(covs[int(off[s * G + g]):int(off[s * G + g + 1])], ecov[g]), (batch, s, g)
    db.close()


def test_deferred_seeding_verdict_and_its_redo(ctx):
    """"borrow_until_finish" (round 4): a session whose device batch stays valid until finish does not wait for the seeding kernel's
    verdict after the push — finish reads it with its own tail block.  The ordinary case, and the three ways the verdict can turn
    out bad or be needed early: a record too long for the short-read kernel, blocks of reads that overflow their slots (20,000
    copies of one pair that carries several seeds), a second push on the same session; plus what must NOT be deferred (host
    memory, tiny batches).  Every table against the oracle; the road taken is read from the context's counters."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2604)
    genome = random_seq(rng, 400_000)

    def pairs(n, L=150):
        st = rng.integers(0, len(genome) - 500, size=n)
        recs = []
        for s in st:
            recs.append(genome[s:s + L])
            recs.append(revcomp(genome[s + 200:s + 200 + L]))
        return recs

    def run(recs_list, paired=True, borrow=True, host=False, c=200):
        sk = S.ReadSketcher(ctx, c=c, k=31, paired=paired)
        if borrow:
            sk.set_option("borrow_until_finish", 1)
        keep = []
        for recs in recs_list:
            b, o = concat(recs)
            if host:
                sk.push(b, o)
            else:
                tb = torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).to(dev)
                to = torch.from_numpy(o.astype(np.int64)).to(dev)
                torch.cuda.synchronize()
                keep.append((tb, to))
                sk.push_device(tb.data_ptr(), to.data_ptr(), len(recs), int(o[-1]))
        g = sk.finish()
        sk.close()
        allrecs = [r for recs in recs_list for r in recs]
        b, o = concat(allrecs)
        e = O.sketch_reads(b, o, c=c, k=31, paired=paired)
        assert np.array_equal(g["kmers"], e["kmers"]) and np.array_equal(g["counts"], e["counts"]) and g["dup_removed"] == e["dup_removed"]
        return g

    def roads(fn):
        ctx.profile(True)
        fn()
        d, r = ctx.kernel_stats("deferred")[1], ctx.kernel_stats("deferred_redo")[1]
        ctx.profile(False)
        return d, r

    normal = pairs(6000) + pairs(50) * 3                                   # ~1.8 Mbp, some exact duplicate pairs
    assert roads(lambda: run([normal])) == (1, 0)
    assert roads(lambda: run([normal], borrow=False)) == (0, 0)
    assert roads(lambda: run([normal], host=True)) == (0, 0)
    assert roads(lambda: run([pairs(100)])) == (0, 0)                      # too small to be worth deferring
    long_rec = pairs(3000) + [genome[1000:1600], revcomp(genome[1200:1350])] + pairs(3000)
    assert roads(lambda: run([long_rec])) == (1, 1)
    one = None
    for s in range(0, 5000, 7):                                            # a pair whose mate 1 carries >= 3 seeds at c = 200
        if len(O.extract_markers(genome[s:s + 150], c=200, k=31)) >= 3:
            one = [genome[s:s + 150], revcomp(genome[s + 200:s + 350])]
            break
    assert one is not None
    assert roads(lambda: run([one * 20_000])) == (1, 1)                    # every block overflows its slots
    assert roads(lambda: run([normal, pairs(4000)])) == (1, 0)             # the second push reads the first one's verdict itself
    assert roads(lambda: run([long_rec, pairs(4000)])) == (1, 1)
    single = [r for r in pairs(6000)]
    assert roads(lambda: run([single], paired=False)) == (1, 0)
