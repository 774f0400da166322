"""BASELINE configs[0] at FULL size on the GPU: the reference's three E. coli assemblies (4.6 / 4.6 / 5.5 Mbp, packed into
tests/golden/ecoli_full_bases.npz by make_ecoli_full.py) and its k12_R1/R2.fq reads go through the HIP path — through the C ABI —
and are compared with SURVEY.md Appendix A.2's known answers (tests/golden/survey_kat.json).  That answer key was computed during
the survey by an independent numpy restatement of the Rust sources; it was NOT produced by oracle/, so this file does not compare
the GPU with the oracle twice.  The derived floats (naive ANI, lambda, adjusted ANI, mean coverage) come from the product's host
statistics (libsylph_host.so) on the GPU's integers and are held to the north_star's 1e-6."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import sylph_amd as S

from .helpers import concat, hist, xor_sum
from .test_host import HostStats

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["e.coli-EC590.fasta.gz", "e.coli-K12.fasta.gz", "e.coli-o157.fasta.gz"]


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "survey_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def genomes(golden_dir):
    z = np.load(os.path.join(golden_dir, "ecoli_full_bases.npz"))
    out = []
    for g in range(3):
        n = int(z[f"g{g}_n"][0])
        p = z[f"g{g}_packed"]
        codes = np.stack([(p >> 6) & 3, (p >> 4) & 3, (p >> 2) & 3, p & 3], axis=1).reshape(-1)[:n]
        bases = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].copy()
        bases[z[f"g{g}_exc_pos"].astype(np.int64)] = z[f"g{g}_exc_val"]
        assert hashlib.md5(bases.tobytes()).hexdigest() == bytes(z[f"g{g}_md5"]).decode()
        out.append((bases, z[f"g{g}_off"].astype(np.uint64)))
    return out


def test_fixture_is_the_reference_file(genomes):
    # md5 of EC590's concatenated sequence bytes
    assert hashlib.md5(genomes[0][0].tobytes()).hexdigest() == "55b3b290787438c6846bb23d510be991"


@pytest.mark.parametrize("mode", [S.SEED_SCALAR, S.SEED_AVX2_COMPAT])
def test_full_genome_sketches_match_the_survey_key(ctx, genomes, kat, mode):
    """sketch_genome (sketch.rs:550-622) on the whole assemblies: lengths, xor and wrapping sums of genome_kmers and of the
    tracked k-mers, and the first three genome_kmers (order!), for both seed modes (identical for these files)."""
    for (bases, off), name in zip(genomes, NAMES):
        e = kat["genomes"][name]
        assert len(off) - 1 == e["contigs"] and int(off[-1]) == e["gn_size"]
        r = ctx.sketch_genome(bases, off, c=200, k=31, seed_mode=mode, min_spacing=30, pseudotax=True)
        gk, tr = r["genome_kmers"], r["tracked"]
        assert [len(gk), *xor_sum(gk)] == e["genome_kmers"], name
        assert [len(tr), *xor_sum(tr)] == e["tracked"], name
        assert gk[:3].tolist() == e["first3"]
        assert len(gk) + len(tr) == e["raw"] - _dup_occurrences(ctx, bases, off, mode, e)


def _dup_occurrences(ctx, bases, off, mode, e):
    """raw seeds minus the seeds of k-mers that occur more than once (every occurrence of those is removed, sketch.rs:594-605);
    `dup` in the answer key counts DISTINCT duplicated k-mers, so recount the occurrences from the raw seeds."""
    contig, pos, h = ctx.extract_markers_positions(bases, off, c=200, k=31, seed_mode=mode)
    assert len(h) == e["raw"]
    u, n = np.unique(h, return_counts=True)
    assert int((n > 1).sum()) == e["dup"]
    return int(n[n > 1].sum())


def test_full_genomes_in_one_batch_call(ctx, genomes, kat):
    """The database build path (sylph_sketch_genomes: all three assemblies in ONE call, dedup + spacing on the device)."""
    bases = np.concatenate([b for b, _ in genomes])
    coff, goff, base = [0], [0], 0
    for b, off in genomes:
        coff += [int(x) + base for x in off[1:]]
        base += len(b)
        goff.append(len(coff) - 1)
    km, koff, tr, toff = ctx.sketch_genomes(bases, np.array(coff, np.uint64), np.array(goff, np.uint64), c=200, k=31)
    for g, name in enumerate(NAMES):
        e = kat["genomes"][name]
        a, t = km[int(koff[g]):int(koff[g + 1])], tr[int(toff[g]):int(toff[g + 1])]
        assert [len(a), *xor_sum(a)] == e["genome_kmers"] and [len(t), *xor_sum(t)] == e["tracked"]
        assert a[:3].tolist() == e["first3"]


def _host_stats(covs, n_kmers, min_ani=0.0):
    L = C.CDLL(os.path.join(ROOT, "sylph_amd", "libsylph_host.so"))
    L.sylph_host_stats.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.POINTER(HostStats)]
    cv = np.ascontiguousarray(covs, dtype=np.uint32)
    out = HostStats()
    L.sylph_host_stats(cv.ctypes.data_as(C.c_void_p), len(cv), n_kmers, 31, 3.0, min_ani, 0, 1, 0, 0, C.byref(out))
    return out


def test_reads_and_containment_match_the_survey_key(ctx, genomes, kat, golden_dir):
    """k12_R1.fq single-end, k12_R1+R2 paired (exact dedup), the t1/t2 toy files, the dedup vectors (files concatenated with
    themselves) -> (k-mer, count) tables; then the tables against the three FULL genome sketches -> contain_count, coverage
    histograms, naive ANI; for the paired sample also lambda / adjusted ANI / mean coverage through the host statistics."""
    z = np.load(os.path.join(golden_dir, "k12_reads.npz"))

    def sketch(b, off, paired=False, no_dedup=False):
        sk = S.ReadSketcher(ctx, c=200, k=31, paired=paired, no_dedup=no_dedup)
        sk.push(b, off)
        r = sk.finish()
        sk.close()
        return r

    def interleave(b1, o1, b2, o2):
        recs = []
        for i in range(len(o1) - 1):
            recs.append(b1[int(o1[i]):int(o1[i + 1])])
            recs.append(b2[int(o2[i]):int(o2[i + 1])])
        return concat(recs)

    r1, o1, r2, o2 = z["r1_bases"], z["r1_off"], z["r2_bases"], z["r2_off"]
    tables = {"k12_single": sketch(r1, o1), "k12_paired": sketch(*interleave(r1, o1, r2, o2), paired=True),
              "t_paired": sketch(*interleave(z["t1_bases"], z["t1_off"], z["t2_bases"], z["t2_off"]), paired=True),
              "t1_single": sketch(z["t1_bases"], z["t1_off"]), "t2_single": sketch(z["t2_bases"], z["t2_off"])}
    for name, e in kat["reads"].items():
        t = tables[name]
        assert len(t["kmers"]) == e["distinct"] and int(t["counts"].sum()) == e["total"], name
        assert hist(t["counts"]) == {int(a): b for a, b in e["hist"].items()}
        assert list(xor_sum(t["kmers"])) == [e["keys_xor"], e["keys_sum"]]
        assert t["dup_removed"] == 0
    # dedup vectors: the file concatenated with itself
    def twice(b, off, times=2):
        return np.concatenate([b] * times), np.concatenate([[0]] + [off[1:] + np.uint64(i * int(off[-1])) for i in range(times)]).astype(np.uint64)
    d = kat["dedup"]
    t = sketch(*twice(r1, o1))
    assert (len(t["kmers"]), int(t["counts"].sum()), t["dup_removed"]) == (d["k12_single_x2"]["distinct"], d["k12_single_x2"]["total"], d["k12_single_x2"]["dup_removed"])
    t = sketch(*twice(r1, o1), no_dedup=True)
    assert int(t["counts"].sum()) == d["k12_single_x2_nodedup"]["total"] and hist(t["counts"]) == {int(a): b for a, b in d["k12_single_x2_nodedup"]["hist"].items()}
    t = sketch(*twice(r1, o1, 6))
    assert (int(t["counts"].sum()), t["dup_removed"]) == (d["k12_single_x6"]["total"], d["k12_single_x6"]["dup_removed"])
    pb, po = interleave(r1, o1, r2, o2)
    pb2, po2 = interleave(*twice(r1, o1), *twice(r2, o2))
    t = sketch(pb2, po2, paired=True)
    assert (len(t["kmers"]), int(t["counts"].sum()), t["dup_removed"]) == (d["k12_paired_x2"]["distinct"], d["k12_paired_x2"]["total"], d["k12_paired_x2"]["dup_removed"])
    t = sketch(pb2, po2, paired=True, no_dedup=True)
    assert int(t["counts"].sum()) == d["k12_paired_x2_nodedup"]["total"] and hist(t["counts"]) == {int(a): b for a, b in d["k12_paired_x2_nodedup"]["hist"].items()}

    # containment against the full genome sketches
    gk = [ctx.sketch_genome(b, off, c=200, k=31)["genome_kmers"] for b, off in genomes]
    goff = np.zeros(4, dtype=np.uint64)
    goff[1:] = np.cumsum([len(x) for x in gk])
    db = S.Database(ctx, np.concatenate(gk), goff)
    for sample in ("k12_single", "k12_paired"):
        cc, off, covs = db.contain(tables[sample]["kmers"], tables[sample]["counts"])
        for g, name in enumerate(NAMES):
            e_cc, e_len, e_hist, e_ani = kat["containment"][sample][name]
            mine = covs[int(off[g]):int(off[g + 1])]
            assert int(cc[g]) == e_cc and len(gk[g]) == e_len
            assert hist(mine) == {int(a): b for a, b in e_hist.items()}
            st = _host_stats(mine, len(gk[g]))
            assert abs(st.naive_ani - e_ani) < 1e-6                       # north_star tolerance for derived floats
            if sample == "k12_paired":
                p = kat["paired_stats"][name]
                assert abs(st.lambda_ - p["lambda"]) < 1e-6 and abs(st.final_est_ani - p["ani"]) < 1e-6
                assert st.median_cov == p["median"] and abs(st.mean_cov - p["mean_cov_geq1"]) < 1e-6
    for sample in ("t1_single", "t2_single", "t_paired"):
        cc, _, _ = db.contain(tables[sample]["kmers"], tables[sample]["counts"])
        assert cc.tolist() == [0, 0, 0]
    db.close()
