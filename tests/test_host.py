"""CPU tests of the C++ host layer above the ABI: bincode layouts of .sylsp/.syldb against an independent struct.pack
encoder, and the statistics half of get_stats against the oracle (float tolerance 1e-6, the north_star's bar)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pyref as P

from .helpers import bgzf_compress

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class HostStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("naive_ani", "final_est_ani", "final_est_cov", "mean_cov", "median_cov", "lambda_",
                                          "ani_ci_lo", "ani_ci_hi", "lambda_ci_lo", "lambda_ci_hi")] + \
               [("lambda_status", C.c_int32), ("passed", C.c_int32), ("has_ci", C.c_int32), ("pad", C.c_int32),
                ("contain_count", C.c_uint64), ("n_kmers", C.c_uint64)]


@pytest.fixture(scope="module")
def host():
    L = C.CDLL(os.path.join(ROOT, "sylph_amd", "libsylph_host.so"))
    L.sylph_host_poisson_cdf.restype = C.c_double
    L.sylph_host_poisson_cdf.argtypes = [C.c_double, C.c_uint64]
    L.sylph_host_stats.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.POINTER(HostStats)]
    L.sylph_host_write_sylsp.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p,
                                         C.c_char_p, C.c_int, C.c_double]
    L.sylph_host_read_sylsp.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64] + [C.c_void_p] * 5 + \
                                       [C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p]
    L.sylph_host_read_syldb.restype = C.c_void_p
    L.sylph_host_read_syldb.argtypes = [C.c_char_p]
    L.sylph_host_syldb_size.restype = C.c_uint64
    L.sylph_host_syldb_size.argtypes = [C.c_void_p]
    L.sylph_host_syldb_genome.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 7 + [C.c_char_p, C.c_char_p, C.c_uint64]
    L.sylph_host_syldb_copy.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.sylph_host_syldb_free.argtypes = [C.c_void_p]
    return L


def host_stats(L, covs, n_kmers, k=31, min_ani=-1.0, pseudotax=0, no_ci=1):
    cv = np.ascontiguousarray(covs, dtype=np.uint32)
    out = HostStats()
    L.sylph_host_stats(cv.ctypes.data_as(C.c_void_p), len(cv), n_kmers, k, 3.0, min_ani, pseudotax, no_ci, 0, 0, C.byref(out))
    return out


def test_stats_match_oracle(host):
    rng = np.random.default_rng(0)
    for trial in range(300):
        n_kmers = int(rng.integers(50, 30000))
        lam = float(rng.choice([0.02, 0.1, 0.5, 1.0, 2.5, 8.0, 40.0]))
        hit = rng.random(n_kmers) < rng.uniform(0.05, 1.0)
        covs = rng.poisson(lam, size=n_kmers)[hit]
        covs = covs[covs > 0].astype(np.uint32)
        if trial % 7 == 0 and len(covs):
            covs[rng.integers(0, len(covs), size=3)] = 100000          # outliers -> Poisson cap
        e = O.stats(covs, n_kmers, min_ani=0.0)
        h = host_stats(host, covs, n_kmers, min_ani=0.0)
        if len(covs) == 0:
            assert h.passed == 0
            continue
        assert h.passed == 1 and h.lambda_status == e.lambda_status
        for f in ("naive_ani", "final_est_ani", "final_est_cov", "mean_cov", "median_cov"):
            assert abs(getattr(h, f) - getattr(e, f)) <= 1e-6 * max(1.0, abs(getattr(e, f))), (trial, f)
        if e.lambda_status == 2:
            assert abs(h.lambda_ - e.lambda_) <= 1e-9
        # default thresholds: query 0.90, profile 0.95 (contain.rs:746-748)
        assert host_stats(host, covs, n_kmers).passed == int(e.final_est_ani >= 0.9)
        assert host_stats(host, covs, n_kmers, pseudotax=1).passed == int(e.final_est_ani >= 0.95)
        assert host_stats(host, covs, n_kmers, min_ani=99.0).passed == int(e.final_est_ani >= 0.99)
    for lam in (1.0, 5.0, 29.0):
        for x in (0, 3, 10, 50, 68):
            assert abs(host.sylph_host_poisson_cdf(lam, x) - O.poisson_cdf(lam, x)) < 1e-12


def test_estimate_unknown_pieces_vs_independent_restatement(host):
    """-u (contain.rs:901-951, :392-408): the read k-mer identity from the count table (both branches: shallow short-read samples
    get 0.995^k, everything else the share of multiplicity > 1 counts) and the explained share of bases, against oracle/pyref.py."""
    from oracle import pyref as P
    host.sylph_host_kmer_identity.restype = C.c_double
    host.sylph_host_kmer_identity.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_double]
    host.sylph_host_covered_bases.restype = C.c_double
    host.sylph_host_covered_bases.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double]
    rng = np.random.default_rng(7)
    branches = set()
    for trial in range(200):
        n = int(rng.choice([0, 1, 7, 300, 5000]))
        lam = float(rng.choice([0.05, 0.8, 2.0, 3.5, 9.0]))
        counts = (rng.poisson(lam, size=n) + 1).astype(np.uint32)
        if trial % 9 == 0 and n:
            counts[rng.integers(0, n, size=2)] = 4_000_000_000                     # the u32 accumulator of the reference wraps
        mean_len = float(rng.choice([100.0, 150.5, 399.9, 400.0, 9000.0]))
        k = int(rng.choice([21, 31]))
        want = P.get_kmer_identity([int(x) for x in counts], k, mean_len)
        got = host.sylph_host_kmer_identity(counts.ctypes.data_as(C.c_void_p), n, k, mean_len)
        assert abs(got - want) <= 1e-12 * max(1.0, abs(want)), (trial, got, want)
        branches.add("fixed" if want == 0.995 ** float(k) else ("one" if want == 1.0 else "eps"))
        G = int(rng.integers(0, 6))
        gn = rng.integers(100_000, 9_000_000, size=G).astype(np.uint64)
        cov = rng.uniform(0.01, 50.0, size=G)
        want = P.estimate_covered_bases([int(x) for x in gn], [float(x) for x in cov], 200, int(counts.astype(np.uint64).sum()), mean_len, k)
        got = host.sylph_host_covered_bases(gn.ctypes.data_as(C.c_void_p), cov.ctypes.data_as(C.c_void_p), G,
                                            counts.ctypes.data_as(C.c_void_p), n, 200, k, mean_len)
        assert abs(got - want) <= 1e-12 * max(1.0, abs(want)), (trial, got, want)
    assert {"fixed", "eps"} <= branches


def test_poisson_cdf_against_mpmath_table(host, golden_dir):
    """The Poisson tail behind the coverage cap (contain.rs:664-675; statrs Poisson::cdf = Q(x+1, lambda)) against a table
    computed with mpmath at 50 digits (tests/golden/make_poisson_table.py) — NOT against the oracle, whose incomplete-gamma
    routine is the host's twin.  The quantity that matters is the decision `cdf < 0.9999999999` for medians 1..29."""
    import json
    with open(os.path.join(golden_dir, "poisson_q_table.json")) as f:
        t = json.load(f)
    closest = 1.0
    for r in t["rows"]:
        lam, x = float(r["lambda"]), int(r["x"])
        exact, tail = float(r["q"]), float(r["one_minus_q"])
        for name, got in (("host", host.sylph_host_poisson_cdf(lam, x)), ("oracle", O.poisson_cdf(lam, x)), ("pyref", P.poisson_cdf(lam, float(x)))):
            assert abs(got - exact) <= (4e-15 if exact > 0.999 else 1e-13), (name, lam, x, got, exact)
            assert (got < 0.9999999999) == r["below_cutoff"], (name, lam, x)
        closest = min(closest, abs(tail - 1e-10) / 1e-10)
    # no (median, coverage) pair sits within f64 noise of the cut-off: the nearest tail (median 7, coverage 29:
    # 1 - Q = 0.9983e-10) differs from 1e-10 by 0.17 % = 1.7e-13 absolute, ~40x the 4e-15 agreement demanded near the cut-off
    assert closest > 1.5e-3
    for lam, cap in t["largest_admitted_cov"].items():
        covs = np.array([int(lam)] * 60 + [cap, cap + 1], dtype=np.uint32)      # median = lam; cap admitted, cap+1 cut
        h = host_stats(host, covs, 100, min_ani=0.0)
        assert h.mean_cov == pytest.approx((60 * int(lam) + cap) / 62.0, rel=1e-12)     # Σfull / |covs| (contain.rs:690)


def test_stats_match_independent_restatement(host):
    """Host statistics (C++, f64) against oracle/pyref.py (pure Python, scipy's incomplete gamma, dict histograms): every
    branch of contain.rs:657-764 and inference.rs:207-242 at 1e-9 relative — tighter than the north_star's 1e-6."""
    rng = np.random.default_rng(20250711)
    n_lambda = 0
    for trial in range(400):
        kind = trial % 6
        n_hits = int(rng.choice([1, 5, 24, 25, 26, 60, 300, 2000]))
        if kind == 0:
            covs = rng.choice([1, 2], size=n_hits, p=[0.85, 0.15])
        elif kind == 1:
            covs = rng.choice([1, 2, 3, 4], size=n_hits)                                  # ties for the mode
        elif kind == 2:
            covs = rng.poisson(float(rng.choice([0.3, 1.0, 2.5, 8.0, 14.0, 16.0, 29.0, 31.0, 80.0])), size=n_hits) + 1
        elif kind == 3:
            covs = np.full(n_hits, int(rng.integers(1, 5)))                                 # a single distinct value
        elif kind == 4:
            covs = np.concatenate([rng.choice([1, 3], size=n_hits), [2] * int(rng.integers(0, 4))])
        else:
            covs = np.concatenate([rng.poisson(1.0, size=n_hits) + 1, rng.integers(50, 100000, size=3)])   # beyond the cap
        covs = covs.astype(np.uint32)
        L = len(covs) + int(rng.integers(0, 40000))
        e = P.stats(len(covs), covs.tolist(), L)
        h = host_stats(host, covs, L, min_ani=0.0)
        assert h.passed == 1 and {0: "LOW", 1: "HIGH", 2: "LAMBDA"}[h.lambda_status] == e["lambda_status"], trial
        for f in ("naive_ani", "final_est_ani", "final_est_cov", "mean_cov", "median_cov"):
            assert getattr(h, f) == pytest.approx(e[f], rel=1e-9, abs=0), (trial, f)
        if e["lambda_"] is not None:
            n_lambda += 1
            assert h.lambda_ == pytest.approx(e["lambda_"], rel=1e-12)
        assert (h.contain_count, h.n_kmers) == (len(covs), L)
        for thr, kw in ((0.90, {}), (0.95, dict(pseudotax=1)), (0.99, dict(min_ani=99.0))):
            assert host_stats(host, covs, L, **kw).passed == int(e["final_est_ani"] >= thr)
    assert n_lambda > 40


def test_bootstrap_ci_is_deterministic_and_ordered(host):
    rng = np.random.default_rng(3)
    covs = rng.poisson(0.8, size=5000)
    covs = covs[covs > 0].astype(np.uint32)
    a = host_stats(host, covs, 8000, min_ani=0.0, no_ci=0)
    b = host_stats(host, covs, 8000, min_ani=0.0, no_ci=0)
    assert a.lambda_status == 2 and a.has_ci == 1
    assert (a.ani_ci_lo, a.ani_ci_hi, a.lambda_ci_lo, a.lambda_ci_hi) == (b.ani_ci_lo, b.ani_ci_hi, b.lambda_ci_lo, b.lambda_ci_hi)
    assert a.ani_ci_lo <= a.final_est_ani <= a.ani_ci_hi + 1e-3 and a.lambda_ci_lo <= a.lambda_ci_hi


def bincode_sylsp(kmers, counts, c, k, file_name, sample_name, paired, mean):
    """Independent encoder of SequencesSketch (types.rs:145-155) in bincode 1.3.3 default options."""
    b = struct.pack("<Q", len(kmers))
    for a, x in zip(kmers, counts):
        b += struct.pack("<QI", int(a), int(x))
    b += struct.pack("<QQ", c, k)
    b += struct.pack("<Q", len(file_name)) + file_name
    b += (b"\x01" + struct.pack("<Q", len(sample_name)) + sample_name) if sample_name is not None else b"\x00"
    b += struct.pack("<Bd", 1 if paired else 0, mean)
    return b


def test_sylsp_layout_and_roundtrip(host, tmp_path):
    rng = np.random.default_rng(1)
    kmers = np.sort(rng.integers(0, 2**63, size=1000, dtype=np.uint64))
    counts = rng.integers(1, 2**32, size=1000, dtype=np.uint64).astype(np.uint32)
    for sample_name, paired in ((None, False), (b"my sample", True)):
        p = str(tmp_path / "x.sylsp").encode()
        assert host.sylph_host_write_sylsp(p, kmers.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), len(kmers), 200, 31,
                                           b"dir/reads.fq.gz", sample_name, int(paired), 149.25) == 0
        assert open(p, "rb").read() == bincode_sylsp(kmers, counts, 200, 31, b"dir/reads.fq.gz", sample_name, paired, 149.25)
        # a file written in a different element order (the reference writes hash-map order) reads back identically
        perm = rng.permutation(len(kmers))
        open(p, "wb").write(bincode_sylsp(kmers[perm], counts[perm], 100, 21, b"a", sample_name, paired, 70.0))
        ok = np.zeros(1000, dtype=np.uint64); oc = np.zeros(1000, dtype=np.uint32)
        n, c, k, mean = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_double()
        pr, hs = C.c_int(), C.c_int()
        fn, sn = C.create_string_buffer(256), C.create_string_buffer(256)
        assert host.sylph_host_read_sylsp(p, ok.ctypes.data_as(C.c_void_p), oc.ctypes.data_as(C.c_void_p), 1000, C.byref(n), C.byref(c),
                                          C.byref(k), C.byref(pr), C.byref(mean), fn, sn, 256, C.byref(hs)) == 0
        assert (n.value, c.value, k.value, pr.value, mean.value, fn.value) == (1000, 100, 21, int(paired), 70.0, b"a")
        assert np.array_equal(ok, kmers) and np.array_equal(oc, counts)
        assert (hs.value == 1 and sn.value == sample_name) or (hs.value == 0 and sample_name is None)
    assert host.sylph_host_read_sylsp(b"/nonexistent.sylsp", None, None, 0, C.byref(n), C.byref(c), C.byref(k), C.byref(pr),
                                      C.byref(mean), fn, sn, 256, C.byref(hs)) == -1


def test_syldb_layout(host, tmp_path):
    """Vec<GenomeSketch> (types.rs:163-173) written by an independent encoder is read back field by field."""
    rng = np.random.default_rng(2)
    genomes = []
    b = struct.pack("<Q", 3)
    for i in range(3):
        gk = rng.integers(0, 2**63, size=int(rng.integers(0, 500)), dtype=np.uint64)
        tr = None if i == 1 else rng.integers(0, 2**63, size=int(rng.integers(0, 80)), dtype=np.uint64)
        fn, cn = f"genomes/g{i}.fa.gz".encode(), f"contig_{i} some description".encode()
        b += struct.pack("<Q", len(gk)) + gk.tobytes()
        b += b"\x00" if tr is None else (b"\x01" + struct.pack("<Q", len(tr)) + tr.tobytes())
        b += struct.pack("<Q", len(fn)) + fn + struct.pack("<Q", len(cn)) + cn
        b += struct.pack("<QQQQ", 200, 31, 5_000_000 + i, 30)
        genomes.append((gk, tr, fn, cn))
    p = str(tmp_path / "d.syldb").encode()
    open(p, "wb").write(b)
    h = host.sylph_host_read_syldb(p)
    assert h and host.sylph_host_syldb_size(h) == 3
    for i, (gk, tr, fn, cn) in enumerate(genomes):
        nk, nt, c, k, gs, ms = (C.c_uint64() for _ in range(6))
        ht = C.c_int()
        f, cname = C.create_string_buffer(256), C.create_string_buffer(256)
        host.sylph_host_syldb_genome(h, i, C.byref(nk), C.byref(nt), C.byref(ht), C.byref(c), C.byref(k), C.byref(gs), C.byref(ms), f, cname, 256)
        assert (nk.value, c.value, k.value, gs.value, ms.value, f.value, cname.value) == (len(gk), 200, 31, 5_000_000 + i, 30, fn, cn)
        assert ht.value == (0 if tr is None else 1) and nt.value == (0 if tr is None else len(tr))
        ok = np.zeros(max(1, len(gk)), dtype=np.uint64); ot = np.zeros(max(1, nt.value), dtype=np.uint64)
        host.sylph_host_syldb_copy(h, i, ok.ctypes.data_as(C.c_void_p), ot.ctypes.data_as(C.c_void_p))
        assert np.array_equal(ok[:len(gk)], gk) and (tr is None or np.array_equal(ot[:len(tr)], tr))
    host.sylph_host_syldb_free(h)
    open(p, "wb").write(b[:100])
    assert not host.sylph_host_read_syldb(p)   # truncated file -> error, no crash


def test_inspect_command_yaml(host, tmp_path):
    """`sylph-hip inspect` (inspect.rs:117-233, no GPU involved) on files written by the independent bincode encoders: the
    reference's own assertions (the output names the sketched files, tests/integration_test.rs:505-549), the documents parsed
    with PyYAML field by field, and the serde_yaml layout (block sequences not indented under their key, field order of the
    structs, `null` for a missing sample name)."""
    import subprocess
    import yaml
    rng = np.random.default_rng(5)
    b = struct.pack("<Q", 2)
    gen = []
    for i, (fn, cn) in enumerate(((b"test_files/e.coli-EC590.fasta.gz", b"NZ_CP016182.2 Escherichia coli strain EC590 chromosome, complete genome"),
                                  (b"test_files/e.coli-K12.fasta.gz", b"contig: with a colon # and a hash"))):
        gk = rng.integers(0, 2**63, size=100 + i, dtype=np.uint64)
        b += struct.pack("<Q", len(gk)) + gk.tobytes() + b"\x00"
        b += struct.pack("<Q", len(fn)) + fn + struct.pack("<Q", len(cn)) + cn
        b += struct.pack("<QQQQ", 200, 31, 4_600_000 + i, 30)
        gen.append((fn.decode(), cn.decode(), len(gk), 4_600_000 + i))
    db = tmp_path / "db.syldb"
    db.write_bytes(b)
    km = np.sort(rng.integers(0, 2**62, size=777, dtype=np.uint64))
    cnt = rng.integers(1, 9, size=777, dtype=np.uint32)
    sp = tmp_path / "k12_R1.fq.paired.sylsp"
    sp.write_bytes(bincode_sylsp(km, cnt, 200, 31, b"test_files/k12_R1.fq", None, True, 148.25))
    sp2 = tmp_path / "named.sylsp"
    sp2.write_bytes(bincode_sylsp(km[:10], cnt[:10], 100, 21, b"reads.fq.gz", b"12345", False, 150.0))
    exe = os.path.join(ROOT, "sylph_amd", "sylph-hip")
    r = subprocess.run([exe, "inspect", str(db), str(sp), str(sp2), str(tmp_path / "notes.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "notes.txt file is not a .sylsp or .syldb file" in r.stderr
    out = r.stdout
    assert "e.coli-EC590.fasta.gz" in out and "e.coli-K12.fasta.gz" in out and "k12_R1.fq" in out      # the reference's assertions
    docs = yaml.safe_load(out)
    assert docs[0] == {"database_file": str(db), "c": 200, "k": 31, "min_spacing_parameter": 30,
                       "genome_files": [{"file_name": f, "genome_kmers_num": n, "first_contig_name": c, "genome_size": g} for f, c, n, g in gen]}
    approx = np.float32(148.25 + 31 - 1) / np.float32(148.25) * np.float32(200) * np.float32(777)
    assert docs[1]["file_name"] == "test_files/k12_R1.fq" and docs[1]["num_sketched_kmers"] == 777 and docs[1]["paired"] is True
    assert docs[1]["sample_name"] is None and docs[1]["mean_read_length"] == 148.25 and docs[1]["c"] == 200 and docs[1]["k"] == 31
    assert np.float32(docs[1]["approximate_number_bases"]) == approx
    assert docs[2]["sample_name"] == "12345" and docs[2]["paired"] is False and docs[2]["mean_read_length"] == 150.0
    lines = out.split("\n")
    assert lines[0] == f"- database_file: {db}" and lines[4] == "  genome_files:" and lines[5].startswith("  - file_name: ")
    assert "    first_contig_name: 'contig: with a colon # and a hash'" in lines
    assert "  sample_name: null" in lines and "  sample_name: '12345'" in lines and "  mean_read_length: 150.0" in lines
    assert [l.split(":")[0].strip("- ") for l in lines[13:21]] == ["file_name", "c", "k", "num_sketched_kmers", "approximate_number_bases",
                                                                   "mean_read_length", "sample_name", "paired"]
    # -o writes the same text to a file
    r2 = subprocess.run([exe, "inspect", str(sp), "-o", str(tmp_path / "o.yaml")], capture_output=True, text=True)
    assert r2.returncode == 0 and (tmp_path / "o.yaml").read_text() == "\n".join(lines[13:21]) + "\n"


def test_inspect_scalars(host):
    """Floats as ryu prints them (shortest round-trip digits; fixed notation up to 16 / 13 integer digits, a trailing .0 on
    integers) against numpy's shortest-digit formatter, and strings that must be quoted to stay strings."""
    for fn in (host.sylph_host_inspect_f32, host.sylph_host_inspect_f64, host.sylph_host_inspect_str):
        fn.restype = C.c_uint64
    host.sylph_host_inspect_f32.argtypes = [C.c_float, C.c_char_p, C.c_uint64]
    host.sylph_host_inspect_f64.argtypes = [C.c_double, C.c_char_p, C.c_uint64]
    host.sylph_host_inspect_str.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
    buf = C.create_string_buffer(256)

    def f32(v):
        assert host.sylph_host_inspect_f32(float(np.float32(v)), buf, 256)
        return buf.value.decode()

    def f64(v):
        assert host.sylph_host_inspect_f64(float(v), buf, 256)
        return buf.value.decode()

    def st(x):
        assert host.sylph_host_inspect_str(x.encode(), buf, 256)
        return buf.value.decode()

    rng = np.random.default_rng(9)
    for v in list(rng.uniform(1, 2e12, size=200)) + list(rng.uniform(1e-4, 1, size=50)) + [1.0, 150.0, 0.5, 1234567936.0, 16777216.0]:
        want = np.format_float_positional(np.float32(v), unique=True, trim="0")
        assert f32(v) == want and np.float32(f32(v)) == np.float32(v), (v, f32(v), want)
        want = np.format_float_positional(np.float64(v), unique=True, trim="0")
        assert f64(v) == want and float(f64(v)) == float(v), (v, f64(v), want)
    assert f64(1e16) == "1e16" and f64(1.5e16) == "1.5e16" and f64(1e15) == "1000000000000000.0" and f64(1e-5) == "0.00001" and f64(1e-6) == "1e-6"
    assert f32(1e13) == "1e13" and f32(3e12) == "3000000000000.0" and f64(0.0) == "0.0"
    assert f64(float("inf")) == ".inf" and f64(float("-inf")) == "-.inf" and f64(float("nan")) == ".nan"
    for plain in ("test_files/k12_R1.fq", "NZ_CP016182.2 Escherichia coli strain EC590 chromosome, complete genome", "a-b", "x:y", "1.2.3", "e5"):
        assert st(plain) == plain
    for quoted in ("12345", "1e5", "-3", "0x1F", "true", "null", "~", "", " lead", "trail ", "a: b", "a #b", "- item", "#c", "'q", "[x]", "key:", "1_000"):
        assert st(quoted) == "'" + quoted.replace("'", "''") + "'", quoted
    assert st("tab\there") == '"tab\\there"'


# ---------------------------------------------------------------------------------------------- FASTX records (host feed)
def _py_records(text):
    """Independent reading of needletail's record semantics: 4-line FASTQ, multi-line FASTA, CR stripped, blank lines
    between records ignored."""
    lines = [l.rstrip(b"\r") for l in text.split(b"\n")]
    if lines and lines[-1] == b"":
        lines.pop()
    recs, i = [], 0
    while i < len(lines) and lines[i] == b"":
        i += 1
    if i == len(lines):
        return recs
    if lines[i][:1] == b"@":
        while i < len(lines):
            if lines[i] == b"":
                i += 1
                continue
            recs.append(lines[i + 1])
            i += 4
    else:
        cur = None
        for l in lines[i:]:
            if l[:1] == b">":
                if cur is not None:
                    recs.append(cur)
                cur = b""
            elif cur is not None:
                cur += l
        if cur is not None:
            recs.append(cur)
    return recs


def _fnv(recs):
    h = 1469598103934665603
    for r in recs:
        for b in r + struct.pack("<Q", len(r)):
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_fastx_reader_and_threaded_feed_agree(host, tmp_path):
    import gzip
    host.sylph_host_fastx_digest.argtypes = [C.c_char_p, C.c_int] + [C.POINTER(C.c_uint64)] * 4
    rng = np.random.default_rng(3)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), size=int(n))) for n in
            list(rng.integers(0, 400, size=3000)) + [0, 1, 70000]]
    fq = b"".join(b"@r%d some text\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs))
    fa = b"".join(b">c%d\n" % i + b"".join(s[j:j + 60] + b"\n" for j in range(0, len(s), 60)) for i, s in enumerate(seqs))
    cases = {"a.fq": fq, "b.fastq": fq.replace(b"\n", b"\r\n"), "c.fa": fa, "d.fasta": fa.replace(b"\n>", b"\n\n>"),
             "f.fa": b">only header\n"}
    for name, text in cases.items():
        for gz in (False, True):
            path = tmp_path / (name + (".gz" if gz else ""))
            path.write_bytes(gzip.compress(text, 1) if gz else text)
            exp = _py_records(text)
            if name == "a.fq" and not gz:
                # the very first byte decides the format (needletail's parse_fastx_reader): a leading blank line is an error
                blank = tmp_path / "leading_blank.fq"
                blank.write_bytes(b"\n" + text)
                v = [C.c_uint64(0) for _ in range(4)]
                assert host.sylph_host_fastx_digest(str(blank).encode(), 0, *[C.byref(x) for x in v]) != 0
            got = []
            for threaded in (0, 1):
                v = [C.c_uint64(0) for _ in range(4)]
                assert host.sylph_host_fastx_digest(str(path).encode(), threaded, *[C.byref(x) for x in v]) == 0
                got.append(tuple(int(x.value) for x in v))
            assert got[0] == got[1], (name, gz)
            assert got[0][0] == len(exp) and got[0][1] == 0 and got[0][2] == sum(map(len, exp)), (name, gz, got[0][:3])
            assert got[0][3] == _fnv(exp), (name, gz)
    # malformed input: both paths must report the same number of records and errors, and terminate
    bad = tmp_path / "bad.fq"
    bad.write_bytes(fq[:5000] + b"@broken\nACGT\nnot a plus line\nIIII\n" + fq[5000:9000])
    got = []
    for threaded in (0, 1):
        v = [C.c_uint64(0) for _ in range(4)]
        assert host.sylph_host_fastx_digest(str(bad).encode(), threaded, *[C.byref(x) for x in v]) == 0
        got.append(tuple(int(x.value) for x in v))
    assert got[0] == got[1] and got[0][1] >= 1
    v = [C.c_uint64(0) for _ in range(4)]
    assert host.sylph_host_fastx_digest(str(tmp_path / "missing.fq").encode(), 0, *[C.byref(x) for x in v]) == -1
    # parse_fastx_file fails up front on an empty file and on a first byte that is neither '>' nor '@' (callers then warn and
    # skip, sketch.rs:911-914, instead of writing an empty sketch); a quality line of the wrong length is a malformed record
    for name, text in (("empty.fq", b""), ("blank.fq", b"\n\n"), ("text.fq", b"hello\n@r\nACGT\n+\nIIII\n")):
        for gz in (False, True):
            path = tmp_path / (name + (".gz" if gz else ""))
            path.write_bytes(gzip.compress(text, 1) if gz else text)
            for threaded in (0, 1):
                assert host.sylph_host_fastx_digest(str(path).encode(), threaded, *[C.byref(x) for x in v]) == -1, (name, gz, threaded)
    short_q = tmp_path / "shortq.fq"
    short_q.write_bytes(b"@a\nACGTACGT\n+\nIIIIIIII\n@b\nACGTACGT\n+\nIIII\n@c\nACGT\n+\nIIII\n")
    for threaded in (0, 1):
        assert host.sylph_host_fastx_digest(str(short_q).encode(), threaded, *[C.byref(x) for x in v]) == 0
        assert v[0].value >= 1 and v[1].value >= 1          # record a parsed, record b reported as an error


def test_block_parallel_fastq_index_matches_the_sequential_reader(host, tmp_path):
    """feed.cpp FastqIndex: an uncompressed 4-line FASTQ cut into byte ranges and indexed by several threads must yield exactly
    the records of the sequential reader (same FNV digest over sequences and lengths) — with quality lines that start with
    '@' (the classic boundary trap), CRLF, a missing final newline, trailing blank lines, empty sequences — and must DECLINE
    (ok = 0, so that the drivers fall back to the sequential reader's needletail semantics) on anything irregular."""
    host.sylph_host_fastx_digest.argtypes = [C.c_char_p, C.c_int] + [C.POINTER(C.c_uint64)] * 4
    host.sylph_host_fastq_index_digest.argtypes = [C.c_char_p, C.c_uint, C.POINTER(C.c_int)] + [C.POINTER(C.c_uint64)] * 3
    rng = np.random.default_rng(8)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=int(n))) for n in rng.integers(0, 300, size=40000)]

    def qual(s, i):
        q = bytearray(rng.integers(33, 74, size=len(s), dtype=np.uint8).tobytes())
        if q and i % 3 == 0:
            q[0] = ord("@")                     # quality lines beginning with '@'
        if len(q) > 1 and i % 5 == 0:
            q[0] = ord("+")
        return bytes(q)
    fq = b"".join(b"@r%d/1 x\n%s\n+\n%s\n" % (i, s, qual(s, i)) for i, s in enumerate(seqs))
    assert len(fq) > 6 << 20                    # several 1 MiB ranges per thread
    good = {"lf.fq": fq, "crlf.fq": fq.replace(b"\n", b"\r\n"), "noeol.fq": fq[:-1], "blank_tail.fq": fq + b"\n\r\n\n",
            "plus_comment.fq": fq.replace(b"\n+\n", b"\n+comment\n"), "tiny.fq": b"@a\nACGT\n+\nIIII\n", "one_empty.fq": b"@a\n\n+\n\n"}
    for name, text in good.items():
        path = tmp_path / name
        path.write_bytes(text)
        v = [C.c_uint64(0) for _ in range(4)]
        assert host.sylph_host_fastx_digest(str(path).encode(), 0, *[C.byref(x) for x in v]) == 0
        assert v[1].value == 0
        for threads in (1, 3, 16, 3 | 0x80000000):     # (bit 31: text first, index afterwards — the two steps of the device route's gzip handling)
            ok = C.c_int(0)
            w = [C.c_uint64(0) for _ in range(3)]
            assert host.sylph_host_fastq_index_digest(str(path).encode(), threads, C.byref(ok), *[C.byref(x) for x in w]) == 0
            assert ok.value == 1, (name, threads)
            assert (w[0].value, w[1].value, w[2].value) == (v[0].value, v[2].value, v[3].value), (name, threads)
    import gzip
    cut = fq.index(b"\n@r20000/1")
    bad = {"blank_inside.fq": fq[:cut] + b"\n" + fq[cut:], "short_qual.fq": fq[:cut - 1] + fq[cut:], "fasta.fa": b">a\nACGT\n" * 10,
           "gz.fq.gz": gzip.compress(fq[:100000], 1), "multiline.fq": b"@a\nACGT\nACGT\n+\nIIII\nIIII\n" * 5, "trunc.fq": fq[:cut + 30],
           "empty.fq": b"", "noplus.fq": b"@a\nACGT\n-\nIIII\n"}
    for name, text in bad.items():
        path = tmp_path / name
        path.write_bytes(text)
        for threads in (1, 4, 4 | 0x80000000):
            ok = C.c_int(1)
            w = [C.c_uint64(0) for _ in range(3)]
            assert host.sylph_host_fastq_index_digest(str(path).encode(), threads, C.byref(ok), *[C.byref(x) for x in w]) == 0
            assert ok.value == 0, (name, threads)


def test_blocked_gzip_is_inflated_in_parallel_and_indexed(host, tmp_path, monkeypatch):
    """feed.cpp: a BGZF file's members are located from their headers, inflated by all parse threads (libdeflate when the
    system has it, zlib otherwise — both are run) and indexed like a plain FASTQ; gzip.decompress accepts the same bytes (the
    file IS a valid multi-member gzip); a damaged member or an ordinary one-stream gzip is declined (sequential reader)."""
    import gzip
    host.sylph_host_fastx_digest.argtypes = [C.c_char_p, C.c_int] + [C.POINTER(C.c_uint64)] * 4
    host.sylph_host_fastq_index_digest.argtypes = [C.c_char_p, C.c_uint, C.POINTER(C.c_int)] + [C.POINTER(C.c_uint64)] * 3
    rng = np.random.default_rng(81)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(n))) for n in rng.integers(1, 300, size=60000)]
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs))
    plain = tmp_path / "p.fq"
    plain.write_bytes(fq)
    z = bgzf_compress(fq)
    assert gzip.decompress(z) == fq and len(z) // 65536 > 40
    bg = tmp_path / "b.fq.gz"
    bg.write_bytes(z)
    v = [C.c_uint64(0) for _ in range(4)]
    assert host.sylph_host_fastx_digest(str(plain).encode(), 0, *[C.byref(x) for x in v]) == 0
    u = [C.c_uint64(0) for _ in range(4)]
    assert host.sylph_host_fastx_digest(str(bg).encode(), 0, *[C.byref(x) for x in u]) == 0          # the sequential gz reader
    assert [x.value for x in u] == [x.value for x in v]

    def index(path, threads):
        ok = C.c_int(0)
        w = [C.c_uint64(0) for _ in range(3)]
        assert host.sylph_host_fastq_index_digest(str(path).encode(), threads, C.byref(ok), *[C.byref(x) for x in w]) == 0
        return ok.value, (w[0].value, w[1].value, w[2].value)
    for threads in (1, 5, 16, 5 | 0x80000000):
        assert index(bg, threads) == (1, (v[0].value, v[2].value, v[3].value))
    # a flipped byte inside one member's deflate data: CRC / inflate failure -> declined
    bad = bytearray(z)
    bad[len(z) // 2] ^= 0x55
    (tmp_path / "bad.fq.gz").write_bytes(bytes(bad))
    assert index(tmp_path / "bad.fq.gz", 4)[0] == 0
    (tmp_path / "trunc.fq.gz").write_bytes(z[:len(z) // 2])
    assert index(tmp_path / "trunc.fq.gz", 4)[0] == 0
    (tmp_path / "plain.fq.gz").write_bytes(gzip.compress(fq[:200000], 1))
    assert index(tmp_path / "plain.fq.gz", 4)[0] == 0
    # the zlib inflater (what runs where libdeflate is not installed): a fresh process, the choice is made once
    import subprocess
    import sys
    code = ("import ctypes as C, sys; h = C.CDLL(sys.argv[1]); ok = C.c_int(0); w = [C.c_uint64(0) for _ in range(3)];"
            "h.sylph_host_fastq_index_digest.argtypes = [C.c_char_p, C.c_uint, C.POINTER(C.c_int)] + [C.POINTER(C.c_uint64)] * 3;"
            "h.sylph_host_fastq_index_digest(sys.argv[2].encode(), 4, C.byref(ok), *[C.byref(x) for x in w]); print(ok.value, w[0].value, w[1].value, w[2].value)")
    r = subprocess.run([sys.executable, "-c", code, host._name, str(bg)], capture_output=True, text=True, env=dict(os.environ, SYLPH_HIP_NO_LIBDEFLATE="1"))
    assert r.stdout.split() == ["1", str(v[0].value), str(v[2].value), str(v[3].value)], r.stdout + r.stderr


def test_parallel_2bit_packer_matches_byte_to_seq(host):
    """Pack2Bit (host/pack2bit.cpp): records packed by several writers into one 2-bit stream — parts that begin and end at any
    base offset, shared bytes merged afterwards, the AVX2 path and the table path — against BYTE_TO_SEQ spelled out here and
    against the library's own sylph_pack_2bit."""
    import sylph_amd as S
    host.sylph_host_pack_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64]
    code = np.zeros(256, dtype=np.uint8)
    for ch, v in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"TtUu", 3)):
        for b in ch:
            code[b] = v
    code[1], code[2], code[3] = 1, 2, 3
    rng = np.random.default_rng(12)
    for trial in range(60):
        n_rec = int(rng.integers(0, 40))
        kind = trial % 3
        recs = []
        for _ in range(n_rec):
            L = int(rng.integers(0, 260))
            if kind == 0:
                r = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)            # the fast path only
            elif kind == 1:
                r = rng.choice(np.frombuffer(b"ACGTacgtNnUu-", dtype=np.uint8), size=L)     # odd bytes inside 32-base groups
            else:
                r = rng.integers(0, 256, size=L).astype(np.uint8)                           # the whole byte alphabet
            recs.append(r.astype(np.uint8))
        bases = np.concatenate(recs + [np.zeros(0, np.uint8)]).astype(np.uint8)
        off = np.zeros(n_rec + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in recs])
        total = int(off[-1])
        c = code[bases]
        pad = (-total) % 4
        c4 = np.concatenate([c, np.zeros(pad, np.uint8)]).reshape(-1, 4)
        expect = ((c4[:, 0] << 6) | (c4[:, 1] << 4) | (c4[:, 2] << 2) | c4[:, 3]).astype(np.uint8)
        assert np.array_equal(S.pack_2bit(bases)[:len(expect)], expect)
        for parts in (1, 2, 3, 7):
            out = np.full(len(expect) + 8, 0xAB, dtype=np.uint8)                           # stale bytes of a reused buffer
            # bytes behind the stream: readable (the vector path of a record's last bases loads them) or not (table path)
            for slack in (64, 0):
                padded = np.concatenate([bases, rng.integers(0, 256, size=64).astype(np.uint8)])
                out[:] = 0xAB
                host.sylph_host_pack_records(padded.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), n_rec, parts,
                                             out.ctypes.data_as(C.c_void_p), total + slack)
                assert np.array_equal(out[:len(expect)], expect), (trial, parts, slack)
            # bytes wholly inside the stream must match; the last (partial) byte too — the writers zero-fill its unused bits
            assert np.array_equal(out[:len(expect)], expect), (trial, parts)


def test_parallel_gunzip_of_single_member_gzip(host, tmp_path):
    """host/pgunzip.cpp: an ORDINARY gzip file inflated by several threads — block starts found by search,
    every stretch decoded without its 32 KiB window (symbolic back-references), windows handed down the chain, the member's CRC as
    the final word.  Against Python's gzip for every compression level's block structure (stored / fixed / dynamic blocks), stretch
    sizes down to 64 KiB (many block-start searches), CRLF and multi-line content; and it must DECLINE — return 0, the feed then
    reads sequentially — whatever it cannot prove: two members, a damaged byte, bytes no FASTQ holds, a file too small to split."""
    import gzip
    host.sylph_host_pgunzip.argtypes = [C.c_char_p, C.c_uint, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64]
    rng = np.random.default_rng(8)

    def fastq(n_reads, crlf=False):
        nl = b"\r\n" if crlf else b"\n"
        out = bytearray()
        base = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=200_000)
        for i in range(n_reads):
            s = int(rng.integers(0, 199_000))
            L = int(rng.integers(50, 251))
            q = rng.integers(35, 45, size=L, dtype=np.uint8).tobytes()
            out += b"@r%d lane:%d" % (i, i % 8) + nl + base[s:s + L].tobytes() + nl + b"+" + nl + q + nl
        return bytes(out)

    def run(path, threads, expect):
        n, c = C.c_uint64(0), C.c_uint32(0)
        buf = (C.c_uint8 * max(1, len(expect) if expect is not None else 1))()
        rc = host.sylph_host_pgunzip(str(path).encode(), threads, C.byref(n), C.byref(c), buf, len(buf))
        if expect is None:
            return rc
        assert rc == 1 and n.value == len(expect) and bytes(buf) == expect, (str(path), threads, rc, n.value, len(expect))
        return rc

    os.environ["SYLPH_HIP_PGZ_STRETCH"] = "65536"
    try:
        data = fastq(40_000)
        for lvl in (1, 4, 6, 9):
            p = tmp_path / f"l{lvl}.fq.gz"
            p.write_bytes(gzip.compress(data, compresslevel=lvl))
            for passes in ("1", "2"):          # one decode into 16-bit cells (CPU-poor hosts) / two cache-resident passes (many cores)
                os.environ["SYLPH_HIP_PGZ_PASSES"] = passes
                for thr in (2, 5, 16):
                    run(p, thr, data)
        os.environ.pop("SYLPH_HIP_PGZ_PASSES", None)
        crlf = fastq(8_000, crlf=True)
        p = tmp_path / "crlf.fq.gz"
        p.write_bytes(gzip.compress(crlf, compresslevel=6))
        run(p, 8, crlf)
        # level 0: stored blocks only — no dynamic block to start a stretch at: declined
        p = tmp_path / "stored.fq.gz"
        p.write_bytes(gzip.compress(data[:2_000_000], compresslevel=0))
        assert run(p, 8, None) == 0
        # two members, a flipped byte in the middle, a binary payload, a tiny file: all declined
        p = tmp_path / "two.fq.gz"
        p.write_bytes(gzip.compress(data[:3_000_000]) + gzip.compress(data[3_000_000:6_000_000]))
        assert run(p, 8, None) == 0
        good = bytearray(gzip.compress(data, compresslevel=6))
        good[len(good) // 2] ^= 0x40
        p = tmp_path / "bad.fq.gz"
        p.write_bytes(bytes(good))
        assert run(p, 8, None) == 0
        p = tmp_path / "bin.gz"
        p.write_bytes(gzip.compress(rng.integers(0, 256, size=3_000_000, dtype=np.uint8).tobytes() + data[:1_000_000], compresslevel=6))
        assert run(p, 8, None) == 0
        p = tmp_path / "tiny.fq.gz"
        p.write_bytes(gzip.compress(data[:20_000]))
        assert run(p, 8, None) == 0
        assert run(tmp_path / "nonexistent.gz", 8, None) == -1
    finally:
        os.environ.pop("SYLPH_HIP_PGZ_STRETCH", None)


def test_feed_thread_counts_follow_the_cpus_the_process_may_use():
    """effective_cpus() (host/feed.cpp): hardware threads cut down to the affinity mask and the cgroup CPU quota — the GPU box's
    container shows 256 hardware threads under a quota of 16 CPUs, and thread counts taken from the former got the whole process
    frozen for most of every scheduler period.  Checked in child processes (the values are read once per process): the quota of this
    very container, an affinity mask of two CPUs, and the SYLPH_HIP_CPUS override; the parse-thread count derived from them."""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "sylph_amd", "libsylph_host.so")
    code = ("import ctypes, os; L = ctypes.CDLL(%r); L.sylph_host_effective_cpus.restype = ctypes.c_uint; L.sylph_host_parse_threads.restype = ctypes.c_uint; "
            "print(L.sylph_host_effective_cpus(), L.sylph_host_parse_threads())" % lib)

    def ask(env=None, pre=""):
        e = dict(os.environ)
        for k in ("SYLPH_HIP_CPUS", "SYLPH_HIP_PARSE_THREADS"):
            e.pop(k, None)
        e.update(env or {})
        out = subprocess.run([sys.executable, "-c", pre + code], capture_output=True, text=True, env=e, check=True).stdout.split()
        return int(out[0]), int(out[1])

    hw = os.cpu_count() or 1
    expect = min(hw, len(os.sched_getaffinity(0)))
    try:                                                           # this container's own quota, read the way the library reads it
        a, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            expect = min(expect, max(1, -(-int(a) // int(per))))
    except Exception:
        pass
    cpus, threads = ask()
    assert cpus == expect and 1 <= cpus <= hw
    assert threads == (min(64, max(2, cpus + cpus // 2)) if cpus < hw else min(64, max(min(hw, 8), hw // 4)))
    if expect >= 2:
        two = sorted(os.sched_getaffinity(0))[:2]
        cpus2, threads2 = ask(pre="import os; os.sched_setaffinity(0, %r); " % (set(two),))
        assert cpus2 == 2 and threads2 == (3 if hw > 2 else threads2)
    assert ask({"SYLPH_HIP_CPUS": "5"})[0] == 5
    assert ask({"SYLPH_HIP_PARSE_THREADS": "7"})[1] == 7
