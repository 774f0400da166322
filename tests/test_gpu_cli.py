"""GPU tests of the C++ host + `sylph-hip` command: the reference's own integration assertions
(tests/integration_test.rs: output files and names, line counts, raw-vs-presketched stdout equality, exit codes)
re-expressed on committed fixtures, plus a row-by-row comparison of the TSV with the oracle's numbers."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pyref as P

from .helpers import ACGT, bgzf_compress, random_seq, revcomp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "sylph_amd", "sylph-hip")


def run(*args, check=True, accept_exact=True, env_extra=None):
    # (most tests here compare with the oracle's EXACT pair set: SYLPH_HIP_EXACT_DEDUP=1 selects it whatever --fpr says; the
    #  reference's default — the filter — is test_paired_reads_are_deduplicated_as_the_reference_does_by_default's subject)
    env = dict(os.environ)
    env.pop("SYLPH_HIP_EXACT_DEDUP", None)
    if accept_exact:
        env["SYLPH_HIP_EXACT_DEDUP"] = "1"
    env.update(env_extra or {})
    p = subprocess.run([BIN] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    if check:
        assert p.returncode == 0, p.stderr[-3000:]
    return p


def write_fasta(path, records, gz=True, width=70):
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for name, seq in records:
            f.write(b">" + name + b"\n")
            s = bytes(seq)
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + b"\n")


def write_fastq(path, reads, gz=False, prefix=b"r"):
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b"@" + prefix + str(i).encode() + b" extra\n" + bytes(s) + b"\n+\n" + b"I" * len(s) + b"\n")


@pytest.fixture(scope="module")
def data(tmp_path_factory, golden_dir):
    d = tmp_path_factory.mktemp("cli")
    z = np.load(os.path.join(golden_dir, "ecoli_slices.npz"))
    rng = np.random.default_rng(123)
    genomes = {}
    for gi, name in enumerate(("EC590", "K12", "O157")):
        b, off = z[f"g{gi}_bases"], z[f"g{gi}_off"]
        recs = [(f"{name}_contig{i} test genome".encode(), b[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
        p = str(d / f"{name}.fasta.gz")
        write_fasta(p, recs)
        genomes[name] = (p, recs)
    unrelated = random_seq(rng, 120000)
    genomes["rand"] = (str(d / "rand.fa"), [(b"random_genome", unrelated)])
    write_fasta(genomes["rand"][0], genomes["rand"][1], gz=False)
    # paired reads: 6x of the K12 slice + 1x of the random genome, 1 % errors, some exact duplicate pairs
    def sim(g, n):
        m1, m2 = [], []
        for _ in range(n):
            ins = int(rng.integers(200, 400))
            s = int(rng.integers(0, len(g) - ins))
            frag = g[s:s + ins] if rng.random() < 0.5 else revcomp(g[s:s + ins])
            a, b = frag[:100].copy(), revcomp(frag)[:100].copy()
            for m in (a, b):
                e = rng.random(100) < 0.01
                m[e] = rng.choice(ACGT, size=int(e.sum()))
            m1.append(a); m2.append(b)
        return m1, m2
    k12 = genomes["K12"][1][0][1]
    a1, a2 = sim(k12, 9000)
    b1, b2 = sim(unrelated, 600)
    c1, c2 = sim(genomes["EC590"][1][0][1], 6000)   # a second, closely related strain: shared k-mers get reassigned
    m1, m2 = a1 + b1 + c1, a2 + b2 + c2
    for i in range(300):
        j = int(rng.integers(0, len(m1)))
        m1.append(m1[j]); m2.append(m2[j])
    write_fastq(str(d / "s_1.fq"), m1)
    write_fastq(str(d / "s_2.fq"), m2)
    write_fastq(str(d / "single.fastq.gz"), m1, gz=True)
    return dict(dir=d, genomes=genomes, m1=m1, m2=m2)


def expected_rows(data, sample, paired, pseudotax, genome_order):
    """Recompute the TSV rows with the oracle (CI columns excluded: the bootstrap is parity-unpinned)."""
    gs = []
    for name in genome_order:
        path, recs = data["genomes"][name]
        b, off = O.concat([bytes(r[1]) for r in recs])
        g = O.sketch_genome(b, off)
        gs.append(dict(path=path, contig=recs[0][0].decode(), kmers=g["genome_kmers"], tracked=g["tracked"], gn_size=g["gn_size"]))
    recs = [x for p in zip(data["m1"], data["m2"]) for x in p] if paired else data["m1"]
    b, off = O.concat([bytes(r) for r in recs])
    s = O.sketch_reads(b, off, paired=paired)
    db = np.concatenate([g["kmers"] for g in gs])
    goff = np.zeros(len(gs) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g["kmers"]) for g in gs])
    cc, covs, _ = O.contain(s["kmers"], s["counts"], db, goff)
    min_ani = 0.95 if pseudotax else 0.90
    res = []
    for i, g in enumerate(gs):
        if cc[i] == 0:
            continue
        st = O.stats(covs[i], len(g["kmers"]), min_ani=min_ani)
        if st.passed:
            res.append(dict(i=i, st=st, cc=int(cc[i]), lost=None))
    if pseudotax:
        smap = dict(zip(s["kmers"].tolist(), s["counts"].tolist()))
        winner = {}
        for r in res:
            g = gs[r["i"]]
            for km in list(g["kmers"].tolist()) + list(g["tracked"].tolist()):
                if km not in winner or r["st"].final_est_ani > winner[km][0]:
                    winner[km] = (r["st"].final_est_ani, r["i"])
        res2 = []
        for r in res:
            g = gs[r["i"]]
            cv, lost = [], 0
            for km in g["kmers"].tolist():
                c = smap.get(km, 0)
                if c == 0:
                    continue
                if winner[km][1] != r["i"]:
                    lost += 1
                    continue
                cv.append(c)
            if not cv:
                continue
            st = O.stats(np.array(cv, dtype=np.uint32), len(g["kmers"]), min_ani=min_ani)
            if st.passed and (r["cc"] - len(cv)) < (0.99 ** 31) * len(g["kmers"]):
                res2.append(dict(i=r["i"], st=st, cc=len(cv), lost=lost))
        res = res2
        tot = sum(r["st"].final_est_cov for r in res)
        tots = sum(r["st"].final_est_cov * gs[r["i"]]["gn_size"] for r in res)
        for r in res:
            r["rel"] = r["st"].final_est_cov / tot * 100.0
            r["seq"] = r["st"].final_est_cov * gs[r["i"]]["gn_size"] / tots * 100.0
        res.sort(key=lambda r: -r["rel"])
    else:
        res.sort(key=lambda r: -r["st"].final_est_ani)
    rows = []
    for r in res:
        st, g = r["st"], gs[r["i"]]
        lam = "%.3f" % st.lambda_ if st.lambda_status == 2 else ("HIGH" if st.lambda_status == 1 else "LOW")
        common = ["%.2f" % min(st.final_est_ani * 100, 100.0), "%.3f" % st.final_est_cov, None, lam, None, "%.0f" % st.median_cov,
                  "%.3f" % st.mean_cov, "%d/%d" % (r["cc"], len(g["kmers"])), "%.2f" % (st.naive_ani * 100)]
        if pseudotax:
            rows.append([sample, g["path"], "%.4f" % r["rel"], "%.4f" % r["seq"]] + common + [str(r["lost"]), g["contig"]])
        else:
            rows.append([sample, g["path"]] + common + [g["contig"]])
    return rows


def compare(stdout, rows, pseudotax):
    lines = stdout.strip().split("\n")
    assert lines[0].startswith("Sample_file\tGenome_file")
    assert len(lines) == 1 + len(rows), stdout
    ci_cols = (6, 8) if pseudotax else (4, 6)
    for line, exp in zip(lines[1:], rows):
        got = line.split("\t")
        assert len(got) == len(exp)
        for j, (a, b) in enumerate(zip(got, exp)):
            if j in ci_cols:
                assert a == "NA-NA" or "-" in a
            else:
                assert a == b, (j, a, b, line)


def test_sketch_outputs_and_names(data):
    d = data["dir"]
    g = data["genomes"]
    out = d / "out1"
    run("sketch", g["EC590"][0], g["K12"][0], g["O157"][0], g["rand"][0], "-o", out / "db", "-d", out / "samples", "-r", d / "single.fastq.gz",
        "-1", d / "s_1.fq", "-2", d / "s_2.fq")
    assert (out / "db.syldb").exists()
    assert (out / "samples" / "single.fastq.gz.sylsp").exists()
    assert (out / "samples" / "s_1.fq.paired.sylsp").exists()        # integration_test.rs:78,313
    # sample names (-S): sketches are renamed (integration_test.rs:298-375)
    run("sketch", "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-r", d / "single.fastq.gz", "-S", "pairA", "singleB", "-d", out / "named")
    assert (out / "named" / "pairA.paired.sylsp").exists() and (out / "named" / "singleB.sylsp").exists()
    assert run("sketch", "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-S", "a", "b", "-d", out / "named2", check=False).returncode == 1


def test_error_exit_codes(data):
    d = data["dir"]
    g = data["genomes"]
    assert run("sketch", g["K12"][0], "--fpr", "2", "-o", d / "e" / "db", check=False).returncode == 1       # integration_test.rs:419
    assert run("sketch", "-1", d / "s_1.fq", "-d", d / "e", check=False).returncode == 1                     # :445
    assert run("sketch", check=False).returncode == 1
    run("sketch", g["K12"][0], "--disable-profiling", "-o", d / "e" / "noprof")
    run("sketch", "-r", d / "single.fastq.gz", "-d", d / "e")
    assert run("profile", d / "e" / "noprof.syldb", d / "e" / "single.fastq.gz.sylsp", check=False).returncode == 1   # :234
    assert run("query", d / "e" / "noprof.syldb", d / "e" / "single.fastq.gz.sylsp").returncode == 0
    assert run("profile", d / "e" / "single.fastq.gz.sylsp", check=False).returncode == 1                    # no genomes
    assert run("profile", d / "e" / "noprof.syldb", check=False).returncode == 1                             # no reads


def test_query_and_profile_rows_match_oracle(data):
    d = data["dir"]
    g = data["genomes"]
    order = ["EC590", "K12", "O157", "rand"]
    out = d / "out2"
    run("sketch", *[g[n][0] for n in order], "-o", out / "db", "-d", out, "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-r", d / "single.fastq.gz")
    sample = str(d / "s_1.fq")
    q = run("query", out / "db.syldb", out / "s_1.fq.paired.sylsp")
    rows = expected_rows(data, sample, True, False, order)
    assert len(rows) >= 2
    compare(q.stdout, rows, False)
    p = run("profile", out / "db.syldb", out / "s_1.fq.paired.sylsp")
    prow = expected_rows(data, sample, True, True, order)
    assert len(prow) >= 2 and any(int(r[13]) > 0 for r in prow)       # some k-mers really were reassigned
    compare(p.stdout, prow, True)
    # single-end sample, gz input, -o output file
    run("query", out / "db.syldb", out / "single.fastq.gz.sylsp", "-o", out / "q.tsv")
    compare(open(out / "q.tsv").read(), expected_rows(data, str(d / "single.fastq.gz"), False, False, order), False)


def test_raw_inputs_equal_presketched(data):
    """integration_test.rs:248-295 and :465-501: profiling raw fasta/fastq gives the same stdout as profiling sketches."""
    d = data["dir"]
    g = data["genomes"]
    out = d / "out3"
    order = ["EC590", "K12", "O157", "rand"]
    run("sketch", *[g[n][0] for n in order], "-o", out / "db", "-d", out, "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-r", d / "single.fastq.gz")
    a = run("profile", out / "db.syldb", out / "s_1.fq.paired.sylsp", out / "single.fastq.gz.sylsp")
    b = run("profile", *[g[n][0] for n in order], "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-r", d / "single.fastq.gz")
    # raw inputs are processed first, then sketches (contain.rs:257-258): compare as sets of rows
    assert sorted(a.stdout.strip().split("\n")) == sorted(b.stdout.strip().split("\n"))
    assert len(a.stdout.strip().split("\n")) >= 4
    # mixing a raw genome with a database, -l list file
    lst = out / "list.txt"
    lst.write_text("\n".join([str(out / "db.syldb"), str(out / "s_1.fq.paired.sylsp")]) + "\n")
    c = run("query", "-l", lst)
    assert c.stdout == run("query", out / "db.syldb", out / "s_1.fq.paired.sylsp").stdout
    # -i individual records: one sketch per contig
    run("sketch", g["O157"][0], "-i", "-o", out / "indiv")
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "sylph_amd", "libsylph_host.so"))
    L.sylph_host_read_syldb.restype = C.c_void_p
    L.sylph_host_read_syldb.argtypes = [C.c_char_p]
    L.sylph_host_syldb_size.restype = C.c_uint64
    L.sylph_host_syldb_size.argtypes = [C.c_void_p]
    h = L.sylph_host_read_syldb(str(out / "indiv.syldb").encode())
    assert L.sylph_host_syldb_size(h) == len(g["O157"][1])


def expected_floats(data, sample, paired, pseudotax, genome_order, unknown=False, seq_id=None):
    """The float columns recomputed by the INDEPENDENT restatement (oracle/pyref.py: Python ints/dicts, scipy Poisson tail —
    no code shared with sylph_amd/host/inference.cpp or with the C++ oracle): statistics, winner table, reassignment
    threshold, abundances, sort order."""
    cache = data.setdefault("_pyref", {})          # pure-Python sketching takes seconds: once per fixture
    gs = []
    for name in genome_order:
        path, recs = data["genomes"][name]
        if ("g", name) not in cache:
            cache[("g", name)] = P.sketch_genome([bytes(r[1]) for r in recs], 200, 31)
        g = cache[("g", name)]
        gs.append(dict(path=path, kmers=g["genome_kmers"], tracked=g["tracked"], gn_size=g["gn_size"]))
    if ("s", paired) not in cache:
        if paired:
            cache[("s", paired)] = P.sketch_pair_sequences([bytes(r) for r in data["m1"]], [bytes(r) for r in data["m2"]], 200, 31)
        else:
            cache[("s", paired)] = P.sketch_sequences_needle([bytes(r) for r in data["m1"]], 200, 31)
    counts = cache[("s", paired)]["kmer_counts"]
    mean_len = cache[("s", paired)]["mean_read_length"]
    min_ani = 0.95 if pseudotax else 0.90
    kmer_id = None
    if unknown:      # -u: read identity from -I, else from the table walked in ascending k-mer order (the host's order)
        kmer_id = (seq_id / 100.0) ** 31 if seq_id is not None else P.get_kmer_identity([counts[km] for km in sorted(counts)], 31, mean_len)

    def true_cov(rs):
        if kmer_id is not None:
            for r, cv in zip(rs, P.estimate_true_cov([r["st"]["final_est_cov"] for r in rs], kmer_id, mean_len, 31)):
                r["st"] = dict(r["st"], final_est_cov=cv)
    res = []
    for i, g in enumerate(gs):
        pr = P.probe(g["kmers"], counts)
        if pr is None or not pr[1]:
            continue
        st = P.stats(pr[0], pr[1], len(g["kmers"]), min_ani=min_ani)
        if st["passed"]:
            res.append(dict(i=i, st=st, lost=None))
    true_cov(res)
    if pseudotax:
        winner = P.winner_table([(r["i"], r["st"]["final_est_ani"], gs[r["i"]]["kmers"], gs[r["i"]]["tracked"]) for r in res])
        res2 = []
        for r in res:
            g = gs[r["i"]]
            cc, covs, lost = P.probe(g["kmers"], counts, winner=winner, me=r["i"])
            if not covs:
                continue
            st = P.stats(cc, covs, len(g["kmers"]), min_ani=min_ani)
            if st["passed"] and P.derep_if_reassign_threshold(r["st"]["contain_count"], cc, len(g["kmers"])):
                res2.append(dict(i=r["i"], st=st, lost=lost))
        res = res2
        true_cov(res)
        explained = 1.0
        if unknown:
            explained = P.estimate_covered_bases([gs[r["i"]]["gn_size"] for r in res], [r["st"]["final_est_cov"] for r in res], 200,
                                                 sum(counts.values()), mean_len, 31)
        tot = sum(r["st"]["final_est_cov"] for r in res)
        tots = sum(r["st"]["final_est_cov"] * gs[r["i"]]["gn_size"] for r in res)
        for r in res:
            r["rel"] = r["st"]["final_est_cov"] / tot * 100.0
            r["seq"] = r["st"]["final_est_cov"] * gs[r["i"]]["gn_size"] / tots * 100.0 * explained
        res.sort(key=lambda r: -r["rel"])
    else:
        res.sort(key=lambda r: -r["st"]["final_est_ani"])
    return res, gs


def compare_f64(stdout, res, gs, pseudotax, rel=1e-6):
    """--debug-f64 rows against the independent restatement: every float column at `rel` (the north_star's 1e-6)."""
    lines = stdout.strip().split("\n")
    assert len(lines) == 1 + len(res), stdout
    o = 2 if pseudotax else 0
    for line, r in zip(lines[1:], res):
        got = line.split("\t")
        st = r["st"]
        assert got[1] == gs[r["i"]]["path"]
        close = lambda a, b: abs(float(a) - b) <= rel * abs(b)
        if pseudotax:
            assert close(got[2], r["rel"]) and close(got[3], r["seq"]), line
            assert int(got[o + 11]) == r["lost"]
        assert close(got[o + 2], st["final_est_ani"] * 100.0), (line, st)
        assert close(got[o + 3], st["final_est_cov"]), (line, st)
        if st["lambda_status"] == "LAMBDA":
            assert close(got[o + 5], st["lambda_"])
            lo, hi = got[o + 4].split("-")      # bootstrap CI (fastrand: parity unpinned) — present, ordered, brackets sanely
            if lo != "NA":
                assert float(lo) <= float(hi)
        else:
            assert got[o + 5] == st["lambda_status"]
        assert close(got[o + 7], st["median_cov"]) and close(got[o + 8], st["mean_cov"])
        assert got[o + 9] == "%d/%d" % (st["contain_count"], st["n_kmers"])
        assert close(got[o + 10], st["naive_ani"] * 100.0)


def test_float_columns_1e6_vs_independent_restatement(data):
    """north_star: 'derived ANI/abundance floats within 1e-6'.  The printed TSV has 2-4 decimals, so the CLI's --debug-f64
    mode prints %.17g and the comparison is against oracle/pyref.py, which shares no code with the host statistics."""
    d = data["dir"]
    g = data["genomes"]
    order = ["EC590", "K12", "O157", "rand"]
    out = d / "out5"
    run("sketch", *[g[n][0] for n in order], "-o", out / "db", "-d", out, "-1", d / "s_1.fq", "-2", d / "s_2.fq", "--fpr", "0",
        "-r", d / "single.fastq.gz")
    for sample, paired in ((out / "s_1.fq.paired.sylsp", True), (out / "single.fastq.gz.sylsp", False)):
        for cmd, pseudotax in (("query", False), ("profile", True)):
            p = run(cmd, out / "db.syldb", sample, "--debug-f64")
            res, gs = expected_floats(data, None, paired, pseudotax, order)
            assert len(res) >= 1
            compare_f64(p.stdout, res, gs, pseudotax)


def test_estimate_unknown_columns_vs_independent_restatement(data):
    """-u/--estimate-unknown (contain.rs:901-951, :377-408): True_cov header, coverages divided by the read k-mer identity and
    scaled by L / (L - k + 1), Sequence_abundance scaled by the share of explained bases — with -I (identity given: no
    order-dependent step anywhere) and with the automatic estimate, at 1e-6 against oracle/pyref.py."""
    d = data["dir"]
    g = data["genomes"]
    order = ["EC590", "K12", "O157", "rand"]
    out = d / "out6"
    run("sketch", *[g[n][0] for n in order], "-o", out / "db", "-d", out, "-1", d / "s_1.fq", "-2", d / "s_2.fq", "--fpr", "0",
        "-r", d / "single.fastq.gz")
    plain = run("profile", out / "db.syldb", out / "single.fastq.gz.sylsp", "--debug-f64").stdout
    for sample, paired in ((out / "s_1.fq.paired.sylsp", True), (out / "single.fastq.gz.sylsp", False)):
        for extra, seq_id in ((("-u", "-I", "98.5"), 98.5), (("-u",), None)):
            p = run("profile", out / "db.syldb", sample, "--debug-f64", *extra)
            assert p.stdout.split("\n")[0].split("\t")[5] == "True_cov"
            res, gs = expected_floats(data, None, paired, True, order, unknown=True, seq_id=seq_id)
            assert len(res) >= 1
            compare_f64(p.stdout, res, gs, True)
            q = run("query", out / "db.syldb", sample, "--debug-f64", *extra)      # query: coverages scaled, header unchanged
            assert q.stdout.split("\n")[0].split("\t")[3] == "Eff_cov"
            res, gs = expected_floats(data, None, paired, False, order, unknown=True, seq_id=seq_id)
            compare_f64(q.stdout, res, gs, False)
    with_u = run("profile", out / "db.syldb", out / "single.fastq.gz.sylsp", "--debug-f64", "-u", "-I", "98.5").stdout
    assert with_u != plain                                # the option changes the coverage and abundance columns


def sylsp_table(path):
    """(k-mers ascending, counts) of a .sylsp: the table leads the file (types.rs:145-155: u64 length, then (u64, u32) entries)"""
    raw = open(path, "rb").read()
    n = int.from_bytes(raw[:8], "little")
    t = np.frombuffer(raw, dtype=np.dtype([("k", "<u8"), ("c", "<u4")]), count=n, offset=8)
    o = np.argsort(t["k"])
    return t["k"][o], t["c"][o]


def test_paired_reads_are_deduplicated_as_the_reference_does_by_default(data):
    """a10 (sketch.rs:733-769): with default flags the pair set lives behind a cuckoo filter of --fpr 1e-4 (cmdline.rs:77; raw
    pairs in profile: contain.rs:591) — the sketch equals the oracle's model of that walk; --fpr 0, --exact-dedup and
    SYLPH_HIP_EXACT_DEDUP=1 give the exact set (the same bytes, equal to the oracle's exact sketch); another --fpr is another
    filter; --no-dedup never consults it."""
    d = data["dir"]
    recs = [x for p in zip(data["m1"], data["m2"]) for x in p]
    b, off = O.concat([bytes(r) for r in recs])
    run("sketch", "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-d", d / "w1", accept_exact=False)                    # the reference's default
    run("sketch", "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-d", d / "w2", "--fpr", "0", accept_exact=False)
    run("sketch", "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-d", d / "w3", "--exact-dedup", accept_exact=False)
    run("sketch", "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-d", d / "w4")                                        # SYLPH_HIP_EXACT_DEDUP=1
    run("sketch", "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-d", d / "w5", "--no-dedup", accept_exact=False)
    run("sketch", "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-d", d / "w6", "--fpr", "0.3", accept_exact=False)
    tab = lambda w: sylsp_table(d / w / "s_1.fq.paired.sylsp")
    for w, e in (("w1", O.sketch_reads_cuckoo_model(b, off, fpr=1e-4)), ("w2", O.sketch_reads(b, off, paired=True)),
                 ("w5", O.sketch_reads(b, off, paired=True, no_dedup=True)), ("w6", O.sketch_reads_cuckoo_model(b, off, fpr=0.3))):
        k, c = tab(w)
        assert np.array_equal(k, e["kmers"]) and np.array_equal(c, e["counts"]), w
    ref = (d / "w2" / "s_1.fq.paired.sylsp").read_bytes()
    assert (d / "w3" / "s_1.fq.paired.sylsp").read_bytes() == ref and (d / "w4" / "s_1.fq.paired.sylsp").read_bytes() == ref
    # (a sample this small leaves a filter of capacity 10^7 nearly empty: even at --fpr 0.3 hardly any test meets a false positive —
    #  the filter's own effects are test_gpu_parity.py::test_read_sketch_paired_filter_dedup's subject)
    # raw pairs in profile / query: the default filter (contain.rs:591), or the exact set on request — same rows as the sketches give
    g = data["genomes"]
    raw = run("profile", g["K12"][0], g["EC590"][0], "-1", d / "s_1.fq", "-2", d / "s_2.fq", accept_exact=False).stdout
    pre = run("profile", g["K12"][0], g["EC590"][0], d / "w1" / "s_1.fq.paired.sylsp").stdout
    assert raw == pre and len(raw.splitlines()) >= 2
    raw0 = run("profile", g["K12"][0], g["EC590"][0], "-1", d / "s_1.fq", "-2", d / "s_2.fq", "--exact-dedup", accept_exact=False).stdout
    assert raw0 == run("profile", g["K12"][0], g["EC590"][0], d / "w2" / "s_1.fq.paired.sylsp").stdout


def test_parallel_feed_equals_sequential_feed(data):
    """Uncompressed FASTQ goes through the block-parallel index (feed.cpp FastqIndex: files cut into byte ranges, batches gathered
    into page-locked memory by worker threads); SYLPH_HIP_SEQUENTIAL_FEED=1 forces the reader-thread path.  The sketches must be
    byte-identical, for single-end and paired input and for every thread count."""
    d = data["dir"]
    outs = []
    for env in ({"SYLPH_HIP_SEQUENTIAL_FEED": "1"}, {"SYLPH_HIP_PARSE_THREADS": "1"}, {"SYLPH_HIP_PARSE_THREADS": "7"}):
        o = d / ("feed_" + "_".join(env.values()))
        p = subprocess.run([BIN, "sketch", "-1", str(d / "s_1.fq"), "-2", str(d / "s_2.fq"), "-r", str(d / "s_1.fq"), "--fpr", "0", "-d", str(o)],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(((o / "s_1.fq.paired.sylsp").read_bytes(), (o / "s_1.fq.sylsp").read_bytes()))
    assert outs[0] == outs[1] == outs[2]
    assert len(outs[0][0]) > 1000
    # blocked gzip (bgzip) input: members inflated in parallel, then the same index; the sketch records the file name, so the
    # comparison is on the k-mer tables the query prints
    for m in ("1", "2"):
        (d / f"z_{m}.fq.gz").write_bytes(bgzf_compress((d / f"s_{m}.fq").read_bytes(), block=4000))
    # (round 6: by default a gzip sample's compressed bytes go to the device — SYLPH_HIP_INFLATE_DEVICE=0 keeps the host's inflate, which
    #  is what this test is about; the default road gives the same table)
    b = outs[0][0]
    n_tab = 8 + 12 * int.from_bytes(b[:8], "little")                               # the k-mer table leads the file (types.rs:145-155)
    for inflate_device, marker in (("0", "gather"), ("1", "gzip inflated on the device")):
        o = d / f"feed_bgzf_{inflate_device}"
        p = subprocess.run([BIN, "sketch", "-1", str(d / "z_1.fq.gz"), "-2", str(d / "z_2.fq.gz"), "--fpr", "0", "-d", str(o)],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, SYLPH_HIP_FEED_TRACE="1", SYLPH_HIP_INFLATE_DEVICE=inflate_device))
        assert p.returncode == 0 and marker in p.stderr, p.stderr[-2000:]        # (the trace line of the indexed path / of the device's inflate)
        a = (o / "z_1.fq.gz.paired.sylsp").read_bytes()
        assert n_tab > 1000 and a[:n_tab] == b[:n_tab] and a[n_tab:n_tab + 16] == b[n_tab:n_tab + 16]   # + c, k; the file names differ


def test_device_fastq_route_equals_the_host_feed(data):
    """SYLPH_HIP_FEED_DEVICE=1: every plain-FASTQ sample whose engine is already up sends its TEXT to the device, where the library finds
    the records (csrc/fastq.hip; host/commands.cpp sketch_fastq_on_device); a process's first sample, gzip input and anything that is
    not plain four-line FASTQ go the host way — gzip files are inflated on the host and their TEXT then takes the same route.  Seven
    samples through one engine — a single-member gzip pair first, plain pairs, a CRLF copy, a pair whose mate 2 is longer, a blocked-gzip
    pair, a pair with a damaged record in the middle, a single-end file — must give byte-identical sketches and the same exit code
    either way, and `profile` on raw pairs (gzip ones too) the same rows."""
    d = data["dir"]
    t1, t2 = (d / "s_1.fq").read_bytes(), (d / "s_2.fq").read_bytes()
    for m, t in (("1", t1), ("2", t2)):
        (d / f"da_{m}.fq").write_bytes(t)
        (d / f"db_{m}.fq").write_bytes(t.replace(b"\n", b"\r\n"))
        (d / f"dd_{m}.fq.gz").write_bytes(bgzf_compress(t, block=4000))
        (d / f"df_{m}.fq.gz").write_bytes(gzip.compress(t, 1))                  # an ordinary single-member gzip file
    (d / "dc_1.fq").write_bytes(t1)
    (d / "dc_2.fq").write_bytes(t2 + b"@extra\nACGTACGTAC\n+\nIIIIIIIIII\n\n\n")
    lines = t2.split(b"\n")
    lines[4 * 57 + 2] = b"-"                                                     # record 57 of mate 2 loses its '+' line
    (d / "de_1.fq").write_bytes(t1)
    (d / "de_2.fq").write_bytes(b"\n".join(lines))
    firsts = [d / "df_1.fq.gz"] + [d / f"d{x}_1.fq" for x in "abc"] + [d / "dd_1.fq.gz", d / "de_1.fq"]
    seconds = [d / "df_2.fq.gz"] + [d / f"d{x}_2.fq" for x in "abc"] + [d / "dd_2.fq.gz", d / "de_2.fq"]
    got = {}
    for dev in ("0", "1"):
        o = d / f"devroute_{dev}"
        p = run("sketch", "-t", "1", "-1", *firsts, "-2", *seconds, "-r", d / "da_1.fq", "-d", o, accept_exact=False, check=False,
                env_extra={"SYLPH_HIP_FEED_DEVICE": dev, "SYLPH_HIP_FEED_TRACE": "1"})
        got[dev] = (p.returncode, {f.name: f.read_bytes() for f in sorted(o.iterdir())})
        if dev == "1":
            assert "device route: pushed" in p.stderr, p.stderr[-3000:]
            # round 6: the two gzip pairs (single-member, blocked) send their COMPRESSED bytes; the device inflates (csrc/inflate.hip)
            assert p.stderr.count("device route: gzip inflated on the device") == 2, p.stderr[-3000:]   # (one line per sample: both mates in one call)
        else:
            assert "device route" not in p.stderr
    # ... and with the device's inflate switched off the host inflates and the TEXT travels, as in round 5: the same files
    o = d / "devroute_hostinflate"
    p = run("sketch", "-t", "1", "-1", *firsts, "-2", *seconds, "-r", d / "da_1.fq", "-d", o, accept_exact=False, check=False,
            env_extra={"SYLPH_HIP_FEED_DEVICE": "1", "SYLPH_HIP_INFLATE_DEVICE": "0", "SYLPH_HIP_FEED_TRACE": "1"})
    assert p.returncode == 0 and "gzip inflated on the device" not in p.stderr and "device route: pushed" in p.stderr
    assert {f.name: f.read_bytes() for f in sorted(o.iterdir())} == got["1"][1]
    assert got["0"][0] == got["1"][0] == 0
    assert sorted(got["0"][1]) == sorted(got["1"][1]) and len(got["0"][1]) >= 5
    for name in got["0"][1]:
        assert got["0"][1][name] == got["1"][1][name], name
    gen = [data["genomes"][n][0] for n in ("EC590", "K12", "O157", "rand")]
    rows = {}
    for dev in ("0", "1"):
        p = run("profile", *gen, "-t", "1", "-1", d / "df_1.fq.gz", d / "da_1.fq", d / "db_1.fq", d / "dc_1.fq", "-2", d / "df_2.fq.gz", d / "da_2.fq", d / "db_2.fq", d / "dc_2.fq",
                accept_exact=False, env_extra={"SYLPH_HIP_FEED_DEVICE": dev})
        rows[dev] = p.stdout
    assert rows["0"] == rows["1"] and rows["0"].count("\n") >= 4


def test_kernel_variants_behind_environment_knobs_give_the_same_sketches(data):
    """Round 6's A/B knobs select other kernels for the same work: the read kernel's 512-lane variant for ragged input
    (SYLPH_HIP_READS_RAGGED_TPB=512, csrc/reads.hip), round 5's two-level filter pass (SYLPH_HIP_A10_LEVELS=2) and one / four workgroups
    per range of the one-level pass (SYLPH_HIP_A10_RANGE_SPLIT, csrc/a10.hip).  Ragged pairs (trimmed to 35..151 bases, some with N),
    through the host feed and through the device route, with sylph's default filter dedup: byte-identical .sylsp files."""
    d = data["dir"]
    rng = np.random.default_rng(77)
    t = [(d / "s_1.fq").read_bytes().split(b"\n"), (d / "s_2.fq").read_bytes().split(b"\n")]
    n = min(len(t[0]), len(t[1])) // 4
    for m in (0, 1):
        out = []
        for r in range(n):
            seq = bytearray(t[m][4 * r + 1])
            keep = int(rng.integers(35, max(36, len(seq) + 1)))
            seq = seq[:keep]
            if r % 17 == 0 and keep > 40:
                seq[int(rng.integers(0, keep))] = ord("N")
            out += [t[m][4 * r], bytes(seq), b"+", b"I" * len(seq)]
        (d / f"rg_{m + 1}.fq").write_bytes(b"\n".join(out) + b"\n")
    got = {}
    for name, env in (("default", {}), ("tpb512", {"SYLPH_HIP_READS_RAGGED_TPB": "512"}), ("two_levels", {"SYLPH_HIP_A10_LEVELS": "2"}),
                      ("split1", {"SYLPH_HIP_A10_RANGE_SPLIT": "1"}), ("split4_tpb512", {"SYLPH_HIP_A10_RANGE_SPLIT": "4", "SYLPH_HIP_READS_RAGGED_TPB": "512"})):
        o = d / f"variants_{name}"
        # two samples per command: the first goes the host feed, the second the device route
        run("sketch", "-t", "1", "-c", "20", "-1", d / "rg_1.fq", d / "rg_1.fq", "-2", d / "rg_2.fq", d / "rg_2.fq", "-S", "a", "b", "-d", o, accept_exact=False, env_extra=env)
        got[name] = {f.name: f.read_bytes() for f in sorted(o.iterdir())}
        assert len(got[name]) == 2
    for name in got:
        assert got[name] == got["default"], name


def test_damaged_gzip_goes_the_host_way(data):
    """A truncated .fastq.gz: the device's inflate declines it (SYLPH_ERR_FORMAT: the chain of deflate blocks ends before a final block)
    and the host reader takes the file, with the reference's behaviour for it — the same exit code, messages and sketch (if any) as with
    the device's inflate switched off."""
    d = data["dir"]
    t1 = (d / "s_1.fq").read_bytes()
    gz = gzip.compress(t1, 6)
    (d / "trunc.fq.gz").write_bytes(gz[: len(gz) * 2 // 3])
    (d / "flipped.fq.gz").write_bytes(gz[:5000] + bytes([gz[5000] ^ 0x40]) + gz[5001:])
    res = {}
    for inf in ("1", "0"):
        o = d / f"damaged_{inf}"
        p = run("sketch", "-t", "1", "-r", d / "da_1.fq", d / "trunc.fq.gz", d / "flipped.fq.gz", d / "single.fastq.gz", "-d", o, check=False,
                env_extra={"SYLPH_HIP_INFLATE_DEVICE": inf, "SYLPH_HIP_FEED_TRACE": "1"})
        res[inf] = (p.returncode, {f.name: f.read_bytes() for f in sorted(o.iterdir())}, sorted(ln for ln in p.stderr.split("\n") if "WARN" in ln or "ERROR" in ln))
        if inf == "1":
            assert p.stderr.count("device inflate declined") == 2 and p.stderr.count("gzip inflated on the device") == 1, p.stderr[-3000:]
    assert res["1"] == res["0"] and "single.fastq.gz.sylsp" in res["1"][1]


def test_reads_from_a_named_pipe(data):
    """A sample that is not a regular file (`mkfifo`, process substitution) can be read once, front to back: nothing may peek at it
    (the gzip test of the device-side route stats before it reads), and it goes through the sequential reader like in the reference."""
    import threading
    d = data["dir"]
    fifo = d / "pipe_1.fq"
    if fifo.exists():
        fifo.unlink()
    os.mkfifo(fifo)
    text = (d / "s_1.fq").read_bytes()

    def feed():
        with open(fifo, "wb") as f:
            f.write(text)
    t = threading.Thread(target=feed, daemon=True)
    t.start()
    o = d / "pipe_out"
    p = run("sketch", "-r", fifo, "-d", o, check=False)
    t.join(timeout=60)
    assert p.returncode == 0, p.stderr[-2000:]
    run("sketch", "-r", d / "s_1.fq", "-d", o)
    a, b = (o / "pipe_1.fq.sylsp").read_bytes(), (o / "s_1.fq.sylsp").read_bytes()
    n_tab = 8 + 12 * int.from_bytes(b[:8], "little")
    assert n_tab > 1000 and a[:n_tab] == b[:n_tab]


def test_database_does_not_depend_on_threads(data):
    """Genome files are parsed and inflated on the -t threads and appended in file order: the .syldb must be the same bytes for
    every -t, with plain and gzip files, a file that is not FASTA in the middle of the list (warned about, skipped), `-i`."""
    d = data["dir"]
    g = data["genomes"]
    files = [g[n][0] for n in ("EC590", "K12", "O157", "rand")] * 3
    bad = d / "not_a_genome.fa"
    bad.write_text("this is not fasta\n")
    files.insert(5, str(bad))
    for extra in ((), ("-i",)):
        outs = []
        for t in ("1", "2", "16"):
            o = d / f"dbt_{t}{'_i' if extra else ''}"
            p = run("sketch", *files, "-o", o, "-t", t, *extra)
            assert "not_a_genome.fa is not a valid fasta/fastq file" in p.stderr
            outs.append((d / f"{o}.syldb").read_bytes())
        assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 10000


def test_database_views_equal_copied_database(data):
    """Round 4: `profile` / `query` read a .syldb as VIEWS of the mapped file and gather the k-mers straight into the library's
    page-locked upload chunks (sylph_upload_*), instead of copying every genome into a vector and the vectors into one flat array
    (SYLPH_HIP_DB_COPY_LOAD=1, rounds 1-3).  Same rows, bit for bit: two databases + a raw genome in one command (views and
    freshly sketched vectors mixed), profile (tracked k-mers, reassignment) and query, several parse-thread counts."""
    d = data["dir"]
    g = data["genomes"]
    out = d / "out_views"
    run("sketch", g["EC590"][0], g["K12"][0], "-o", out / "db1")
    run("sketch", g["O157"][0], "-o", out / "db2")
    run("sketch", "-1", d / "s_1.fq", "-2", d / "s_2.fq", "-r", d / "single.fastq.gz", "-d", out)
    args = [out / "db1.syldb", g["rand"][0], out / "db2.syldb", out / "s_1.fq.paired.sylsp", out / "single.fastq.gz.sylsp"]
    ref = {}
    for cmd in ("profile", "query"):
        env = dict(os.environ, SYLPH_HIP_DB_COPY_LOAD="1", SYLPH_HIP_EXACT_DEDUP="1")
        p = subprocess.run([BIN, cmd] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        ref[cmd] = p.stdout
        assert len(p.stdout.strip().split("\n")) >= 3
    for threads in ("1", "3", "64"):
        for cmd in ("profile", "query"):
            env = dict(os.environ, SYLPH_HIP_PARSE_THREADS=threads, SYLPH_HIP_EXACT_DEDUP="1", SYLPH_HIP_DEBUG="1")
            p = subprocess.run([BIN, cmd] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
            assert p.returncode == 0, p.stderr[-2000:]
            assert p.stdout == ref[cmd], (cmd, threads)
            assert "uploaded and indexed in" in p.stderr


def test_profile_over_several_gpus_equals_one_gpu(data):
    """Round 5: `--gpus N|all` — the database replicated (index copied device to device), sample threads dealt to the GPUs in turn, ONE
    router pipeline (sylph_pipeline_create_multi) handing the samples back in input order: the TSV must be the one-GPU TSV, byte for
    byte, for query and profile.  On the one-GPU box SYLPH_HIP_SHARE_GPUS=1 lets three replicas share device 0."""
    d = data["dir"]
    g = data["genomes"]
    order = ["EC590", "K12", "O157", "rand"]
    gen = [g[n][0] for n in order]
    raw = ["-1", d / "s_1.fq", d / "s_1.fq", d / "s_1.fq", "-2", d / "s_2.fq", d / "s_2.fq", d / "s_2.fq", "-r", d / "single.fastq.gz", d / "single.fastq.gz"]
    for cmd in ("profile", "query"):
        one = run(cmd, *gen, *raw, "-t", "3")
        many = run(cmd, *gen, *raw, "-t", "3", "--gpus", "3", env_extra={"SYLPH_HIP_SHARE_GPUS": "1"})
        assert "replicated on 3 GPUs" in many.stderr
        assert one.stdout == many.stdout and len(one.stdout.strip().split("\n")) >= 6
        assert run(cmd, *gen, *raw, "--gpus", "all").stdout == one.stdout


def test_sketch_over_several_gpus_equals_one_gpu(data):
    """Round 6: `sylph-hip sketch --gpus N|all` — the samples' workers dealt to the GPUs (worker w on device w mod N, each with its own
    context, page-locked batch and uploader; nothing exchanged: what the reference's rayon pool does with the machine's cores,
    sketch.rs:313, :371).  The sketches must be the one-GPU sketches, byte for byte.  On the one-GPU box SYLPH_HIP_FAKE_GPUS=3 deals the
    workers as for three GPUs and runs them all on device 0."""
    d = data["dir"]
    for i in range(5):
        for m in ("1", "2"):
            if not (d / f"mg{i}_{m}.fq").exists():
                (d / f"mg{i}_{m}.fq").write_bytes((d / f"s_{m}.fq").read_bytes())
    firsts = [d / f"mg{i}_1.fq" for i in range(5)]
    seconds = [d / f"mg{i}_2.fq" for i in range(5)]
    one = d / "mg_one"
    run("sketch", "-1", *firsts, "-2", *seconds, "-r", d / "single.fastq.gz", "-t", "1", "-d", one)
    many = d / "mg_many"
    p = run("sketch", "-1", *firsts, "-2", *seconds, "-r", d / "single.fastq.gz", "-t", "2", "--gpus", "3", "-d", many, env_extra={"SYLPH_HIP_FAKE_GPUS": "3"})
    assert "sketch worker 1 runs on GPU 1" in p.stderr and "sketch worker 2 runs on GPU 2" in p.stderr, p.stderr[-2000:]
    a = {f.name: f.read_bytes() for f in sorted(one.iterdir())}
    b = {f.name: f.read_bytes() for f in sorted(many.iterdir())}
    assert a == b and len(a) == 6
    allg = d / "mg_all"
    run("sketch", "-1", *firsts, "-2", *seconds, "-r", d / "single.fastq.gz", "--gpus", "all", "-d", allg)
    assert {f.name: f.read_bytes() for f in sorted(allg.iterdir())} == a
