"""ctypes binding of the CPU oracle (oracle/sylph_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under sylph_amd/ may import this module.  Parity status: unpinned against a reference binary (see the
header of sylph_oracle.cpp and DESIGN.md §Oracle).
"""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_SCALAR, MODE_AVX2_COMPAT, MODE_AVX2_FAST = 0, 1, 2

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class OrcStats(C.Structure):
    _fields_ = [("naive_ani", C.c_double), ("final_est_ani", C.c_double), ("final_est_cov", C.c_double),
                ("mean_cov", C.c_double), ("median_cov", C.c_double), ("lambda_", C.c_double),
                ("max_cov", C.c_double), ("full_mean_cov", C.c_double), ("lambda_status", C.c_int32),
                ("passed", C.c_int32), ("contain_count", C.c_uint64), ("n_kmers", C.c_uint64),
                ("n_full", C.c_uint64)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "sylph_oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    L.orc_mm_hash64.restype = C.c_uint64
    L.orc_mm_hash64.argtypes = [C.c_uint64]
    L.orc_threshold.restype = C.c_uint64
    L.orc_threshold.argtypes = [C.c_uint64]
    L.orc_byte_to_seq.restype = C.c_uint8
    L.orc_byte_to_seq.argtypes = [C.c_uint8]
    L.orc_extract_markers.restype = C.c_int64
    L.orc_extract_markers.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
    L.orc_extract_markers_positions.restype = C.c_int64
    L.orc_extract_markers_positions.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_uint64]
    L.orc_pair_kmer_single.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.orc_pair_kmer.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    L.orc_cuckoo_walk.restype = C.c_uint64
    L.orc_cuckoo_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_uint64, C.c_void_p]
    L.orc_sketch_reads_cuckoo.restype = C.c_void_p
    L.orc_sketch_reads_cuckoo.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_uint64]
    L.orc_sketch_reads.restype = C.c_void_p
    L.orc_sketch_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]
    for f in ("orc_sketch_size", "orc_sketch_dup_removed"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [C.c_void_p]
    L.orc_sketch_mean_read_length.restype = C.c_double
    L.orc_sketch_mean_read_length.argtypes = [C.c_void_p]
    L.orc_sketch_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_sketch_free.argtypes = [C.c_void_p]
    L.orc_sketch_genome.restype = C.c_void_p
    L.orc_sketch_genome.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_int]
    for f in ("orc_genome_n_kmers", "orc_genome_n_tracked", "orc_genome_gn_size", "orc_genome_n_raw_seeds",
              "orc_genome_n_dup_kmers"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [C.c_void_p]
    L.orc_genome_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_genome_free.argtypes = [C.c_void_p]
    L.orc_sample_load.restype = C.c_void_p
    L.orc_sample_load.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.orc_sample_free.argtypes = [C.c_void_p]
    L.orc_contain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_stats.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_int,
                            C.POINTER(OrcStats)]
    L.orc_poisson_cdf.restype = C.c_double
    L.orc_poisson_cdf.argtypes = [C.c_double, C.c_uint64]
    L.orc_ratio_lambda.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.POINTER(C.c_double)]
    L.orc_has_avx2.restype = C.c_int
    _LIB = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _bases(b):
    a = np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else np.ascontiguousarray(b, dtype=np.uint8)
    return a


def mm_hash64(x):
    return int(lib().orc_mm_hash64(int(x) & 0xFFFFFFFFFFFFFFFF))


def threshold(c):
    return int(lib().orc_threshold(c))


def extract_markers(seq, c=200, k=31, mode=MODE_AVX2_COMPAT):
    """sketch.rs:53 extract_markers -> np.uint64 array in the reference's emission order."""
    a = _bases(seq)
    cap = max(16, len(a))
    out = np.empty(cap, dtype=np.uint64)
    n = lib().orc_extract_markers(_ptr(a), len(a), c, k, mode, _ptr(out), cap)
    if n < 0:
        raise ValueError("reference would panic (k must be 21 or 31 on the AVX2 path)")
    return out[:n].copy()


def extract_markers_positions(seq, c=200, k=31, mode=MODE_AVX2_COMPAT):
    """sketch.rs:71 extract_markers_positions -> (pos, hash) arrays; pos = index of the k-mer's last base."""
    a = _bases(seq)
    cap = max(16, len(a))
    pos = np.empty(cap, dtype=np.uint64)
    h = np.empty(cap, dtype=np.uint64)
    n = lib().orc_extract_markers_positions(_ptr(a), len(a), c, k, mode, _ptr(pos), _ptr(h), cap)
    if n < 0:
        raise ValueError("reference would panic")
    return pos[:n].copy(), h[:n].copy()


def pair_kmer_single(seq):
    a = _bases(seq)
    out = np.zeros(4, dtype=np.uint32)
    some = lib().orc_pair_kmer_single(_ptr(a), len(a), _ptr(out))
    return (tuple(int(x) for x in out) if some else None)


def pair_kmer(s1, s2):
    a, b = _bases(s1), _bases(s2)
    out = np.zeros(4, dtype=np.uint32)
    some = lib().orc_pair_kmer(_ptr(a), len(a), _ptr(b), len(b), _ptr(out))
    return (tuple(int(x) for x in out) if some else None)


def concat(records):
    """list of bytes -> (uint8 bases, uint64 offsets[n+1])."""
    off = np.zeros(len(records) + 1, dtype=np.uint64)
    if records:
        off[1:] = np.cumsum([len(r) for r in records], dtype=np.uint64)
    bases = np.frombuffer(b"".join(records), dtype=np.uint8).copy() if records else np.zeros(0, dtype=np.uint8)
    return bases, off


def sketch_reads(bases, off, c=200, k=31, mode=MODE_AVX2_COMPAT, paired=False, no_dedup=False):
    """sketch_sequences_needle (sketch.rs:897) / sketch_pair_sequences with --fpr 0 (sketch.rs:771).
    Paired input is interleaved (mate1, mate2, mate1, ...).  Returns dict(kmers, counts, dup_removed, mean_read_length);
    kmers ascending."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    if len(bases) == 0:
        bases = np.zeros(1, dtype=np.uint8)
    h = lib().orc_sketch_reads(_ptr(bases), _ptr(off), n, c, k, mode, int(paired), int(no_dedup))
    if not h:
        raise ValueError("oracle sketch failed (k unsupported on AVX2 path)")
    try:
        m = lib().orc_sketch_size(h)
        kmers = np.empty(m, dtype=np.uint64)
        counts = np.empty(m, dtype=np.uint32)
        lib().orc_sketch_copy(h, _ptr(kmers), _ptr(counts))
        return dict(kmers=kmers, counts=counts, dup_removed=int(lib().orc_sketch_dup_removed(h)),
                    mean_read_length=float(lib().orc_sketch_mean_read_length(h)))
    finally:
        lib().orc_sketch_free(h)


def sketch_files(first, second=None, c=200, k=31, mode=None, fpr=0.0, threads=1):
    """The reference's `sylph sketch` from FILES as the CPU runs it (sketch.rs:313 / :371: one worker per sample; needletail + flate2 on
    that thread): every sample's file(s) read through zlib, cut into records and sketched on ONE thread, `threads` samples at a time.
    -> dict(seconds=[per sample], table_sizes=[...], n_bases=[...], wall_seconds)."""
    import ctypes as C
    import time
    n = len(first)
    a1 = (C.c_char_p * n)(*[str(f).encode() for f in first])
    a2 = (C.c_char_p * n)(*[str(f).encode() for f in second]) if second is not None else None
    sec = (C.c_double * n)()
    tab = (C.c_uint64 * n)()
    nb = (C.c_uint64 * n)()
    L = lib()
    L.orc_sketch_files.restype = C.c_int
    L.orc_sketch_files.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_int,
                                   C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    t = time.perf_counter()
    rc = L.orc_sketch_files(a1, a2, n, c, k, MODE_AVX2_FAST if mode is None else mode, float(fpr), int(threads), sec, tab, nb)
    wall = time.perf_counter() - t
    if rc:
        raise ValueError(f"oracle could not read sample {rc - 1}")
    return dict(seconds=list(sec), table_sizes=list(tab), n_bases=list(nb), wall_seconds=wall)


def sketch_reads_cuckoo_model(bases, off, c=200, k=31, mode=MODE_AVX2_COMPAT, fpr=1e-4, initial_capacity=10_000_000):
    """sketch_pair_sequences with the DEFAULT dedup (sketch.rs:733-769 over a scalable cuckoo filter; --fpr 1e-4, capacity 10^7
    at :796-804), the filter restated from the paper — a model of the third-party crate, not its bits (see sylph_oracle.cpp)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    h = lib().orc_sketch_reads_cuckoo(_ptr(bases), _ptr(off), len(off) - 1, c, k, mode, float(fpr), int(initial_capacity))
    if not h:
        raise ValueError("oracle sketch failed")
    try:
        m = lib().orc_sketch_size(h)
        kmers = np.empty(m, dtype=np.uint64)
        counts = np.empty(m, dtype=np.uint32)
        lib().orc_sketch_copy(h, _ptr(kmers), _ptr(counts))
        return dict(kmers=kmers, counts=counts, dup_removed=int(lib().orc_sketch_dup_removed(h)),
                    mean_read_length=float(lib().orc_sketch_mean_read_length(h)))
    finally:
        lib().orc_sketch_free(h)


def cuckoo_walk(km, marker, fpr=1e-4, initial_capacity=10_000_000):
    """The model's filter walked item by item (test, insert when absent: sketch.rs:747-760) -> (contained[n] bool, number of filters)."""
    km = np.ascontiguousarray(km, dtype=np.uint64)
    marker = np.ascontiguousarray(marker, dtype=np.uint64)
    out = np.zeros(len(km), dtype=np.uint8)
    nf = lib().orc_cuckoo_walk(_ptr(km), _ptr(marker), len(km), float(fpr), int(initial_capacity), _ptr(out))
    return out.astype(bool), int(nf)


def sketch_genome(bases, off, c=200, k=31, mode=MODE_AVX2_COMPAT, min_spacing=30, pseudotax=True):
    """sketch_genome (sketch.rs:550) -> dict(genome_kmers, tracked, gn_size, n_raw_seeds, n_dup_kmers)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    if len(bases) == 0:
        bases = np.zeros(1, dtype=np.uint8)
    h = lib().orc_sketch_genome(_ptr(bases), _ptr(off), len(off) - 1, c, k, mode, min_spacing, int(pseudotax))
    if not h:
        raise ValueError("oracle genome sketch failed")
    try:
        gk = np.empty(lib().orc_genome_n_kmers(h), dtype=np.uint64)
        tr = np.empty(lib().orc_genome_n_tracked(h), dtype=np.uint64)
        lib().orc_genome_copy(h, _ptr(gk), _ptr(tr))
        return dict(genome_kmers=gk, tracked=tr, gn_size=int(lib().orc_genome_gn_size(h)),
                    n_raw_seeds=int(lib().orc_genome_n_raw_seeds(h)), n_dup_kmers=int(lib().orc_genome_n_dup_kmers(h)))
    finally:
        lib().orc_genome_free(h)


def contain(sample_kmers, sample_counts, db_kmers, genome_off, min_number_kmers=50.0, winner_of=None,
            genome_ids=None, n_threads=1):
    """Probe half of get_stats (contain.rs:601-656) for every genome.
    Returns (contain_count[G], covs list-of-arrays in genome k-mer order, kmers_lost[G])."""
    sk = np.ascontiguousarray(sample_kmers, dtype=np.uint64)
    sc = np.ascontiguousarray(sample_counts, dtype=np.uint32)
    db = np.ascontiguousarray(db_kmers, dtype=np.uint64)
    go = np.ascontiguousarray(genome_off, dtype=np.uint64)
    G = len(go) - 1
    s = lib().orc_sample_load(_ptr(sk), _ptr(sc), len(sk))
    try:
        cc = np.zeros(G, dtype=np.uint32)
        lost = np.zeros(G, dtype=np.uint32)
        cov = np.zeros(max(1, len(db)), dtype=np.uint32)
        w = None if winner_of is None else np.ascontiguousarray(winner_of, dtype=np.uint32)
        gi = None if genome_ids is None else np.ascontiguousarray(genome_ids, dtype=np.uint32)
        lib().orc_contain(s, _ptr(db), _ptr(go), G, float(min_number_kmers), _ptr(w), _ptr(gi), _ptr(cc), _ptr(cov),
                          _ptr(lost), n_threads)
    finally:
        lib().orc_sample_free(s)
    covs = [cov[int(go[g]):int(go[g]) + int(cc[g])].copy() for g in range(G)]
    return cc, covs, lost


class LoadedSample:
    """A sample table loaded once into the Fx-hashed map (what contain.rs:559 does when it deserialises a .sylsp);
    probe() times only the genome loop (contain.rs:284-291), for bench.py's cpu_baseline leg."""

    def __init__(self, kmers, counts):
        self.k = np.ascontiguousarray(kmers, dtype=np.uint64)
        self.c = np.ascontiguousarray(counts, dtype=np.uint32)
        self.h = lib().orc_sample_load(_ptr(self.k), _ptr(self.c), len(self.k))

    def probe(self, db_kmers, genome_off, min_number_kmers=50.0, n_threads=1):
        db = np.ascontiguousarray(db_kmers, dtype=np.uint64)
        go = np.ascontiguousarray(genome_off, dtype=np.uint64)
        G = len(go) - 1
        cc = np.zeros(G, dtype=np.uint32)
        cov = np.zeros(max(1, len(db)), dtype=np.uint32)
        import time
        t = time.perf_counter()
        lib().orc_contain(self.h, _ptr(db), _ptr(go), G, float(min_number_kmers), None, None, _ptr(cc), _ptr(cov), None,
                          n_threads)
        return cc, cov, time.perf_counter() - t

    def close(self):
        if self.h:
            lib().orc_sample_free(self.h)
            self.h = None


def stats(covs, n_genome_kmers, k=31, min_count_correct=3.0, min_ani=0.0, no_adj=False, mean_coverage=False):
    """Statistics half of get_stats (contain.rs:657-813), default ratio estimator."""
    cv = np.ascontiguousarray(covs, dtype=np.uint32)
    out = OrcStats()
    lib().orc_stats(_ptr(cv), len(cv), n_genome_kmers, k, min_count_correct, min_ani, int(no_adj), int(mean_coverage),
                    C.byref(out))
    return out


def poisson_cdf(lam, x):
    return float(lib().orc_poisson_cdf(lam, x))


def ratio_lambda(full_covs, min_count_correct=3.0):
    a = np.ascontiguousarray(full_covs, dtype=np.uint32)
    out = C.c_double(0)
    ok = lib().orc_ratio_lambda(_ptr(a), len(a), min_count_correct, C.byref(out))
    return out.value if ok else None


# ---- minimal FASTA/FASTQ reader with needletail 0.5.1 record semantics (seq() = sequence with newlines
# ---- stripped, id() = whole header line after '>'/'@').  Call sites: sketch.rs:488,557,780-781,906.
def read_fastx(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        data = f.read()
    recs = []
    if not data:
        return recs
    lines = data.split(b"\n")
    if data[:1] == b">":
        name, chunks = None, []
        for ln in lines:
            ln = ln.rstrip(b"\r")
            if ln.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(chunks)))
                name, chunks = ln[1:], []
            elif name is not None:
                chunks.append(ln)
        if name is not None:
            recs.append((name, b"".join(chunks)))
    elif data[:1] == b"@":
        i = 0
        while i + 3 < len(lines) + 1 and i < len(lines) and lines[i].startswith(b"@"):
            recs.append((lines[i][1:].rstrip(b"\r"), lines[i + 1].rstrip(b"\r")))
            i += 4
    return recs
