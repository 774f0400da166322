"""pyref.py — SECOND, independent restatement of the reference's hot path, in pure Python integers.

TEST INFRASTRUCTURE ONLY (same rule as oracle.py: only tests/ may import it).  It exists to cross-check the C++
oracle (oracle/sylph_oracle.cpp): it was written directly from the Rust sources under /root/reference/src — not from
the C++ — and shares no code with it: Python ints / dicts / sets instead of 64-bit registers and open-addressing tables,
the 4-lane AVX2 code emulated lane by lane in its real emission order, the Poisson tail from scipy/mpmath instead of a
hand-written incomplete gamma.  tests/test_pyref.py diffs the two on random inputs (hypothesis) and pins this file to
the survey's known answers (tests/golden/survey_kat.json).
Parity status is unchanged by this file: "unpinned against a reference binary" (no Rust toolchain here or on the GPU
box: `which cargo rustc` finds nothing on either, checked in round 2) — two restatements and the survey's numbers agree.

Every function cites the Rust it follows (file:line under /root/reference/src).
"""
import math

M64 = (1 << 64) - 1


# types.rs:50-59 — BYTE_TO_SEQ: raw 0..3 map to themselves, A/a=0, C/c=1, G/g=2, T/t/U/u=3, everything else 0
def _byte_to_seq_table():
    t = [0] * 256
    t[0], t[1], t[2], t[3] = 0, 1, 2, 3
    for ch, v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):
        t[ord(ch)] = v
        t[ord(ch.lower())] = v
    return t


BYTE_TO_SEQ = _byte_to_seq_table()


# seeding.rs:4-15
def mm_hash64(kmer):
    key = kmer & M64
    key = (~((key + (key << 21)) & M64)) & M64
    key = key ^ (key >> 24)
    key = (key + (key << 3) + (key << 8)) & M64
    key = key ^ (key >> 14)
    key = (key + (key << 2) + (key << 4)) & M64
    key = key ^ (key >> 28)
    key = (key + (key << 31)) & M64
    return key


def threshold(c):
    return M64 // c          # seeding.rs:108 `u64::MAX / (c as u64)`


# seeding.rs:86-146 (fmh_seeds) and :148-209 (fmh_seeds_positions): -> [(i, hash)], i = index of the k-mer's last base
def fmh_seeds_positions(s, c, k):
    out = []
    if len(s) < k:
        return out
    f = r = 0
    rshift = 2 * (k - 1)
    mask = M64 >> (64 - 2 * k)
    rev_mask = (~(3 << (2 * k - 2))) & M64
    thr = threshold(c)
    for i in range(k - 1):
        nf = BYTE_TO_SEQ[s[i]]
        f = ((f << 2) | nf) & M64
        r = (r >> 2) | ((3 - nf) << rshift)
    for i in range(k - 1, len(s)):
        nf = BYTE_TO_SEQ[s[i]]
        f = ((f << 2) | nf) & mask
        r = ((r >> 2) & rev_mask) | ((3 - nf) << rshift)
        canon = f if f < r else r
        h = mm_hash64(canon)
        if h < thr:
            out.append((i, h))
    return out


def fmh_seeds(s, c, k):
    return [h for _, h in fmh_seeds_positions(s, c, k)]


# avx2_seeding.rs:33-148 (extract_markers_avx2, min_len = k+1) and :150-266 (positions variant, min_len = 2k):
# four lanes walk four quarters of the string in lock step; pushes happen lane 1..4 per step.  -> [(pos, hash)]
def _avx2_lanes(s, c, k, min_len):
    out = []
    n = len(s)
    if n < k:
        return out
    ln = (n - k + 1) // 4
    if n < min_len:
        return out
    if 2 * (k - 1) not in (40, 60):
        raise ValueError("panic!() at avx2_seeding.rs:46-52: k must be 21 or 31")
    starts = [0, ln, 2 * ln, 3 * ln]
    f = [0, 0, 0, 0]
    r = [0, 0, 0, 0]
    rshift = 2 * (k - 1)
    mask = M64 >> (64 - 2 * k)
    rev_mask = (~(3 << (2 * k - 2))) & M64
    thr = threshold(c)
    for i in range(k - 1):
        for l in range(4):
            nf = BYTE_TO_SEQ[s[starts[l] + i]]
            f[l] = ((f[l] << 2) | nf) & M64
            r[l] = (r[l] >> 2) | ((3 - nf) << rshift)
    for i in range(k - 1, ln + k - 1):
        for l in range(4):
            nf = BYTE_TO_SEQ[s[starts[l] + i]]
            f[l] = ((f[l] << 2) | nf) & mask
            r[l] = ((r[l] >> 2) & rev_mask) | ((3 - nf) << rshift)
            # _mm256_cmpgt_epi64(r, f) is a signed compare; both values are < 2^62, so it is the unsigned one
            canon = f[l] if r[l] > f[l] else r[l]
            h = mm_hash64(canon)
            if h < thr:
                out.append((l * ln + i, h))
    return out


def extract_markers(s, c, k, avx2=True):
    """sketch.rs:53-69 — hashes in the reference's emission order."""
    if avx2:
        return [h for _, h in _avx2_lanes(s, c, k, k + 1)]
    return fmh_seeds(s, c, k)


def extract_markers_positions(s, c, k, avx2=True):
    """sketch.rs:71-93 — (pos, hash) in emission order."""
    if avx2:
        return _avx2_lanes(s, c, k, 2 * k)
    return fmh_seeds_positions(s, c, k)


# sketch.rs:625-656.  Marker = u32 => k = 16; -> ((f, r), (g, t)) or None
def pair_kmer_single(s):
    k = 16
    if len(s) < 4 * k + 2:
        return None
    f = g = r = t = 0
    half = len(s) // 2
    for i in range(k):
        f = ((f << 2) | BYTE_TO_SEQ[s[2 * i]]) & 0xFFFFFFFF
        r = ((r << 2) | BYTE_TO_SEQ[s[2 * i + half]]) & 0xFFFFFFFF
        g = ((g << 2) | BYTE_TO_SEQ[s[1 + 2 * i]]) & 0xFFFFFFFF
        t = ((t << 2) | BYTE_TO_SEQ[s[1 + 2 * i + half]]) & 0xFFFFFFFF
    return ((f, r), (g, t))


# sketch.rs:659-688
def pair_kmer(s1, s2):
    k = 16
    if len(s1) < 2 * k + 1 or len(s2) < 2 * k + 1:
        return None
    f = g = r = t = 0
    for i in range(k):
        f = ((f << 2) | BYTE_TO_SEQ[s1[2 * i]]) & 0xFFFFFFFF
        r = ((r << 2) | BYTE_TO_SEQ[s2[2 * i]]) & 0xFFFFFFFF
        g = ((g << 2) | BYTE_TO_SEQ[s1[1 + 2 * i]]) & 0xFFFFFFFF
        t = ((t << 2) | BYTE_TO_SEQ[s2[1 + 2 * i]]) & 0xFFFFFFFF
    return ((f, r), (g, t))


class _Dedup:
    """dup_removal_lsh_full_exact, sketch.rs:690-731, with the FxHashMap / FxHashSet as dict / set."""

    def __init__(self):
        self.counts = {}
        self.pairs = set()
        self.removed = 0

    def add(self, km, kmer_pair, no_dedup, cutoff):
        c = self.counts.setdefault(km, 0)
        c_threshold = cutoff if cutoff is not None else 0xFFFFFFFF
        if not no_dedup and c < c_threshold:
            if kmer_pair is not None:
                ret = False
                for half in kmer_pair:
                    if (km, half) in self.pairs:
                        if c > 0:
                            ret = True
                    else:
                        self.pairs.add((km, half))
                if ret:
                    self.removed += 1
                    return
        self.counts[km] = c + 1


# ---- a10: the DEFAULT pair dedup, dup_removal_lsh_full (sketch.rs:733-769) over a scalable cuckoo filter -----------------------
# The crate (scalable_cuckoo_filter 0.2.4) is not in the reference tree.  What follows is the published structure (Fan, Andersen,
# Kaminsky, Mitzenmacher: "Cuckoo Filter: Practically Better Than Bloom", 2014) with the crate's documented defaults — four
# entries per bucket, fingerprints of ceil(log2(1/fpr) + 3) bits, a further filter of twice the capacity and 0.9x the rate when
# one is full — and the same CHOICE of hash bits as the C++ model (FxHasher of the (u64, [u32; 2]) tuple, mixed; fingerprint from
# the high word, bucket from the low bits, the partner bucket i ^ (fx(f) >> 11)).  Written from that description, not from the
# C++: Python lists, and a DIFFERENT eviction policy (the victim is taken round-robin, the C++ model draws it from an LCG) —
# which must not matter: an eviction moves a fingerprint between ITS two buckets only, so what `contains` answers depends on what
# was inserted, not on where it ended up.  tests/test_pyref.py holds the two models against each other.
_FX_K = 0x517cc1b727220a95
_GOLD = 0x9E3779B97F4A7C15


def _fx_add(h, w):
    return ((((h << 5) | (h >> 59)) & M64) ^ w) * _FX_K & M64


def _item_hash(km, half):
    # FxHasher: the u64, then the two u32 of the marker pair, one write each
    return _fx_add(_fx_add(_fx_add(0, km), half[0]), half[1]) * _GOLD & M64


class _CuckooFilter:
    def __init__(self, capacity, fpr):
        self.capacity = capacity
        self.fp_bits = min(31, int(math.ceil(math.log2(1.0 / fpr) + math.log2(8.0))))
        nb = 1
        while nb * 4 < capacity:
            nb <<= 1
        self.nb = nb
        self.buckets = [[] for _ in range(nb)]
        self.n_items = 0
        self.victim = 0

    def _where(self, h):
        f = (h >> 32) & ((1 << self.fp_bits) - 1) or 1
        i1 = h & (self.nb - 1)
        return f, i1, (i1 ^ (_fx_add(0, f) >> 11)) & (self.nb - 1)

    def contains(self, h):
        f, i1, i2 = self._where(h)
        return f in self.buckets[i1] or f in self.buckets[i2]

    def insert(self, h):
        f, i1, i2 = self._where(h)
        for i in (i1, i2):
            if len(self.buckets[i]) < 4:
                self.buckets[i].append(f)
                self.n_items += 1
                return True
        i = i1
        for _ in range(512):
            self.victim = (self.victim + 1) & 3
            f, self.buckets[i][self.victim] = self.buckets[i][self.victim], f
            i = (i ^ (_fx_add(0, f) >> 11)) & (self.nb - 1)
            if len(self.buckets[i]) < 4:
                self.buckets[i].append(f)
                self.n_items += 1
                return True
        return False


class _ScalableCuckoo:
    def __init__(self, initial_capacity, fpr):
        self.cap0, self.fpr = initial_capacity, fpr
        self.filters = [_CuckooFilter(initial_capacity, fpr)]

    def contains(self, h):
        return any(f.contains(h) for f in self.filters)

    def insert(self, h):
        last = self.filters[-1]
        if last.n_items >= last.capacity or not last.insert(h):
            n = len(self.filters)
            self.filters.append(_CuckooFilter(self.cap0 << n, self.fpr * 0.9 ** n))
            self.filters[-1].insert(h)


class _FilterDedup:
    """dup_removal_lsh_full, sketch.rs:733-769: no cut-off; the set is the filter."""

    def __init__(self, fpr, initial_capacity):
        self.counts = {}
        self.set = _ScalableCuckoo(initial_capacity, fpr)
        self.removed = 0

    def add(self, km, kmer_pair, no_dedup, _cutoff):
        c = self.counts.setdefault(km, 0)                          # :743
        if not no_dedup and kmer_pair is not None:                 # :744-745
            ret = False
            for half in kmer_pair:                                 # :747-760
                h = _item_hash(km, half)
                if self.set.contains(h):
                    if c > 0:
                        ret = True
                else:
                    self.set.insert(h)
            if ret:                                                # :761-764
                self.removed += 1
                return
        self.counts[km] = c + 1                                    # :767


MAX_DEDUP_COUNT = 4   # constants.rs:14


# sketch.rs:897-959
def sketch_sequences_needle(records, c, k, no_dedup=False, avx2=True):
    d = _Dedup()
    mean, counter = 0.0, 0.0
    for seq in records:
        kmer_pair = None if len(seq) > 400 else pair_kmer_single(seq)
        for km in extract_markers(seq, c, k, avx2):
            d.add(km, kmer_pair, no_dedup, MAX_DEDUP_COUNT)
        counter += 1.0
        mean = mean + (float(len(seq)) - mean) / counter
    return dict(kmer_counts=d.counts, dup_removed=d.removed, mean_read_length=mean)


# sketch.rs:771-895: dedup_fpr == 0 the exact set (:829-838), else the filter with that false-positive probability (:796-804,
# :839-848; initial capacity 10^7 in the reference, a parameter here so that small tests can make it grow)
def sketch_pair_sequences(records1, records2, c, k, no_dedup=False, avx2=True, dedup_fpr=0.0, initial_capacity=10_000_000):
    d = _FilterDedup(dedup_fpr, initial_capacity) if dedup_fpr != 0.0 else _Dedup()
    mean, counter = 0.0, 0.0
    for s1, s2 in zip(records1, records2):
        v1 = extract_markers(s1, c, k, avx2)
        v2 = extract_markers(s2, c, k, avx2)
        kmer_pair = pair_kmer(s1, s2)
        counter += 1.0
        mean = mean + (float(len(s1)) - mean) / counter
        for km in v1:
            d.add(km, kmer_pair, no_dedup, None)
        for km in v2:
            if km in v1:                # sketch.rs:852 `temp_vec1.contains(km)`
                continue
            d.add(km, kmer_pair, no_dedup, None)
    return dict(kmer_counts=d.counts, dup_removed=d.removed, mean_read_length=mean)


# sketch.rs:550-622 (whole file = one genome) — contigs: list of sequences
def sketch_genome(contigs, c, k, min_spacing=30, pseudotax=True, avx2=True):
    vec = []
    gn_size = 0
    for ci, seq in enumerate(contigs):
        gn_size += len(seq)
        vec += [(ci, pos, h) for pos, h in extract_markers_positions(seq, c, k, avx2)]
    vec.sort()
    seen, dup = set(), set()
    for _, _, km in vec:
        if km not in seen:
            seen.add(km)
        else:
            dup.add(km)
    kept, tracked = [], []
    last_pos, last_contig = 0, 0
    for contig, pos, km in vec:
        if km not in dup:
            if last_pos == 0 or last_contig != contig or pos - last_pos > min_spacing:
                kept.append(km)
                last_contig, last_pos = contig, pos
            elif pseudotax:
                tracked.append(km)
    return dict(genome_kmers=kept, tracked=tracked, gn_size=gn_size, n_raw_seeds=len(vec), n_dup_kmers=len(dup))


# sketch.rs:481-548 (every record its own genome: no last_contig test, contig number 0)
def sketch_genome_individual(seq, c, k, min_spacing=30, pseudotax=True, avx2=True):
    vec = sorted((0, pos, h) for pos, h in extract_markers_positions(seq, c, k, avx2))
    seen, dup = set(), set()
    for _, _, km in vec:
        if km not in seen:
            seen.add(km)
        else:
            dup.add(km)
    kept, tracked = [], []
    last_pos = 0
    for _, pos, km in vec:
        if km not in dup:
            if last_pos == 0 or pos - last_pos > min_spacing:
                kept.append(km)
                last_pos = pos
            elif pseudotax:
                tracked.append(km)
    return dict(genome_kmers=kept, tracked=tracked, gn_size=len(seq))


# ---- contain.rs:601-656: probe half of get_stats --------------------------------------------------------------
def probe(genome_kmers, kmer_counts, min_number_kmers=50.0, winner=None, me=None):
    """-> None (too few k-mers) or (contain_count, covs in genome order, kmers_lost).  winner: k-mer -> genome key."""
    if float(len(genome_kmers)) < min_number_kmers:
        return None
    contain, covs, lost = 0, [], 0
    for km in genome_kmers:
        if km in kmer_counts:
            if kmer_counts[km] == 0:
                continue
            if winner is not None and winner[km] != me:
                lost += 1
                continue
            contain += 1
            covs.append(kmer_counts[km])
    return contain, covs, lost


# statrs 0.16.1 Poisson::cdf(x) = gamma_ur(floor(x) + 1, lambda) — the regularised UPPER incomplete gamma Q(x+1, lambda)
# (third-party crate, not under /root/reference: restated from its definition; call site contain.rs:664-669)
def poisson_cdf(lam, x):
    from scipy.special import gammaincc
    return float(gammaincc(math.floor(x) + 1.0, lam))


CUTOFF_PVALUE = 0.9999999999      # constants.rs:3
SAMPLE_SIZE_CUTOFF = 25           # constants.rs:4
MEDIAN_ANI_THRESHOLD = 2.0        # constants.rs:5
MAX_MEDIAN_FOR_MEAN_FINAL_EST = 15.0


# inference.rs:207-242
def ratio_lambda(full_covs, min_count_correct):
    num_zero = 0
    count_map = {}
    for x in full_covs:
        if x == 0:
            num_zero += 1
        else:
            count_map[x] = count_map.get(x, 0) + 1
    if len(count_map) == 1:
        return None
    if len(full_covs) - num_zero < SAMPLE_SIZE_CUTOFF:
        return None
    sort_vec = sorted(((cnt, val) for val, cnt in count_map.items()), reverse=True)
    most_ind = sort_vec[0][1]
    if (most_ind + 1) not in count_map:
        return None
    count_p1 = float(count_map[most_ind + 1])
    count = float(count_map[most_ind])
    if count_p1 < min_count_correct or count < min_count_correct:
        return None
    return count_p1 / count * float(most_ind + 1)


# contain.rs:817-847
def ani_from_lambda(lam, k, full_cov):
    if lam is None:
        return None
    contain = sum(1 for x in full_cov if x != 0)
    adj = contain / (1.0 - math.exp(-lam)) / len(full_cov)
    if adj < 0:
        return None
    ani = adj ** (1.0 / k)
    if ani < 0 or math.isnan(ani):
        return None
    return ani


# contain.rs:657-813 without the bootstrap — default estimator (ratio), u32 sums as in the reference
def stats(contain_count, covs, n_genome_kmers, k=31, min_count_correct=3.0, min_ani=0.0, no_adj=False, mean_coverage=False):
    if not covs:
        return None
    naive_ani = (contain_count / n_genome_kmers) ** (1.0 / k)
    covs = sorted(covs)
    median_cov = float(covs[len(covs) // 2])
    max_cov = math.inf            # f64::MAX in the reference: no count reaches either
    if median_cov < 30.0:
        for i in range(len(covs) // 2, len(covs)):
            cov = covs[i]
            if poisson_cdf(median_cov, float(cov)) < CUTOFF_PVALUE:
                max_cov = float(cov)
            else:
                break
    full_covs = [0] * (n_genome_kmers - contain_count) + [x for x in covs if float(x) <= max_cov]
    total = sum(full_covs) & 0xFFFFFFFF          # iter().sum::<u32>() (a release build wraps)
    mean_cov = total / len(full_covs)
    geq1_mean_cov = total / len(covs)
    if median_cov > MEDIAN_ANI_THRESHOLD:
        status, lam = "HIGH", None
    else:
        lam = ratio_lambda(full_covs, min_count_correct)
        status = "LOW" if lam is None else "LAMBDA"
    if status == "LAMBDA":
        final_est_cov = lam
    elif median_cov < MAX_MEDIAN_FOR_MEAN_FINAL_EST:
        final_est_cov = geq1_mean_cov
    else:
        final_est_cov = geq1_mean_cov if mean_coverage else median_cov
    opt_lambda = final_est_cov if status == "LAMBDA" else None
    opt_est_ani = ani_from_lambda(opt_lambda, float(k), full_covs)
    final_est_ani = naive_ani if (opt_lambda is None or opt_est_ani is None or no_adj) else opt_est_ani
    return dict(naive_ani=naive_ani, final_est_ani=final_est_ani, final_est_cov=final_est_cov, mean_cov=geq1_mean_cov,
                full_mean_cov=mean_cov, median_cov=median_cov, lambda_status=status, lambda_=lam, max_cov=max_cov,
                n_full=len(full_covs), passed=final_est_ani >= min_ani, contain_count=contain_count, n_kmers=n_genome_kmers)


# contain.rs:410-430: results in the order of the reference's result vector; strict > keeps the first inserted on ties
def winner_table(results):
    """results: [(genome_key, final_est_ani, genome_kmers, tracked_or_None)] -> {kmer: genome_key}"""
    table = {}
    for key, ani, kmers, tracked in results:
        for group in (kmers, tracked or []):
            for km in group:
                v = table.get(km)
                if v is None:
                    table[km] = (ani, key)
                elif ani > v[0]:
                    table[km] = (ani, key)
    return {km: v[1] for km, v in table.items()}


# contain.rs:353-375
def derep_if_reassign_threshold(old_contain, new_contain, n_kmers, ani_thresh=99.0, k=31):
    """True = the genome is kept."""
    threshold = (ani_thresh / 100.0) ** k
    return float(old_contain - new_contain) < threshold * float(n_kmers)


# ---- --estimate-unknown (-u) ------------------------------------------------------------------------------------------------
MED_KMER_FOR_ID_EST = 3.0   # constants.rs:17


# contain.rs:901-951.  `counts`: the sample's k-mer counts in the order they are walked.  The reference walks its hash map
# (hashbrown order, not reproducible here); the host walks the table in ascending k-mer order, and callers of this restatement
# pass the counts in that order.  Only the moving "median" depends on the order; eps does not.
def get_kmer_identity(counts, k, mean_read_length):
    median = 0
    mov_avg_median = 0.0
    n = 1.0
    for count in counts:
        if count > 1:
            median += 1 if count > median else -1
            mov_avg_median += float(median)
            n += 1.0
    mov_avg_median /= n
    num_1s = sum(1 for c in counts if c == 1)
    num_not1s = sum(c for c in counts if c != 1) & 0xFFFFFFFF      # a u32 in the reference
    eps = float(num_not1s) / (float(num_not1s) + float(num_1s) + 0.1)
    if mov_avg_median < MED_KMER_FOR_ID_EST and mean_read_length < 400.0:
        return 0.995 ** float(k)
    return eps if eps < 1.0 else 1.0


# contain.rs:377-390 (covs: the final_est_cov values) -> scaled values
def estimate_true_cov(covs, kmer_id, read_length, k):
    multiplier = read_length / (read_length - float(k) + 1.0)
    return [c / kmer_id * multiplier for c in covs]


# contain.rs:392-408
def estimate_covered_bases(gn_sizes, covs, c, total_counts, read_length, k):
    multiplier = read_length / (read_length - float(k) + 1.0)
    covered = 0.0
    for g, cv in zip(gn_sizes, covs):
        covered += float(g) * cv
    tentative = float(c * total_counts) * multiplier
    if tentative == 0.0:
        return 0.0
    return min(covered / tentative, 1.0)
