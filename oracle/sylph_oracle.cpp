// sylph_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
//
// A CPU restatement of the sketch + profile hot path of bluenote-1577/sylph v0.8.1, written from the
// reference's Rust sources (cited per function as file:line under /root/reference/src).  It exists so
// that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the HIP path
// against the reference's algorithm.  Nothing under sylph_amd/ may import, link or call it.
//
// PARITY STATUS: *parity unpinned against a reference binary*.  The reference is pure Rust, no
// rustc/cargo exists in the build container, and its own tests hold no numeric golden vectors
// (tests/integration_test.rs asserts exit codes, line counts and output equalities only).  The oracle is
// pinned instead against (i) SURVEY.md Appendix A known answers, produced by an independent numpy
// restatement of the same sources (tests/golden/survey_kat.json), and (ii) the reference's relational
// assertions (raw-vs-presketched equality etc.).  Third-party arithmetic that is not under
// /root/reference (statrs 0.16.1 Poisson::cdf, fastrand 2.1.1, scalable_cuckoo_filter 0.2.4) is restated
// from its published definition (Poisson CDF = Q(x+1, lambda)) or left out (bootstrap CI, cuckoo dedup).
//
// Build: see oracle/Makefile  (g++ -O3 -mavx2 -fopenmp -shared -fPIC).

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#if defined(__AVX2__)
#include <immintrin.h>
#endif
#if defined(_OPENMP)
#include <omp.h>
#include <zlib.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// types.rs:50-59  BYTE_TO_SEQ — A/a=0 C/c=1 G/g=2 T/t/U/u=3, raw bytes 1,2,3 -> 1,2,3, all else 0.
// Built programmatically from that description (not copied as a literal table).
// ---------------------------------------------------------------------------------------------
struct ByteToSeq {
    uint8_t t[256];
    ByteToSeq() {
        memset(t, 0, sizeof(t));
        t[1] = 1; t[2] = 2; t[3] = 3;
        t['A'] = t['a'] = 0;
        t['C'] = t['c'] = 1;
        t['G'] = t['g'] = 2;
        t['T'] = t['t'] = 3;
        t['U'] = t['u'] = 3;
    }
};
const ByteToSeq BTS;

// seeding.rs:4-15 mm_hash64 (first step is !(key + (key<<21)), *not* the textbook (!key)+(key<<21);
// avx2_seeding.rs:8-11 does the same: add, then xor with all-ones).
inline uint64_t mm_hash64(uint64_t key) {
    key = ~(key + (key << 21));
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

// seeding.rs:108,142 / avx2_seeding.rs:95 — threshold = u64::MAX / c, comparison is strict '<'.
inline uint64_t fmh_threshold(uint64_t c) { return UINT64_MAX / c; }

struct Seed { uint64_t contig, pos, hash; };

// seeding.rs:86-146 fmh_seeds and :148-209 fmh_seeds_positions (same loop; positions variant pushes
// (contig_number, i, hash) where i is the index of the k-mer's LAST base).
template <class Emit>
void fmh_seeds_scalar(const uint8_t* s, uint64_t len, uint64_t c, uint64_t k, Emit emit) {
    if (len < k) return;                                           // seeding.rs:93
    uint64_t f = 0, r = 0;
    const uint64_t rshift = 2 * (k - 1);                           // :101
    const uint64_t mask = UINT64_MAX >> (64 - 2 * k);              // :102
    const uint64_t rev_mask = ~(3ULL << (2 * k - 2));              // :103
    const uint64_t thr = fmh_threshold(c);
    for (uint64_t i = 0; i + 1 < k; i++) {                         // :109-119 warm-up (no masking)
        uint64_t nf = BTS.t[s[i]], nr = 3 - nf;
        f <<= 2; f |= nf;
        r >>= 2; r |= nr << rshift;
    }
    for (uint64_t i = k - 1; i < len; i++) {                       // :120-145
        uint64_t nf = BTS.t[s[i]], nr = 3 - nf;
        f <<= 2; f |= nf; f &= mask;
        r >>= 2; r &= rev_mask; r |= nr << rshift;
        uint64_t canon = (f < r) ? f : r;                          // :134-139
        uint64_t h = mm_hash64(canon);
        if (h < thr) emit(i, h);                                   // :142 strict
    }
}

// avx2_seeding.rs:33-148 (extract_markers_avx2) and :151-266 (…_positions), restated lane by lane:
// the sequence is cut into 4 chunks of `len4 = (L-k+1)/4` k-mers (:37-41), each chunk is rolled from a
// zero state, the 4 lanes advance in lock-step and emit in lane order within a step (:135-146,
// :253-264).  K-mers starting at >= 4*len4 are never hashed.  min_len guard: k+1 for the plain variant
// (:42), 2k for the positions variant (:160).  k must be 21 or 31 (:46-52, panics otherwise).
template <class Emit>
int fmh_seeds_avx2_compat(const uint8_t* s, uint64_t L, uint64_t c, uint64_t k, uint64_t min_len, Emit emit) {
    if (L < k) return 0;                                           // :34 / :152
    if (L < min_len) return 0;                                     // :42 / :160
    if (!(k == 21 || k == 31)) return -1;                          // :46-52 panic!()
    const uint64_t len4 = (L - k + 1) / 4;
    const uint64_t rshift = 2 * (k - 1);
    const uint64_t mask = UINT64_MAX >> (64 - 2 * k);
    const uint64_t rev_mask = ~(3ULL << (2 * k - 2));
    const uint64_t thr = fmh_threshold(c);
    const uint8_t* str[4] = {s, s + len4, s + 2 * len4, s + 3 * len4};
    uint64_t f[4] = {0, 0, 0, 0}, r[4] = {0, 0, 0, 0};
    for (uint64_t i = 0; i + 1 < k; i++) {                         // :57-79
        for (int j = 0; j < 4; j++) {
            uint64_t nf = BTS.t[str[j][i]], nr = 3 - nf;
            f[j] = (f[j] << 2) | nf;
            r[j] = (r[j] >> 2) | (nr << rshift);
        }
    }
    for (uint64_t i = k - 1; i < len4 + k - 1; i++) {              // :95-147
        uint64_t h[4];
        for (int j = 0; j < 4; j++) {
            uint64_t nf = BTS.t[str[j][i]], nr = 3 - nf;
            f[j] = ((f[j] << 2) | nf) & mask;
            r[j] = ((r[j] >> 2) & rev_mask) | (nr << rshift);
            // :117-120 cmpgt(r,f) signed; blendv picks f where r>f else r.  Values < 2^62 so signed==unsigned.
            uint64_t canon = ((int64_t)r[j] > (int64_t)f[j]) ? f[j] : r[j];
            h[j] = mm_hash64(canon);
        }
        for (int j = 0; j < 4; j++)
            if (h[j] < thr) emit((uint64_t)j * len4 + i, h[j]);    // :253-264 global end index
    }
    return 0;
}

#if defined(__AVX2__)
// Fast variant used only for the cpu_baseline timing leg: real AVX2 intrinsics with the same structure
// as avx2_seeding.rs:6-30,33-148 (4 x u64 lanes, LUT gathers done as 4 scalar loads).  Must produce
// the same output as fmh_seeds_avx2_compat (checked in tests/test_oracle.py).
inline __m256i mm_hash256(__m256i key) {
    key = _mm256_add_epi64(key, _mm256_slli_epi64(key, 21));
    key = _mm256_xor_si256(key, _mm256_cmpeq_epi64(key, key));
    key = _mm256_xor_si256(key, _mm256_srli_epi64(key, 24));
    key = _mm256_add_epi64(_mm256_add_epi64(key, _mm256_slli_epi64(key, 3)), _mm256_slli_epi64(key, 8));
    key = _mm256_xor_si256(key, _mm256_srli_epi64(key, 14));
    key = _mm256_add_epi64(_mm256_add_epi64(key, _mm256_slli_epi64(key, 2)), _mm256_slli_epi64(key, 4));
    key = _mm256_xor_si256(key, _mm256_srli_epi64(key, 28));
    key = _mm256_add_epi64(key, _mm256_slli_epi64(key, 31));
    return key;
}

template <class Emit>
int fmh_seeds_avx2_fast(const uint8_t* s, uint64_t L, uint64_t c, uint64_t k, uint64_t min_len, Emit emit) {
    if (L < k || L < min_len) return 0;
    if (!(k == 21 || k == 31)) return -1;
    const uint64_t len4 = (L - k + 1) / 4;
    const int rshift = (int)(2 * (k - 1));
    const uint64_t thr = fmh_threshold(c);
    const __m256i vmask = _mm256_set1_epi64x((long long)(UINT64_MAX >> (64 - 2 * k)));
    const __m256i vrmask = _mm256_set1_epi64x((long long)~(3ULL << (2 * k - 2)));
    const __m256i three = _mm256_set1_epi64x(3);
    const __m128i sh = _mm_cvtsi32_si128(rshift);
    const uint8_t *s1 = s, *s2 = s + len4, *s3 = s + 2 * len4, *s4 = s + 3 * len4;
    __m256i f = _mm256_setzero_si256(), r = _mm256_setzero_si256();
    for (uint64_t i = 0; i + 1 < k; i++) {
        __m256i nf = _mm256_set_epi64x(BTS.t[s4[i]], BTS.t[s3[i]], BTS.t[s2[i]], BTS.t[s1[i]]);
        __m256i nr = _mm256_sub_epi64(three, nf);
        f = _mm256_or_si256(_mm256_slli_epi64(f, 2), nf);
        r = _mm256_or_si256(_mm256_srli_epi64(r, 2), _mm256_sll_epi64(nr, sh));
    }
    for (uint64_t i = k - 1; i < len4 + k - 1; i++) {
        __m256i nf = _mm256_set_epi64x(BTS.t[s4[i]], BTS.t[s3[i]], BTS.t[s2[i]], BTS.t[s1[i]]);
        __m256i nr = _mm256_sub_epi64(three, nf);
        f = _mm256_and_si256(_mm256_or_si256(_mm256_slli_epi64(f, 2), nf), vmask);
        r = _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi64(r, 2), vrmask), _mm256_sll_epi64(nr, sh));
        __m256i cmp = _mm256_cmpgt_epi64(r, f);
        __m256i canon = _mm256_blendv_epi8(r, f, cmp);
        __m256i h = mm_hash256(canon);
        uint64_t v[4];
        _mm256_storeu_si256((__m256i*)v, h);
        if (v[0] < thr) emit(i, v[0]);
        if (v[1] < thr) emit(len4 + i, v[1]);
        if (v[2] < thr) emit(2 * len4 + i, v[2]);
        if (v[3] < thr) emit(3 * len4 + i, v[3]);
    }
    return 0;
}
#endif

enum SeedMode { MODE_SCALAR = 0, MODE_AVX2_COMPAT = 1, MODE_AVX2_FAST = 2 };

// sketch.rs:53-69 extract_markers: AVX2 path when the host has AVX2, scalar otherwise.
template <class Emit>
int seeds_dispatch(const uint8_t* s, uint64_t L, uint64_t c, uint64_t k, int mode, bool positions, Emit emit) {
    const uint64_t min_len = positions ? 2 * k : k + 1;
    switch (mode) {
        case MODE_SCALAR: fmh_seeds_scalar(s, L, c, k, emit); return 0;
        case MODE_AVX2_COMPAT: return fmh_seeds_avx2_compat(s, L, c, k, min_len, emit);
        case MODE_AVX2_FAST:
#if defined(__AVX2__)
            return fmh_seeds_avx2_fast(s, L, c, k, min_len, emit);
#else
            return fmh_seeds_avx2_compat(s, L, c, k, min_len, emit);
#endif
    }
    return -2;
}

// ---------------------------------------------------------------------------------------------
// sketch.rs:625-656 pair_kmer_single / :659-688 pair_kmer — "locality markers", 16-mers of u32.
// ---------------------------------------------------------------------------------------------
struct Markers { uint32_t m[4]; bool some; };   // m = {f, r, g, t}; set elements are [f,r] and [g,t]

Markers pair_kmer_single(const uint8_t* s, uint64_t len) {
    Markers o{{0, 0, 0, 0}, false};
    const uint64_t k = 16;                                         // size_of::<u32>()*4
    if (len < 4 * k + 2) return o;                                 // :627
    const uint64_t half = len / 2;                                 // :634
    uint32_t f = 0, g = 0, r = 0, t = 0;
    for (uint64_t i = 0; i < k; i++) {                             // :636-653
        f = (f << 2) | BTS.t[s[2 * i]];
        r = (r << 2) | BTS.t[s[2 * i + half]];
        g = (g << 2) | BTS.t[s[1 + 2 * i]];
        t = (t << 2) | BTS.t[s[1 + 2 * i + half]];
    }
    o.m[0] = f; o.m[1] = r; o.m[2] = g; o.m[3] = t; o.some = true;
    return o;
}

Markers pair_kmer(const uint8_t* s1, uint64_t l1, const uint8_t* s2, uint64_t l2) {
    Markers o{{0, 0, 0, 0}, false};
    const uint64_t k = 16;
    if (l1 < 2 * k + 1 || l2 < 2 * k + 1) return o;                // :661
    uint32_t f = 0, g = 0, r = 0, t = 0;
    for (uint64_t i = 0; i < k; i++) {                             // :668-685
        f = (f << 2) | BTS.t[s1[2 * i]];
        r = (r << 2) | BTS.t[s2[2 * i]];
        g = (g << 2) | BTS.t[s1[1 + 2 * i]];
        t = (t << 2) | BTS.t[s2[1 + 2 * i]];
    }
    o.m[0] = f; o.m[1] = r; o.m[2] = g; o.m[3] = t; o.some = true;
    return o;
}

// ---------------------------------------------------------------------------------------------
// Open-addressing containers hashed with the Fx hash (fxhash 0.2.1: h = (rotl(h,5) ^ w) * K per word,
// K = 0x517cc1b727220a95), standing in for FxHashMap<u64,u32> / FxHashSet<(u64,[u32;2])>
// (sketch.rs:5-7, types.rs:9).  Only membership/count semantics matter for parity; Fx hashing is kept
// so the cpu_baseline leg has the reference's memory-access character.
// ---------------------------------------------------------------------------------------------
const uint64_t FX_K = 0x517cc1b727220a95ULL;
inline uint64_t fx_add(uint64_t h, uint64_t w) { return (((h << 5) | (h >> 59)) ^ w) * FX_K; }

struct CountMap {                       // u64 -> u32, linear probing, grows at 7/8 load
    std::vector<uint64_t> keys; std::vector<uint32_t> vals; std::vector<uint8_t> used;
    uint64_t n = 0, cap = 0;
    void init(uint64_t c) { cap = 1; while (cap < c) cap <<= 1; keys.assign(cap, 0); vals.assign(cap, 0); used.assign(cap, 0); n = 0; }
    CountMap() { init(1024); }
    void grow() {
        CountMap o; o.init(cap * 2);
        for (uint64_t i = 0; i < cap; i++) if (used[i]) *o.entry(keys[i]) = vals[i];
        *this = std::move(o);
    }
    uint32_t* entry(uint64_t key) {     // entry(k).or_insert(0)
        if ((n + 1) * 8 > cap * 7) grow();
        uint64_t i = (fx_add(0, key) >> 7) & (cap - 1);
        while (used[i] && keys[i] != key) i = (i + 1) & (cap - 1);
        if (!used[i]) { used[i] = 1; keys[i] = key; vals[i] = 0; n++; }
        return &vals[i];
    }
    const uint32_t* find(uint64_t key) const {
        uint64_t i = (fx_add(0, key) >> 7) & (cap - 1);
        while (used[i] && keys[i] != key) i = (i + 1) & (cap - 1);
        return used[i] ? &vals[i] : nullptr;
    }
};

struct PairSet {                        // set of (u64 kmer, [u32;2])
    std::vector<uint64_t> k1, k2; std::vector<uint8_t> used;
    uint64_t n = 0, cap = 0;
    void init(uint64_t c) { cap = 1; while (cap < c) cap <<= 1; k1.assign(cap, 0); k2.assign(cap, 0); used.assign(cap, 0); n = 0; }
    PairSet() { init(1024); }
    static uint64_t hash(uint64_t a, uint64_t b) { return fx_add(fx_add(fx_add(0, a), b & 0xffffffffULL), b >> 32); }
    void grow() {
        PairSet o; o.init(cap * 2);
        for (uint64_t i = 0; i < cap; i++) if (used[i]) o.insert(k1[i], k2[i]);
        *this = std::move(o);
    }
    bool contains(uint64_t a, uint64_t b) const {
        uint64_t i = (hash(a, b) >> 7) & (cap - 1);
        while (used[i] && !(k1[i] == a && k2[i] == b)) i = (i + 1) & (cap - 1);
        return used[i];
    }
    void insert(uint64_t a, uint64_t b) {
        if ((n + 1) * 8 > cap * 7) grow();
        uint64_t i = (hash(a, b) >> 7) & (cap - 1);
        while (used[i] && !(k1[i] == a && k2[i] == b)) i = (i + 1) & (cap - 1);
        if (!used[i]) { used[i] = 1; k1[i] = a; k2[i] = b; n++; }
    }
};

// sketch.rs:690-731 dup_removal_lsh_full_exact.
inline void dup_removal_lsh_full_exact(CountMap& counts, PairSet& set, uint64_t km, const Markers& pair,
                                       uint64_t& num_dup_removed, bool no_dedup, bool has_threshold,
                                       uint32_t threshold) {
    uint32_t* c = counts.entry(km);                                // :701
    uint32_t c_threshold = has_threshold ? threshold : UINT32_MAX; // :702-705
    if (!no_dedup && *c < c_threshold) {                           // :706
        if (pair.some) {                                           // :707
            bool ret = false;
            const uint64_t m0 = (uint64_t)pair.m[0] | ((uint64_t)pair.m[1] << 32);
            const uint64_t m1 = (uint64_t)pair.m[2] | ((uint64_t)pair.m[3] << 32);
            if (set.contains(km, m0)) { if (*c > 0) ret = true; }  // :709-714
            else set.insert(km, m0);
            if (set.contains(km, m1)) { if (*c > 0) ret = true; }  // :716-722
            else set.insert(km, m1);
            if (ret) { num_dup_removed++; return; }                // :723-726
        }
    }
    *c += 1;                                                       // :730
}

struct ReadSketch {
    CountMap counts;
    uint64_t num_dup_removed = 0;
    double mean_read_length = 0.0;
    uint64_t n_records = 0;
    std::vector<uint64_t> sorted_keys; std::vector<uint32_t> sorted_vals;
    void finalize() {
        std::vector<std::pair<uint64_t, uint32_t>> v;
        v.reserve(counts.n);
        for (uint64_t i = 0; i < counts.cap; i++) if (counts.used[i]) v.emplace_back(counts.keys[i], counts.vals[i]);
        std::sort(v.begin(), v.end());
        sorted_keys.resize(v.size()); sorted_vals.resize(v.size());
        for (size_t i = 0; i < v.size(); i++) { sorted_keys[i] = v[i].first; sorted_vals[i] = v[i].second; }
    }
};

// sketch.rs:897-959 sketch_sequences_needle, operating on already-parsed records
// (bases concatenated, record i = bases[off[i], off[i+1])).
int sketch_single(ReadSketch& sk, const uint8_t* bases, const uint64_t* off, uint64_t n_reads, uint64_t c,
                  uint64_t k, int mode, bool no_dedup) {
    PairSet set;
    std::vector<uint64_t> vec;
    double mean = 0.0, counter = 0.0;
    for (uint64_t i = 0; i < n_reads; i++) {
        const uint8_t* seq = bases + off[i];
        const uint64_t len = off[i + 1] - off[i];
        vec.clear();
        Markers kmer_pair{{0, 0, 0, 0}, false};
        if (len <= 400) kmer_pair = pair_kmer_single(seq, len);   // :922-927 (MAX 400 literal)
        int rc = seeds_dispatch(seq, len, c, k, mode, false, [&](uint64_t, uint64_t h) { vec.push_back(h); });
        if (rc) return rc;
        for (uint64_t km : vec)                                    // :929-939, threshold Some(MAX_DEDUP_COUNT=4)
            dup_removal_lsh_full_exact(sk.counts, set, km, kmer_pair, sk.num_dup_removed, no_dedup, true, 4);
        counter += 1.0;                                            // :941-943
        mean = mean + ((double)len - mean) / counter;
    }
    sk.mean_read_length = mean;
    sk.n_records = n_reads;
    return 0;
}

// sketch.rs:771-895 sketch_pair_sequences with dedup_fpr == 0 (exact set, :829-838); records are
// interleaved: record 2p = mate 1 of pair p, record 2p+1 = mate 2.
int sketch_paired(ReadSketch& sk, const uint8_t* bases, const uint64_t* off, uint64_t n_pairs, uint64_t c,
                  uint64_t k, int mode, bool no_dedup) {
    PairSet set;
    std::vector<uint64_t> v1, v2;
    double mean = 0.0, counter = 0.0;
    for (uint64_t p = 0; p < n_pairs; p++) {
        const uint8_t* s1 = bases + off[2 * p];
        const uint64_t l1 = off[2 * p + 1] - off[2 * p];
        const uint8_t* s2 = bases + off[2 * p + 1];
        const uint64_t l2 = off[2 * p + 2] - off[2 * p + 1];
        v1.clear(); v2.clear();
        int rc = seeds_dispatch(s1, l1, c, k, mode, false, [&](uint64_t, uint64_t h) { v1.push_back(h); });  // :819
        if (rc) return rc;
        rc = seeds_dispatch(s2, l2, c, k, mode, false, [&](uint64_t, uint64_t h) { v2.push_back(h); });      // :820
        if (rc) return rc;
        Markers kmer_pair = pair_kmer(s1, l1, s2, l2);             // :821
        counter += 1.0;                                            // :824-826 (mate-1 length only)
        mean = mean + ((double)l1 - mean) / counter;
        for (uint64_t km : v1)                                     // :828-849, threshold None
            dup_removal_lsh_full_exact(sk.counts, set, km, kmer_pair, sk.num_dup_removed, no_dedup, false, 0);
        for (uint64_t km : v2) {                                   // :851-875
            if (std::find(v1.begin(), v1.end(), km) != v1.end()) continue;   // :852
            dup_removal_lsh_full_exact(sk.counts, set, km, kmer_pair, sk.num_dup_removed, no_dedup, false, 0);
        }
    }
    sk.mean_read_length = mean;
    sk.n_records = n_pairs;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// a10: the APPROXIMATE paired-end dedup the reference uses by default (sketch.rs:733-769 dup_removal_lsh_full over a
// scalable_cuckoo_filter 0.2.4, built at :796-804 with initial capacity 10^7 and --fpr 1e-4).  The crate is NOT in /root/reference:
// what follows is the published structure (Fan et al., "Cuckoo Filter: Practically Better Than Bloom", CoNEXT 2014) with the
// crate's defaults AS DOCUMENTED — 4 entries per bucket, fingerprint width ceil(log2(1/fpr) + log2(2 * 4)) bits, 512 kicks, a
// further filter of twice the capacity and fpr * 0.9 when one fills up.  Its hash bits, eviction choices and growth points are
// not the crate's: this is a MODEL of the default path for measuring how far the exact set (what the GPU implements) is from it,
// not a parity target ("parity unpinned").
struct CuckooFilter {
    std::vector<uint32_t> slots;            // n_buckets x 4 fingerprints (0 = empty)
    uint64_t n_buckets = 0, n_items = 0, capacity = 0;
    int fp_bits = 17;
    uint64_t rng = 0x9E3779B97F4A7C15ULL;
    void init(uint64_t cap, double fpr) {
        capacity = cap;
        fp_bits = (int)std::ceil(std::log2(1.0 / fpr) + std::log2(8.0));
        if (fp_bits > 31) fp_bits = 31;
        n_buckets = 1;
        while (n_buckets * 4 < cap) n_buckets <<= 1;       // next power of two of capacity / entries per bucket
        slots.assign(n_buckets * 4, 0);
        n_items = 0;
    }
    uint32_t fingerprint(uint64_t h) const { const uint32_t f = (uint32_t)(h >> 32) & ((1u << fp_bits) - 1u); return f ? f : 1u; }
    uint64_t alt(uint64_t i, uint32_t f) const { return (i ^ (fx_add(0, f) >> 11)) & (n_buckets - 1); }
    bool contains(uint64_t h) const {
        const uint32_t f = fingerprint(h);
        const uint64_t i1 = h & (n_buckets - 1), i2 = alt(i1, f);
        for (int e = 0; e < 4; e++) if (slots[i1 * 4 + e] == f || slots[i2 * 4 + e] == f) return true;
        return false;
    }
    bool insert(uint64_t h) {
        uint32_t f = fingerprint(h);
        uint64_t i = h & (n_buckets - 1);
        for (int side = 0; side < 2; side++) {
            const uint64_t b = side ? alt(i, f) : i;
            for (int e = 0; e < 4; e++) if (!slots[b * 4 + e]) { slots[b * 4 + e] = f; n_items++; return true; }
        }
        for (int kick = 0; kick < 512; kick++) {
            rng = rng * 6364136223846793005ULL + 1442695040888963407ULL;
            const int e = (int)(rng >> 62);
            std::swap(f, slots[i * 4 + e]);
            i = alt(i, f);
            for (int e2 = 0; e2 < 4; e2++) if (!slots[i * 4 + e2]) { slots[i * 4 + e2] = f; n_items++; return true; }
        }
        return false;                                       // full (the displaced fingerprint is lost, as in the paper's algorithm)
    }
};
struct ScalableCuckoo {
    std::vector<CuckooFilter> filters;
    double fpr;
    uint64_t cap0;
    ScalableCuckoo(uint64_t initial_capacity, double f) : fpr(f), cap0(initial_capacity) { grow(); }
    void grow() {
        CuckooFilter cf;
        const size_t n = filters.size();
        cf.init(cap0 << n, fpr * std::pow(0.9, (double)n));
        filters.push_back(std::move(cf));
    }
    static uint64_t hash(uint64_t a, uint64_t b) { return PairSet::hash(a, b) * 0x9E3779B97F4A7C15ULL; }   // (FxHasher of the tuple, mixed)
    bool contains(uint64_t a, uint64_t b) const {
        const uint64_t h = hash(a, b);
        for (const auto& f : filters) if (f.contains(h)) return true;
        return false;
    }
    void insert(uint64_t a, uint64_t b) {
        const uint64_t h = hash(a, b);
        if (filters.back().n_items >= filters.back().capacity || !filters.back().insert(h)) { grow(); filters.back().insert(h); }
    }
};
// sketch.rs:733-769 dup_removal_lsh_full
inline void dup_removal_lsh_full(CountMap& counts, ScalableCuckoo& set, uint64_t km, const Markers& pair, uint64_t& num_dup_removed,
                                 bool no_dedup) {
    uint32_t* c = counts.entry(km);                                // :743
    if (!no_dedup && pair.some) {                                  // :744-745
        bool ret = false;
        const uint64_t m0 = (uint64_t)pair.m[0] | ((uint64_t)pair.m[1] << 32);
        const uint64_t m1 = (uint64_t)pair.m[2] | ((uint64_t)pair.m[3] << 32);
        if (set.contains(km, m0)) { if (*c > 0) ret = true; }      // :747-753
        else set.insert(km, m0);
        if (set.contains(km, m1)) { if (*c > 0) ret = true; }      // :754-760
        else set.insert(km, m1);
        if (ret) { num_dup_removed++; return; }                    // :761-764
    }
    *c += 1;                                                       // :767
}
// sketch.rs:771-895 with dedup_fpr != 0 (the default): the same record loop as sketch_paired, the filter in place of the set
int sketch_paired_cuckoo(ReadSketch& sk, const uint8_t* bases, const uint64_t* off, uint64_t n_pairs, uint64_t c, uint64_t k, int mode,
                         double fpr, uint64_t initial_capacity) {
    ScalableCuckoo set(initial_capacity, fpr);
    std::vector<uint64_t> v1, v2;
    double mean = 0.0, counter = 0.0;
    for (uint64_t p = 0; p < n_pairs; p++) {
        const uint8_t* s1 = bases + off[2 * p];
        const uint64_t l1 = off[2 * p + 1] - off[2 * p];
        const uint8_t* s2 = bases + off[2 * p + 1];
        const uint64_t l2 = off[2 * p + 2] - off[2 * p + 1];
        v1.clear(); v2.clear();
        int rc = seeds_dispatch(s1, l1, c, k, mode, false, [&](uint64_t, uint64_t h) { v1.push_back(h); });
        if (rc) return rc;
        rc = seeds_dispatch(s2, l2, c, k, mode, false, [&](uint64_t, uint64_t h) { v2.push_back(h); });
        if (rc) return rc;
        Markers kmer_pair = pair_kmer(s1, l1, s2, l2);
        counter += 1.0;
        mean = mean + ((double)l1 - mean) / counter;
        for (uint64_t km : v1) dup_removal_lsh_full(sk.counts, set, km, kmer_pair, sk.num_dup_removed, false);
        for (uint64_t km : v2) {
            if (std::find(v1.begin(), v1.end(), km) != v1.end()) continue;   // :852
            dup_removal_lsh_full(sk.counts, set, km, kmer_pair, sk.num_dup_removed, false);
        }
    }
    sk.mean_read_length = mean;
    sk.n_records = n_pairs;
    return 0;
}

// sketch.rs:550-622 sketch_genome on already-parsed contigs (contig i = bases[off[i], off[i+1])).
// types.rs:88-90 MMHashSet only decides membership, so std::unordered_set is equivalent.
struct GenomeSketchO {
    std::vector<uint64_t> genome_kmers, tracked;
    uint64_t gn_size = 0, n_raw_seeds = 0, n_dup_kmers = 0;
};

int sketch_genome(GenomeSketchO& g, const uint8_t* bases, const uint64_t* off, uint64_t n_contigs, uint64_t c,
                  uint64_t k, int mode, uint64_t min_spacing, bool pseudotax) {
    std::vector<Seed> vec;
    for (uint64_t ci = 0; ci < n_contigs; ci++) {
        const uint64_t len = off[ci + 1] - off[ci];
        g.gn_size += len;                                          // :581
        int rc = seeds_dispatch(bases + off[ci], len, c, k, mode, true,
                                [&](uint64_t pos, uint64_t h) { vec.push_back(Seed{ci, pos, h}); });   // :582
        if (rc) return rc;
    }
    g.n_raw_seeds = vec.size();
    std::sort(vec.begin(), vec.end(), [](const Seed& a, const Seed& b) {                               // :593
        if (a.contig != b.contig) return a.contig < b.contig;
        if (a.pos != b.pos) return a.pos < b.pos;
        return a.hash < b.hash;
    });
    std::unordered_set<uint64_t> kmer_set, duplicate_set;
    for (const Seed& s : vec) {                                    // :594-600
        if (!kmer_set.count(s.hash)) kmer_set.insert(s.hash);
        else duplicate_set.insert(s.hash);
    }
    g.n_dup_kmers = duplicate_set.size();
    uint64_t last_pos = 0, last_contig = 0;
    for (const Seed& s : vec) {                                    // :602-614
        if (duplicate_set.count(s.hash)) continue;
        if (last_pos == 0 || last_contig != s.contig || s.pos - last_pos > min_spacing) {
            g.genome_kmers.push_back(s.hash);
            last_contig = s.contig;
            last_pos = s.pos;
        } else if (pseudotax) {
            g.tracked.push_back(s.hash);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// statrs 0.16.1 Poisson::cdf(x) = gamma_ur(x+1, lambda) (third-party, not under /root/reference;
// call site contain.rs:664-669).  Restated as the regularised upper incomplete gamma Q(a,x) by the
// textbook series / Lentz continued fraction.  PARITY UNPINNED — only used through the comparison
// `< CUTOFF_PVALUE`; SURVEY.md Appendix A lists the resulting cut-offs (scipy cross-check).
// ---------------------------------------------------------------------------------------------
double gamma_q(double a, double x) {
    if (x <= 0.0) return 1.0;
    const double gln = std::lgamma(a);
    if (x < a + 1.0) {                                             // series for P, Q = 1-P
        double ap = a, sum = 1.0 / a, del = sum;
        for (int n = 0; n < 10000; n++) {
            ap += 1.0; del *= x / ap; sum += del;
            if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
        }
        return 1.0 - sum * std::exp(-x + a * std::log(x) - gln);
    }
    const double FPMIN = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / FPMIN, d = 1.0 / b, h = d;
    for (int i = 1; i < 10000; i++) {
        double an = -(double)i * ((double)i - a);
        b += 2.0;
        d = an * d + b; if (std::fabs(d) < FPMIN) d = FPMIN;
        c = b + an / c; if (std::fabs(c) < FPMIN) c = FPMIN;
        d = 1.0 / d;
        double del = d * c; h *= del;
        if (std::fabs(del - 1.0) < 1e-16) break;
    }
    return std::exp(-x + a * std::log(x) - gln) * h;
}
inline double poisson_cdf(double lambda, uint64_t x) { return gamma_q((double)x + 1.0, lambda); }

// inference.rs:207-242 ratio_lambda.
bool ratio_lambda(const std::vector<uint32_t>& full_covs, double min_count_correct, double& out) {
    uint64_t num_zero = 0;
    std::map<uint64_t, uint64_t> count_map;
    for (uint32_t x : full_covs) { if (x == 0) num_zero++; else count_map[x]++; }   // :211-218
    if (count_map.size() == 1) return false;                       // :221
    if (full_covs.size() - num_zero < 25) return false;            // :225 SAMPLE_SIZE_CUTOFF (constants.rs:4)
    if (count_map.empty()) return false;                           // (unreachable: size-zero implies <25)
    // :228-230 sort (count, value) descending, take first: max count, ties -> larger value
    uint64_t best_cnt = 0, most_ind = 0;
    for (auto& kv : count_map)
        if (kv.second > best_cnt || (kv.second == best_cnt && kv.first > most_ind)) { best_cnt = kv.second; most_ind = kv.first; }
    auto it = count_map.find(most_ind + 1);
    if (it == count_map.end()) return false;                       // :231
    const double count_p1 = (double)it->second, count = (double)count_map[most_ind];
    if (count_p1 < min_count_correct || count < min_count_correct) return false;     // :236
    out = count_p1 / count * (double)(most_ind + 1);               // :239
    return true;
}

// contain.rs:817-847 ani_from_lambda.
bool ani_from_lambda(bool has_lambda, double lambda, double k, const std::vector<uint32_t>& full_cov, double& out) {
    if (!has_lambda) return false;
    uint64_t contain_count = 0;
    for (uint32_t x : full_cov) if (x != 0) contain_count++;
    const double adj_index = (double)contain_count / (1.0 - std::exp(-lambda)) / (double)full_cov.size();
    const double ani = std::pow(adj_index, 1.0 / k);
    if (ani < 0.0 || std::isnan(ani)) return false;
    out = ani;
    return true;
}

}  // namespace

// =================================================================================================
// C ABI for ctypes (tests/, bench.py cpu_baseline).  All buffers are caller-owned unless noted.
// =================================================================================================
extern "C" {

uint64_t orc_mm_hash64(uint64_t key) { return mm_hash64(key); }
uint64_t orc_threshold(uint64_t c) { return fmh_threshold(c); }
uint8_t orc_byte_to_seq(uint8_t b) { return BTS.t[b]; }
int orc_has_avx2(void) {
#if defined(__AVX2__)
    return __builtin_cpu_supports("avx2") ? 1 : 0;
#else
    return 0;
#endif
}

// extract_markers (sketch.rs:53).  Returns the number of seeds (may exceed cap; only cap are written),
// or a negative code if the reference would panic (k not in {21,31} under AVX2).
int64_t orc_extract_markers(const uint8_t* s, uint64_t len, uint64_t c, uint64_t k, int mode, uint64_t* out,
                            uint64_t cap) {
    uint64_t n = 0;
    int rc = seeds_dispatch(s, len, c, k, mode, false, [&](uint64_t, uint64_t h) { if (n < cap) out[n] = h; n++; });
    return rc ? rc : (int64_t)n;
}

// extract_markers_positions (sketch.rs:71).  pos = index of the k-mer's last base.
int64_t orc_extract_markers_positions(const uint8_t* s, uint64_t len, uint64_t c, uint64_t k, int mode,
                                      uint64_t* out_pos, uint64_t* out_hash, uint64_t cap) {
    uint64_t n = 0;
    int rc = seeds_dispatch(s, len, c, k, mode, true, [&](uint64_t p, uint64_t h) {
        if (n < cap) { out_pos[n] = p; out_hash[n] = h; }
        n++;
    });
    return rc ? rc : (int64_t)n;
}

int orc_pair_kmer_single(const uint8_t* s, uint64_t len, uint32_t out[4]) {
    Markers m = pair_kmer_single(s, len);
    memcpy(out, m.m, sizeof(m.m));
    return m.some ? 1 : 0;
}
int orc_pair_kmer(const uint8_t* s1, uint64_t l1, const uint8_t* s2, uint64_t l2, uint32_t out[4]) {
    Markers m = pair_kmer(s1, l1, s2, l2);
    memcpy(out, m.m, sizeof(m.m));
    return m.some ? 1 : 0;
}

// Read sketches.  paired=0: sketch_sequences_needle; paired=1: sketch_pair_sequences (--fpr 0), records interleaved.
void* orc_sketch_reads(const uint8_t* bases, const uint64_t* off, uint64_t n_records, uint64_t c, uint64_t k, int mode,
                       int paired, int no_dedup) {
    ReadSketch* sk = new ReadSketch();
    int rc = paired ? sketch_paired(*sk, bases, off, n_records / 2, c, k, mode, no_dedup != 0)
                    : sketch_single(*sk, bases, off, n_records, c, k, mode, no_dedup != 0);
    if (rc) { delete sk; return nullptr; }
    sk->finalize();
    return sk;
}
// the default (approximate) paired dedup, as modelled above
void* orc_sketch_reads_cuckoo(const uint8_t* bases, const uint64_t* off, uint64_t n_records, uint64_t c, uint64_t k, int mode, double fpr,
                              uint64_t initial_capacity) {
    ReadSketch* sk = new ReadSketch();
    if (sketch_paired_cuckoo(*sk, bases, off, n_records / 2, c, k, mode, fpr, initial_capacity)) { delete sk; return nullptr; }
    sk->finalize();
    return sk;
}
// ---- the reference's `sylph sketch` FROM FILES, as the CPU runs it (bench.py's cpu_baseline_from_files; round 6) --------------------
// sketch.rs:313 / :371 hand every sample to ONE rayon worker, which reads its file(s) record by record through needletail — flate2
// inflating gzip input on the same thread — and sketches as it goes.  Here: one thread per sample reads the file(s) through zlib's
// gzread (plain text passes through it unchanged), cuts the four-line records, and sketches the sample with the fast seeding variant
// and the paired dedup the flags select (fpr = 0: exact set; else the filter model).  seconds[s]: wall clock of sample s alone, from
// opening its files to its finished table.  Returns 0, or the 1-based index of the first sample that could not be read.
static bool read_fastq_records(const char* path, std::vector<uint8_t>& bases, std::vector<uint64_t>& lens) {
    gzFile f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1u << 20);
    std::vector<uint8_t> buf(8u << 20);
    std::vector<uint8_t> carry;
    int line = 0;
    uint64_t cur = 0;
    for (;;) {
        const int got = gzread(f, buf.data(), (unsigned)buf.size());
        if (got < 0) { gzclose(f); return false; }
        if (got == 0) break;
        const uint8_t* p = buf.data();
        const uint8_t* end = p + got;
        while (p < end) {
            const uint8_t* nl = (const uint8_t*)memchr(p, '\n', (size_t)(end - p));
            const uint8_t* stop = nl ? nl : end;
            if ((line & 3) == 1) { bases.insert(bases.end(), p, stop); cur += (uint64_t)(stop - p); }
            if (!nl) break;
            if ((line & 3) == 1) {
                if (cur && bases.back() == '\r') { bases.pop_back(); cur--; }
                lens.push_back(cur);
                cur = 0;
            }
            line++;
            p = nl + 1;
        }
    }
    gzclose(f);
    return true;
}
int orc_sketch_files(const char* const* f1, const char* const* f2, uint64_t n_samples, uint64_t c, uint64_t k, int mode, double fpr, int threads,
                     double* seconds, uint64_t* table_sizes, uint64_t* n_bases_out) {
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&] {
        for (uint64_t s = next++; s < n_samples; s = next++) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<uint8_t> b1, b2;
            std::vector<uint64_t> l1, l2;
            if (!read_fastq_records(f1[s], b1, l1) || (f2 && !read_fastq_records(f2[s], b2, l2))) { int z = 0; failed.compare_exchange_strong(z, (int)s + 1); continue; }
            ReadSketch sk;
            uint64_t nb = b1.size() + b2.size();
            if (f2) {
                // interleave the mates' records the way orc_sketch_reads takes pairs (the reference's lock-step readers, sketch.rs:813-815)
                const uint64_t n = std::min(l1.size(), l2.size());
                std::vector<uint8_t> bases;
                bases.reserve(nb + 64);
                std::vector<uint64_t> off(2 * n + 1, 0);
                uint64_t a = 0, b = 0;
                for (uint64_t i = 0; i < n; i++) {
                    bases.insert(bases.end(), b1.begin() + a, b1.begin() + a + l1[i]); a += l1[i];
                    off[2 * i + 1] = bases.size();
                    bases.insert(bases.end(), b2.begin() + b, b2.begin() + b + l2[i]); b += l2[i];
                    off[2 * i + 2] = bases.size();
                }
                bases.resize(bases.size() + 64, 'A');
                std::vector<uint8_t>().swap(b1);
                std::vector<uint8_t>().swap(b2);
                if (fpr == 0.) sketch_paired(sk, bases.data(), off.data(), n, c, k, mode, false);
                else sketch_paired_cuckoo(sk, bases.data(), off.data(), n, c, k, mode, fpr, 10000000);
            } else {
                std::vector<uint64_t> off(l1.size() + 1, 0);
                for (size_t i = 0; i < l1.size(); i++) off[i + 1] = off[i] + l1[i];
                b1.resize(b1.size() + 64, 'A');
                sketch_single(sk, b1.data(), off.data(), l1.size(), c, k, mode, false);
            }
            sk.finalize();
            seconds[s] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (table_sizes) table_sizes[s] = sk.sorted_keys.size();
            if (n_bases_out) n_bases_out[s] = nb;
        }
    };
    std::vector<std::thread> pool;
    const int T = std::max(1, std::min<int>(threads, (int)n_samples));
    for (int t = 1; t < T; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    return failed.load();
}

// The filter alone, walked as sketch.rs:747-760 walks it — test, insert when absent — over a stream of (k-mer, markers) items:
// contained[i] = what `contains` answered for item i; returns the number of filters at the end.  (The checker of the
// data-parallel formulation in sylph_amd/csrc/a10.hip: tests/test_oracle.py restates that formulation in numpy against this walk.)
uint64_t orc_cuckoo_walk(const uint64_t* km, const uint64_t* marker, uint64_t n, double fpr, uint64_t initial_capacity, uint8_t* contained) {
    ScalableCuckoo set(initial_capacity, fpr);
    for (uint64_t i = 0; i < n; i++) {
        contained[i] = set.contains(km[i], marker[i]) ? 1 : 0;
        if (!contained[i]) set.insert(km[i], marker[i]);
    }
    return set.filters.size();
}
// The items sketch.rs:828-867 hands to the filter, in the order it does: per pair the seeds of mate 1 as extract_markers pushed
// them, then those of mate 2 that mate 1 did not produce; two items per seed (markers.0, markers.1); nothing for pairs without
// markers.  rec / seed = record index and the seed's place in its record's emission order.  Returns the number of items
// (call with null outputs to size them).
uint64_t orc_pair_filter_items(const uint8_t* bases, const uint64_t* off, uint64_t n_records, uint64_t c, uint64_t k, int mode,
                               uint64_t* km, uint64_t* marker, uint64_t* rec, uint32_t* seed) {
    uint64_t n = 0;
    std::vector<uint64_t> v1, v2;
    for (uint64_t p = 0; p < n_records / 2; p++) {
        const uint8_t* s1 = bases + off[2 * p];
        const uint64_t l1 = off[2 * p + 1] - off[2 * p];
        const uint8_t* s2 = bases + off[2 * p + 1];
        const uint64_t l2 = off[2 * p + 2] - off[2 * p + 1];
        v1.clear(); v2.clear();
        if (seeds_dispatch(s1, l1, c, k, mode, false, [&](uint64_t, uint64_t h) { v1.push_back(h); })) return ~0ull;
        if (seeds_dispatch(s2, l2, c, k, mode, false, [&](uint64_t, uint64_t h) { v2.push_back(h); })) return ~0ull;
        const Markers pair = pair_kmer(s1, l1, s2, l2);
        if (!pair.some) continue;
        const uint64_t m[2] = {(uint64_t)pair.m[0] | ((uint64_t)pair.m[1] << 32), (uint64_t)pair.m[2] | ((uint64_t)pair.m[3] << 32)};
        auto put = [&](uint64_t h, uint64_t r, uint32_t e) {
            for (int w = 0; w < 2; w++) {
                if (km) { km[n] = h; marker[n] = m[w]; rec[n] = r; seed[n] = e; }
                n++;
            }
        };
        for (size_t e = 0; e < v1.size(); e++) put(v1[e], 2 * p, (uint32_t)e);
        for (size_t e = 0; e < v2.size(); e++)
            if (std::find(v1.begin(), v1.end(), v2[e]) == v1.end()) put(v2[e], 2 * p + 1, (uint32_t)e);
    }
    return n;
}
uint64_t orc_sketch_size(void* h) { return ((ReadSketch*)h)->sorted_keys.size(); }
uint64_t orc_sketch_dup_removed(void* h) { return ((ReadSketch*)h)->num_dup_removed; }
double orc_sketch_mean_read_length(void* h) { return ((ReadSketch*)h)->mean_read_length; }
void orc_sketch_copy(void* h, uint64_t* kmers, uint32_t* counts) {   // ascending k-mer order
    ReadSketch* sk = (ReadSketch*)h;
    memcpy(kmers, sk->sorted_keys.data(), sk->sorted_keys.size() * 8);
    memcpy(counts, sk->sorted_vals.data(), sk->sorted_vals.size() * 4);
}
void orc_sketch_free(void* h) { delete (ReadSketch*)h; }

// Genome sketch (sketch_genome, sketch.rs:550).
void* orc_sketch_genome(const uint8_t* bases, const uint64_t* off, uint64_t n_contigs, uint64_t c, uint64_t k, int mode,
                        uint64_t min_spacing, int pseudotax) {
    GenomeSketchO* g = new GenomeSketchO();
    if (sketch_genome(*g, bases, off, n_contigs, c, k, mode, min_spacing, pseudotax != 0)) { delete g; return nullptr; }
    return g;
}
uint64_t orc_genome_n_kmers(void* h) { return ((GenomeSketchO*)h)->genome_kmers.size(); }
uint64_t orc_genome_n_tracked(void* h) { return ((GenomeSketchO*)h)->tracked.size(); }
uint64_t orc_genome_gn_size(void* h) { return ((GenomeSketchO*)h)->gn_size; }
uint64_t orc_genome_n_raw_seeds(void* h) { return ((GenomeSketchO*)h)->n_raw_seeds; }
uint64_t orc_genome_n_dup_kmers(void* h) { return ((GenomeSketchO*)h)->n_dup_kmers; }
void orc_genome_copy(void* h, uint64_t* kmers, uint64_t* tracked) {
    GenomeSketchO* g = (GenomeSketchO*)h;
    if (kmers) memcpy(kmers, g->genome_kmers.data(), g->genome_kmers.size() * 8);
    if (tracked) memcpy(tracked, g->tracked.data(), g->tracked.size() * 8);
}
void orc_genome_free(void* h) { delete (GenomeSketchO*)h; }

// ---------------------------------------------------------------------------------------------
// Containment.  A sample is loaded once into an Fx-hashed map (contain.rs:559 deserialises into
// FxHashMap<u64,u32>); orc_contain then restates the probe loop contain.rs:624-656 for a batch of genomes
// (the rayon par_iter over genomes at contain.rs:284 becomes an OpenMP loop when n_threads > 1).
// winner_gid: optional per-(genome k-mer) owner id array implementing the winner-map variant (:637-646):
// pass NULL for the first pass; otherwise winner_of[j] is the genome id that owns db k-mer j.
// ---------------------------------------------------------------------------------------------
void* orc_sample_load(const uint64_t* kmers, const uint32_t* counts, uint64_t n) {
    CountMap* m = new CountMap();
    m->init(n * 2 + 16);
    for (uint64_t i = 0; i < n; i++) *m->entry(kmers[i]) = counts[i];
    return m;
}
void orc_sample_free(void* h) { delete (CountMap*)h; }

// Outputs: contain_count[g]; covs written per genome at cov_out + genome_off[g] (capacity = genome length,
// genome order preserved, contain.rs:649/645); kmers_lost[g] (winner variant) may be NULL.
void orc_contain(void* sample, const uint64_t* db_kmers, const uint64_t* genome_off, uint64_t n_genomes,
                 double min_number_kmers, const uint32_t* winner_of, const uint32_t* genome_ids,
                 uint32_t* contain_count, uint32_t* cov_out, uint32_t* kmers_lost, int n_threads) {
    const CountMap* m = (const CountMap*)sample;
    (void)n_threads;
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t g = 0; g < (int64_t)n_genomes; g++) {
        const uint64_t b = genome_off[g], e = genome_off[g + 1];
        uint32_t cc = 0, lost = 0;
        if ((double)(e - b) >= min_number_kmers) {                 // contain.rs:627
            for (uint64_t j = b; j < e; j++) {                     // :632-652
                const uint32_t* v = m->find(db_kmers[j]);
                if (!v || *v == 0) continue;                       // :633-636
                if (winner_of && winner_of[j] != (genome_ids ? genome_ids[g] : (uint32_t)g)) { lost++; continue; }  // :637-642
                cov_out[b + cc] = *v;
                cc++;
            }
        }
        contain_count[g] = cc;
        if (kmers_lost) kmers_lost[g] = lost;
    }
}

// ---------------------------------------------------------------------------------------------
// Statistics half of get_stats (contain.rs:657-813) for one genome, default estimator (ratio_lambda).
// ---------------------------------------------------------------------------------------------
struct OrcStats {
    double naive_ani, final_est_ani, final_est_cov, mean_cov /*geq1*/, median_cov, lambda, max_cov, full_mean_cov;
    int32_t lambda_status;   // 0 = Low, 1 = High, 2 = Lambda(x)
    int32_t passed;          // 1 if an AniResult would be returned
    uint64_t contain_count, n_kmers, n_full;
};

int orc_stats(const uint32_t* covs_in, uint64_t n_covs, uint64_t n_genome_kmers, uint64_t k, double min_count_correct,
              double min_ani, int no_adj, int mean_coverage, OrcStats* out) {
    memset(out, 0, sizeof(*out));
    out->n_kmers = n_genome_kmers;
    if (n_covs == 0) return 0;                                     // :654
    const uint64_t contain_count = n_covs;
    std::vector<uint32_t> covs(covs_in, covs_in + n_covs);
    const double naive_ani = std::pow((double)contain_count / (double)n_genome_kmers, 1.0 / (double)k);   // :657-660
    std::sort(covs.begin(), covs.end());                           // :661
    const double median_cov = (double)covs[covs.size() / 2];       // :663
    double max_cov = 1.7976931348623157e308;                       // f64::MAX :665
    if (median_cov < 30.0) {                                       // :666-675
        for (size_t i = covs.size() / 2; i < covs.size(); i++) {
            if (poisson_cdf(median_cov, covs[i]) < 0.9999999999) max_cov = (double)covs[i];   // constants.rs:3
            else break;
        }
    }
    std::vector<uint32_t> full(n_genome_kmers - contain_count, 0); // :679
    for (uint32_t cv : covs) if ((double)cv <= max_cov) full.push_back(cv);   // :680-684
    uint32_t sum = 0; for (uint32_t x : full) sum += x;            // u32 wrapping sum as in release Rust
    const double mean_cov = (double)sum / (double)full.size();     // :689
    const double geq1 = (double)sum / (double)covs.size();         // :690
    int status; double lam = 0.0;
    if (median_cov > 2.0) status = 1;                              // :692-694 MEDIAN_ANI_THRESHOLD
    else status = ratio_lambda(full, min_count_correct, lam) ? 2 : 0;   // :695-713
    double final_est_cov;
    if (status == 2) final_est_cov = lam;                          // :717-728
    else if (median_cov < 15.0) final_est_cov = geq1;
    else final_est_cov = mean_coverage ? geq1 : median_cov;
    const bool has_lambda = (status == 2);                         // :730-735
    double est_ani = 0.0;
    const bool has_est = ani_from_lambda(has_lambda, final_est_cov, (double)k, full, est_ani);   // :737
    const double final_est_ani = (!has_lambda || !has_est || no_adj) ? naive_ani : est_ani;      // :739-744
    out->naive_ani = naive_ani; out->final_est_ani = final_est_ani; out->final_est_cov = final_est_cov;
    out->mean_cov = geq1; out->median_cov = median_cov; out->lambda = lam; out->max_cov = max_cov;
    out->full_mean_cov = mean_cov; out->lambda_status = status;
    out->contain_count = contain_count; out->n_full = full.size();
    out->passed = (final_est_ani < min_ani) ? 0 : 1;               // :746-764
    return 0;
}

double orc_poisson_cdf(double lambda, uint64_t x) { return poisson_cdf(lambda, x); }

int orc_ratio_lambda(const uint32_t* full_covs, uint64_t n, double min_count_correct, double* out) {
    std::vector<uint32_t> v(full_covs, full_covs + n);
    double lam = 0;
    if (!ratio_lambda(v, min_count_correct, lam)) return 0;
    *out = lam;
    return 1;
}

}  // extern "C"
