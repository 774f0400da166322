// sketch.hip — read-sketch sessions and genome sketches on gfx950.
//
//   push():   K1 seeds (seeds.hip) -> sort survivors by flat position -> K2 annotate (record lookup, boundary /
//             AVX2-tail validation, locality markers) -> append to the session's occurrence arrays in file order
//   finish(): stable radix sort of occurrences by hash -> K3 replay of dup_removal_lsh_full_exact
//             (sketch.rs:690-731) as data-parallel segment kernels -> (k-mer, count) table in ascending k-mer order
//
// Replay formulation.  For one k-mer, let its occurrences in file order be i = 0..n-1, each with
// an optional marker pair (m0_i, m1_i) (pair_kmer_single sketch.rs:625 / pair_kmer :659).  Every occurrence that
// reaches the marker test inserts both of its markers, whether it is then counted or dropped (:709-722), so
//     dropped_i  <=>  has_marker_i  AND  i is not the first processed occurrence (count > 0, :711,:718)
//                     AND ( m0_i == m1_i  OR  {m0_i, m1_i} meets {m0_j, m1_j} for some processed j < i with a marker )
// which needs no sequential state.  The single-end cut-off (`count < 4`, :706,:937) only makes a suffix of the
// occurrences unconditional, handled by one short walk per k-mer.  Mate-2 occurrences whose hash also occurs in
// mate 1 of the same pair are removed first (sketch.rs:852).
#include <algorithm>
#include <unordered_map>

#include "common.h"
#include "device_common.h"
#include "sketch_session.h"

namespace sylph {

bool finish_bucketed(sylph_sketch* sk);   // replay_lds.hip
bool push_short_reads(sylph_sketch* sk, const uint8_t* d_bases, uint32_t phase, const uint64_t* d_off, uint64_t n_records, uint64_t n_bases, int enc);   // reads.hip

void launch_seeds(sylph_ctx* ctx, const uint8_t* d_bases, uint32_t n_bases, uint32_t c, uint32_t k, uint64_t* d_out_hash,
                  uint32_t* d_out_pos, uint32_t out_cap, uint32_t* d_count);
uint32_t seeds_slot_capacity(uint32_t c);
uint32_t seeds_n_tiles(uint64_t n_bases);
void launch_seeds_slots(sylph_ctx* ctx, const uint8_t* d_bases, uint32_t n_bases, uint32_t c, uint32_t k, uint32_t slot_cap,
                        uint64_t* d_slot_hash, uint32_t* d_slot_pos, uint32_t* d_tile_count, SpillState* d_spill,
                        const uint32_t* d_tile_list, uint32_t n_list, uint32_t* d_spill_slot_of_tile);
uint32_t seeds_tile_bases();
void launch_compact_slots(sylph_ctx* ctx, const uint64_t* d_slot_hash, const uint32_t* d_slot_pos, const uint32_t* d_tile_count,
                          const uint32_t* d_tile_off, uint32_t n_tiles, uint32_t slot_cap, const uint64_t* d_spill_hash,
                          const uint32_t* d_spill_pos, const uint32_t* d_spill_slot_of_tile, uint32_t* d_out_pos,
                          uint64_t* d_out_hash);

namespace {


// 32 consecutive bases starting at p (any alignment) -> (even-position 16-mer, odd-position 16-mer), each base a
// 2-bit BYTE_TO_SEQ code, first base most significant (the order pair_kmer[_single] builds them in,
// sketch.rs:636-653,:668-685).  Two unaligned 16-byte loads instead of 32 byte loads.
__device__ __forceinline__ void marker_halves(const uint8_t* __restrict__ p, uint32_t& even, uint32_t& odd) {
    uint32_t e = 0, o = 0;
    uint64_t xs[4];
    __builtin_memcpy(&xs[0], p, 16);
    __builtin_memcpy(&xs[2], p + 16, 16);
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint64_t x = xs[w];
        uint32_t bad = 0;
        uint32_t lo = codes4_fast((uint32_t)x, bad), hi = codes4_fast((uint32_t)(x >> 32), bad);
        if (bad) { lo = codes4_exact((uint32_t)x); hi = codes4_exact((uint32_t)(x >> 32)); }
        // byte lanes b0..b3 of lo, b4..b7 of hi hold bases 8w..8w+7
        const uint32_t ev = ((lo & 3u) << 6) | (((lo >> 16) & 3u) << 4) | ((hi & 3u) << 2) | ((hi >> 16) & 3u);
        const uint32_t od = (((lo >> 8) & 3u) << 6) | (((lo >> 24) & 3u) << 4) | (((hi >> 8) & 3u) << 2) | ((hi >> 24) & 3u);
        e = (e << 8) | ev;
        o = (o << 8) | od;
    }
    even = e;
    odd = o;
}

// Coarse record index of a batch: coarse[t] = record containing flat base t * 2^COARSE_SHIFT (the last record for
// positions at or beyond the end).  One thread per entry, all binary searches in flight at once — so the annotate
// kernel's workgroups need ONE dependent load to bound their record window instead of a 23-step search each
// (that chain of L2/HBM round trips was half of the kernel's time).
constexpr int COARSE_SHIFT = 14;
__global__ __launch_bounds__(256) void coarse_index_kernel(const uint64_t* __restrict__ off, uint64_t n_rec, uint32_t n_entries,
                                                           uint32_t* __restrict__ coarse) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_entries) return;
    const uint64_t total = off[n_rec];
    uint64_t p = (uint64_t)t << COARSE_SHIFT;
    if (total == 0) { coarse[t] = 0; return; }
    if (p >= total) p = total - 1;
    coarse[t] = (uint32_t)find_record(off, n_rec, p);
}

// ---- K2: annotate survivors of one batch (already sorted by flat position) ---------------------------------
// Validates that the k-mer lies inside one record and among the k-mers the reference hashes, finds the record,
// and computes the dedup markers.  Invalid survivors get hash = ~0 (sorts last, dropped in finish()).
// Survivors arrive sorted by position, so a workgroup's 256 survivors span a short run of records: two lanes bound
// it through the coarse index (first and last survivor), the run's offsets are staged in LDS and every lane searches
// only inside that run.
template <bool LIGHT>   // LIGHT: validity only — o_hash, no markers, no OccRec (marker-less single-end batches)
__global__ __launch_bounds__(256) void annotate_reads_kernel(
    const uint8_t* __restrict__ bases, const uint64_t* __restrict__ off, uint64_t n_rec, const uint32_t* __restrict__ pos,
    const uint64_t* __restrict__ hash, uint32_t n, uint32_t pos_bias, uint32_t k, int avx2_compat, int paired,
    int want_markers, uint64_t rec_base, const uint32_t* __restrict__ coarse, uint64_t* __restrict__ o_hash,
    OccRec* __restrict__ o_rec) {
    constexpr uint32_t WIN = 1024;           // record offsets staged in LDS (8 KiB)
    __shared__ uint64_t s_lo, s_hi;
    __shared__ uint64_t s_off[WIN + 2];
    const uint32_t first = blockIdx.x * blockDim.x;
    const uint32_t i = first + threadIdx.x;
    const uint64_t total = off[n_rec];
    if (threadIdx.x == 0 || threadIdx.x == 64) {
        const uint32_t j = threadIdx.x == 0 ? first : min(n, first + blockDim.x) - 1;
        uint64_t p = pos[j] >= pos_bias ? pos[j] - pos_bias : 0;   // positions are relative to the 16 B-aligned load base
        if (total && p >= total) p = total - 1;
        // the record of p lies in [coarse[t], coarse[t + 1]]: bound the window with those instead of searching for it
        const uint64_t t = p >> COARSE_SHIFT;
        if (threadIdx.x == 0) s_lo = total ? coarse[t] : 0; else s_hi = (total ? coarse[t + 1] : 0) + 1;
    }
    __syncthreads();
    // stage off[s_lo .. s_hi + 1] (the run of records this workgroup's survivors fall into, +1 for the mate lookup)
    const uint64_t w_lo = s_lo & ~1ull;      // even, so a pair's three offsets are inside the window too
    const uint64_t w_n = min(s_hi + 2, n_rec + 1) - w_lo;
    const bool in_lds = w_n <= WIN + 2;
    if (in_lds)
        for (uint32_t t = threadIdx.x; t < w_n; t += blockDim.x) s_off[t] = off[w_lo + t];
    __syncthreads();
    const bool live = i < n && pos[i] >= pos_bias;
    const uint64_t p = live ? pos[i] - pos_bias : ~0ull;
    uint64_t h = live ? hash[i] : 0, rid = 0, m0 = 0, m1 = 0;
    bool valid = false;
    if (p < total) {
        // record search + the (up to five) offsets needed, from LDS when the run fits (the usual case) — kept as two
        // separate code paths so that the LDS one compiles to ds_read, not to flat loads through a selected pointer
        uint64_t r, start, L, s1 = 0, s2 = 0, e2 = 0;
        auto locate = [&](auto OFF) {
            uint64_t lo = s_lo, hi = s_hi;   // off[lo] <= p < off[hi]
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (OFF(mid) <= p) lo = mid; else hi = mid;
            }
            r = lo;
            start = OFF(r);
            L = OFF(r + 1) - start;
            if (paired) { const uint64_t r1 = r & ~1ull; s1 = OFF(r1); s2 = OFF(r1 + 1); e2 = OFF(r1 + 2); }
        };
        if (in_lds) locate([&](uint64_t q) -> uint64_t { return s_off[q - w_lo]; });
        else locate([&](uint64_t q) -> uint64_t { return off[q]; });
        valid = (p - start) < n_hashed_kmers(L, k, avx2_compat, 0);
        if (valid) {
            rid = rec_base + r;
            if (!LIGHT && want_markers) {
                const uint8_t* a = nullptr;
                const uint8_t* b = nullptr;
                if (!paired) {
                    // pair_kmer_single, sketch.rs:625-656; caller passes None above 400 bp (sketch.rs:922-927)
                    if (L >= 66 && L <= 400) { a = bases + start; b = a + L / 2; }
                } else {
                    // pair_kmer, sketch.rs:659-688: both mates >= 33 bp
                    if (s2 - s1 >= 33 && e2 - s2 >= 33) { a = bases + s1; b = bases + s2; }
                }
                if (a) {
                    uint32_t f, g, rr, t;
                    marker_halves(a, f, g);    // f: bases 0,2,..,30   g: bases 1,3,..,31
                    marker_halves(b, rr, t);
                    m0 = (uint64_t)f | ((uint64_t)rr << 32);
                    m1 = (uint64_t)g | ((uint64_t)t << 32);
                    rid |= RID_MARKER_BIT;
                    if (paired)
                        rid |= min(emission_rank((uint32_t)(p - start), (uint32_t)n_hashed_kmers(L, k, avx2_compat, 0), avx2_compat), RID_RANK_MAX)
                               << RID_RANK_SHIFT;
                }
            }
        }
    }
    if (i < n) {
        const uint64_t hv = valid ? h : INVALID_HASH;
        o_hash[i] = hv;
        if constexpr (!LIGHT) o_rec[i] = OccRec{hv, rid, m0, m1};
    }
    // (no counter of valid survivors here: one atomic per wavefront on a single word runs at ~88 atomics/us and was
    //  the whole 0.9 ms of this kernel; finish() finds the count by binary search in the hash-sorted array instead)
}

// flag[0] |= 1 when some record of the batch has a length that carries a dedup marker for single-end reads (66 .. 400 bases,
// sketch.rs:627,923)
__global__ __launch_bounds__(256) void marker_lengths_kernel(const uint64_t* __restrict__ off, uint64_t n_rec, uint32_t* __restrict__ flag) {
    bool any = false;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t L = off[r + 1] - off[r];
        any |= L >= 66 && L <= 400;
    }
    if (__ballot(any) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
__global__ __launch_bounds__(256) void plain_records_kernel(const uint64_t* __restrict__ hash, uint64_t n, OccRec* __restrict__ recs) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) recs[i] = OccRec{hash[i], 0ull, 0ull, 0ull};
}

// Genome flavour: (contig, end position, hash), validated with the positions-variant rules
// (avx2_seeding.rs:160: nothing for contigs shorter than 2k).
__global__ __launch_bounds__(256) void annotate_contigs_kernel(const uint64_t* __restrict__ off, uint64_t n_contigs,
                                                               const uint32_t* __restrict__ pos,
                                                               const uint64_t* __restrict__ hash, uint32_t n, uint32_t k,
                                                               int avx2_compat, uint32_t* __restrict__ o_contig,
                                                               uint64_t* __restrict__ o_pos, uint64_t* __restrict__ o_hash) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = pos[i];
    uint64_t h = INVALID_HASH, endpos = 0;
    uint32_t contig = 0;
    if (p < off[n_contigs]) {
        const uint64_t r = find_record(off, n_contigs, p);
        const uint64_t start = off[r], L = off[r + 1] - start;
        if ((p - start) < n_hashed_kmers(L, k, avx2_compat, 1)) {
            h = hash[i];
            contig = (uint32_t)r;
            endpos = p - start + k - 1;   // index of the k-mer's last base (seeding.rs:205)
        }
    }
    o_contig[i] = contig;
    o_pos[i] = endpos;
    o_hash[i] = h;
}

// number of valid occurrences = first index holding INVALID_HASH in the hash-sorted array (one lane, log n loads)
__global__ void count_valid_kernel(const uint64_t* __restrict__ hs, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (hs[mid] < INVALID_HASH) lo = mid + 1; else hi = mid;
    }
    *out = lo;
}

__global__ void iota_kernel(uint32_t* v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// gather payload into hash-sorted order and mark segment heads
__global__ __launch_bounds__(256) void gather_heads_kernel(const uint64_t* __restrict__ hs, const uint32_t* __restrict__ perm,
                                                           const OccRec* __restrict__ recs, uint32_t n,
                                                           uint64_t* __restrict__ rid_s, uint64_t* __restrict__ m0_s,
                                                           uint64_t* __restrict__ m1_s, uint32_t* __restrict__ head,
                                                           uint32_t* __restrict__ headidx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = perm[i];
    const OccRec r = recs[p];
    rid_s[i] = r.rid;
    m0_s[i] = r.m0;
    m1_s[i] = r.m1;
    const bool hd = (i == 0) || (hs[i] != hs[i - 1]);
    head[i] = hd ? 1u : 0u;
    headidx[i] = hd ? i : 0u;
}

// K3a: mate-2 skip rule (sketch.rs:852): a mate-2 occurrence is not processed at all if the same hash was
// produced by mate 1 of the same pair.  Occurrences are in file order inside a segment, so the pair's mate-1
// occurrences sit immediately before its mate-2 occurrences.
__global__ __launch_bounds__(256) void skip_kernel(const uint64_t* __restrict__ rid_s, const uint32_t* __restrict__ seg_start,
                                                   uint32_t n, uint8_t* __restrict__ skip,
                                                   unsigned int* __restrict__ n_skipped) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t rec = i < n ? (rid_s[i] & RID_MASK) : 0;
    uint8_t s = 0;
    if (rec & 1) {
        const uint32_t s0 = seg_start[i];
        for (uint32_t j = i; j > s0;) {
            j--;
            const uint64_t rj = rid_s[j] & RID_MASK;
            if ((rj >> 1) != (rec >> 1)) break;
            if ((rj & 1) == 0) { s = 1; break; }
        }
    }
    if (i < n) skip[i] = s;
    wave_count_add(n_skipped, s != 0);
}

// K3b: dropped_i as defined in the file header, computed by SORTING instead of scanning: an occurrence is dropped iff it is
// not the first processed occurrence of its k-mer and (m0 == m1, or one of its two markers was already inserted by an earlier
// processed occurrence of the same k-mer).  "Already inserted" = the (k-mer segment, marker value) pair has an entry with a
// smaller occurrence index.  Every processed occurrence with a marker contributes two entries e = 2 i + w (w: m0 / m1);
// entries are ordered by (segment, value, e) with two stable radix sorts (by value, then by segment) and every entry whose
// predecessor carries the same (segment, value) marks its occurrence as a hit.  O(n log n) whatever the coverage: the earlier
// formulation scanned all previous occurrences of the k-mer (quadratic for a k-mer with thousands of distinct reads).
__global__ __launch_bounds__(256) void marker_entries_kernel(const uint64_t* __restrict__ rid_s, const uint64_t* __restrict__ m0_s,
                                                             const uint64_t* __restrict__ m1_s, const uint8_t* __restrict__ skip,
                                                             uint32_t n, uint64_t* __restrict__ key, uint32_t* __restrict__ ent) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * n) return;
    const uint32_t i = e >> 1;
    key[e] = (e & 1) ? m1_s[i] : m0_s[i];
    ent[e] = e;
}
// segment key of an entry: the index of the first occurrence of its k-mer, or n (behind every real segment) for entries of
// occurrences that insert no marker (skipped mate 2, sketch.rs:852; records without marker, :627,:661)
__global__ __launch_bounds__(256) void marker_segkey_kernel(const uint32_t* __restrict__ ent, const uint64_t* __restrict__ rid_s,
                                                            const uint32_t* __restrict__ seg_start, const uint8_t* __restrict__ skip,
                                                            uint32_t n, uint32_t* __restrict__ segkey) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= 2 * n) return;
    const uint32_t i = ent[p] >> 1;
    const bool inserts = (rid_s[i] & RID_MARKER_BIT) && !(skip && skip[i]);
    segkey[p] = inserts ? seg_start[i] : n;
}
__global__ __launch_bounds__(256) void marker_hits_kernel(const uint32_t* __restrict__ ent, const uint32_t* __restrict__ segkey,
                                                          const uint64_t* __restrict__ m0_s, const uint64_t* __restrict__ m1_s,
                                                          uint32_t n, uint8_t* __restrict__ hit) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0 || p >= 2 * n) return;
    const uint32_t sk = segkey[p];
    if (sk == n || segkey[p - 1] != sk) return;
    const uint32_t e = ent[p], f = ent[p - 1];
    const uint64_t v = (e & 1) ? m1_s[e >> 1] : m0_s[e >> 1], w = (f & 1) ? m1_s[f >> 1] : m0_s[f >> 1];
    if (v == w) hit[e >> 1] = 1;   // (f >> 1 == e >> 1 only when m0 == m1, which drops the occurrence anyway)
}
// DEDUP_FILTER: hit = the bit a10_mark left in the occurrence's record
__global__ __launch_bounds__(256) void filter_hits_kernel(const uint64_t* __restrict__ rid_s, uint32_t n, uint8_t* __restrict__ hit) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hit[i] = (rid_s[i] & RID_A10_BIT) ? 1 : 0;
}
// flags[i]: bit0 = skip, bit1 = would-be-dropped.  Eproc = exclusive count of processed (non-skipped) occurrences.
__global__ __launch_bounds__(256) void dup_flags_kernel(const uint64_t* __restrict__ rid_s, const uint64_t* __restrict__ m0_s,
                                                        const uint64_t* __restrict__ m1_s,
                                                        const uint32_t* __restrict__ seg_start,
                                                        const uint8_t* __restrict__ skip, const uint32_t* __restrict__ Eproc,
                                                        const uint8_t* __restrict__ hit, uint32_t n, int filter,
                                                        uint8_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t fl = skip ? skip[i] : 0;
    if (!fl && (rid_s[i] & RID_MARKER_BIT)) {
        const uint32_t s0 = seg_start[i];
        bool any_prev = skip ? (Eproc[i] != Eproc[s0]) : (i != s0);
        if (filter && hit[i]) {
            // `*c > 0` (sketch.rs:749, :756) in the order of the WALK: a record's seeds go through the dedup in emission order (rank bits
            // of the rid), the sorted segment lists them by position — the first of the walk is the lowest rank among the segment's
            // leading occurrences of the head's record (replay_lds.hip walk_first)
            const uint64_t rec0 = rid_s[s0] & RID_MASK, ri = rid_s[i];
            any_prev = true;
            if ((ri & RID_MASK) == rec0) {
                any_prev = false;
                const uint64_t rank_i = (ri >> RID_RANK_SHIFT) & RID_RANK_MAX;
                for (uint32_t q = s0; q < n && seg_start[q] == s0 && (rid_s[q] & RID_MASK) == rec0; q++)
                    if (q != i && ((rid_s[q] >> RID_RANK_SHIFT) & RID_RANK_MAX) < rank_i) { any_prev = true; break; }
            }
        }
        // (filter mode: equal markers are the filter's business too — its second test finds what the first inserted)
        if (any_prev && (hit[i] || (!filter && m0_s[i] == m1_s[i]))) fl |= 2;
    }
    flags[i] = fl;
}
__global__ __launch_bounds__(256) void processed_kernel(const uint8_t* __restrict__ skip, uint32_t n, uint32_t* __restrict__ u) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) u[i] = (i < n && !skip[i]) ? 1u : 0u;
}

// K3c.  With u_i = 1 if occurrence i would be counted were there no cut-off (processed and not dropped) and
// P_i = number of such occurrences before i inside its k-mer (a segmented exclusive prefix sum), the reference's
// walk (sketch.rs:701-730) reduces to:  counted_i = (cutoff && P_i >= cutoff) ? 1 : u_i   (once the count has
// reached MAX_DEDUP_COUNT nothing is ever dropped again, :706), removed_i = processed_i && !counted_i.
__global__ __launch_bounds__(256) void would_count_kernel(const uint8_t* __restrict__ flags, uint32_t n, int no_dedup,
                                                          uint32_t* __restrict__ u) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { u[i] = 0; return; }   // sentinel so the exclusive scan has n+1 entries
    const uint8_t fl = flags[i];
    u[i] = (fl & 1) ? 0u : ((no_dedup || !(fl & 2)) ? 1u : 0u);
}

__global__ __launch_bounds__(256) void counted_kernel(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ Eu,
                                                      const uint32_t* __restrict__ seg_start, uint32_t n, int no_dedup,
                                                      uint32_t cutoff, uint32_t* __restrict__ counted) {
    // removed occurrences = processed - counted, taken from the scan total on the host side (no atomics here)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint8_t fl = flags[i];
        uint32_t c = 0;
        if (!(fl & 1)) {
            const uint32_t P = Eu[i] - Eu[seg_start[i]];
            const bool u = no_dedup || !(fl & 2);
            c = (cutoff && P >= cutoff) ? 1u : (u ? 1u : 0u);
        }
        counted[i] = c;
    } else if (i == n) {
        counted[i] = 0;
    }
}

// start[s] = index of the first occurrence of k-mer s (start[n_seg] = n)
__global__ __launch_bounds__(256) void seg_starts_kernel(const uint32_t* __restrict__ head, const uint32_t* __restrict__ seg_id,
                                                         uint32_t n, uint32_t n_seg, uint32_t* __restrict__ start) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && head[i]) start[seg_id[i]] = i;
    if (i == 0) start[n_seg] = n;
}

__global__ __launch_bounds__(256) void emit_table_kernel(const uint64_t* __restrict__ hs, const uint32_t* __restrict__ start,
                                                         const uint32_t* __restrict__ Ec, uint32_t n_seg,
                                                         uint64_t* __restrict__ out_k, uint32_t* __restrict__ out_c) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t a = start[s], b = start[s + 1];
    out_k[s] = hs[a];
    out_c[s] = Ec[b] - Ec[a];
}

}  // namespace
}  // namespace sylph

using namespace sylph;

namespace sylph {

static uint32_t grid_for(uint64_t n, uint32_t tpb = 256) { return (uint32_t)((n + tpb - 1) / tpb); }

// Runs K1 with capacity retry; returns survivors sorted by flat position in (ctx->scratch[2] = pos, [3] = hash).
uint32_t seeds_sorted_by_pos(sylph_ctx* ctx, const uint8_t* d_bases, uint64_t n_bases, uint32_t c, uint32_t k,
                                    uint32_t* d_count) {
    SY_REQUIRE(n_bases < (1ull << 32), "a batch may hold at most 2^32-1 bases (got %llu)", (unsigned long long)n_bases);
    if (n_bases == 0) return 0;
    if (ctx->seeds_mode != 1) {
        // ordered K1: per-tile slots in position order, then scan + gather (no radix sort, no atomics)
        HostPhase ph(ctx, "push: seeds (ordered slots)");
        const uint32_t n_tiles = seeds_n_tiles(n_bases), slot_cap = seeds_slot_capacity(c);
        DevBuf &b_sh = ctx->scratch[0], &b_sp = ctx->scratch[1], &b_tc = ctx->scratch[4];
        b_sh.reserve((size_t)n_tiles * slot_cap * 8);
        b_sp.reserve((size_t)n_tiles * slot_cap * 4);
        // [tile_count (n_tiles+1) | tile_off (n_tiles+1) | spill_slot_of_tile (n_tiles) | SpillState]
        b_tc.reserve(((size_t)n_tiles + 1) * 4 * 3 + sizeof(SpillState) + 16);
        uint32_t* tile_count = b_tc.as<uint32_t>();
        uint32_t* tile_off = tile_count + (n_tiles + 1);
        uint32_t* spill_slot = tile_off + (n_tiles + 1);
        SpillState* d_spill = reinterpret_cast<SpillState*>(spill_slot + n_tiles + 1);
        SY_HIP(hipMemsetAsync(tile_count + n_tiles, 0, 4, ctx->stream));   // scan sentinel
        SY_HIP(hipMemsetAsync(d_spill, 0, 4, ctx->stream));                // n_tiles of the spill list
        launch_seeds_slots(ctx, d_bases, (uint32_t)n_bases, c, k, slot_cap, b_sh.as<uint64_t>(), b_sp.as<uint32_t>(), tile_count,
                           d_spill, nullptr, 0, nullptr);
        exclusive_sum_u32(ctx, tile_count, tile_off, (size_t)n_tiles + 1);
        uint32_t res[2] = {0, 0};   // total survivors, tiles that overflowed their slots
        SY_HIP(hipMemcpyAsync(ctx->pinned, tile_off + n_tiles, 4, hipMemcpyDeviceToHost, ctx->stream));
        SY_HIP(hipMemcpyAsync((uint8_t*)ctx->pinned + 4, d_spill, 4, hipMemcpyDeviceToHost, ctx->stream));
        SY_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(res, ctx->pinned, 8);
        if (!ctx->pending.empty()) profile_collect(ctx);
        if (res[1] <= SPILL_MAX_TILES) {
            if (res[0] == 0) return 0;
            const uint64_t* sp_h = nullptr;
            const uint32_t* sp_p = nullptr;
            if (res[1]) {   // a few tiles (low-complexity reads, repeats) are redone with room for every position
                DevBuf& b_x = ctx->scratch[7];   // [spill hashes | spill positions]
                const size_t tb = seeds_tile_bases();
                b_x.reserve((size_t)res[1] * tb * 12);
                uint64_t* xh = b_x.as<uint64_t>();
                uint32_t* xp = reinterpret_cast<uint32_t*>(xh + (size_t)res[1] * tb);
                ScopedKernelTimer ts(ctx, "seeds_spill");   // (family of its own so that tests can see this path was taken)
                launch_seeds_slots(ctx, d_bases, (uint32_t)n_bases, c, k, slot_cap, xh, xp, tile_count, d_spill, d_spill->tiles, res[1],
                                   spill_slot);
                sp_h = xh;
                sp_p = xp;
            }
            ctx->scratch[2].reserve((size_t)res[0] * 4);
            ctx->scratch[3].reserve((size_t)res[0] * 8);
            launch_compact_slots(ctx, b_sh.as<uint64_t>(), b_sp.as<uint32_t>(), tile_count, tile_off, n_tiles, slot_cap, sp_h, sp_p,
                                 spill_slot, ctx->scratch[2].as<uint32_t>(), ctx->scratch[3].as<uint64_t>());
            return res[0];
        }
        // more overflowing tiles than spill regions (a batch dominated by repeats): redo it with the unordered kernel
    }
    uint64_t cap = n_bases / c + n_bases / (4ull * c) + 65536;
    if (cap > n_bases) cap = n_bases;
    uint32_t n = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        HostPhase ph(ctx, "push: seeds kernel");
        ctx->scratch[0].reserve(cap * 8);   // hash (unsorted)
        ctx->scratch[1].reserve(cap * 4);   // pos (unsorted)
        static const bool slowlog = getenv("SYLPH_HIP_SLOWLOG") != nullptr;
        const double t0 = slowlog ? HostPhase::now() : 0;
        SY_HIP(hipMemsetAsync(d_count, 0, 4, ctx->stream));
        const double t1 = slowlog ? HostPhase::now() : 0;
        if (slowlog && t1 - t0 > 3.0) fprintf(stderr, "[sylph_hip] slow memset issue %.3f ms\n", t1 - t0);
        launch_seeds(ctx, d_bases, (uint32_t)n_bases, c, k, ctx->scratch[0].as<uint64_t>(), ctx->scratch[1].as<uint32_t>(),
                     (uint32_t)cap, d_count);
        ctx->read_back(&n, d_count, 4);
        if (n <= cap) break;
        cap = n;                            // the counter kept counting: exact size for the retry
        SY_REQUIRE(attempt == 0, "seed buffer overflow persisted");
    }
    if (n == 0) return 0;
    HostPhase ph(ctx, "push: sort by position");
    ctx->scratch[2].reserve((size_t)n * 4);
    ctx->scratch[3].reserve((size_t)n * 8);
    sort_pairs_u32_u64(ctx, ctx->scratch[1].as<uint32_t>(), ctx->scratch[2].as<uint32_t>(), ctx->scratch[0].as<uint64_t>(),
                       ctx->scratch[3].as<uint64_t>(), n, 0, std::max(1, bit_length(n_bases)));
    return n;
}

// packed (SYLPH_ENC_2BIT) stream -> ASCII, for the batches the short-read kernel declines (long records, repeats everywhere)
__global__ __launch_bounds__(256) void unpack_2bit_kernel(const uint8_t* __restrict__ packed, uint32_t phase, uint64_t n_bases,
                                                          uint8_t* __restrict__ ascii) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bases; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = i + phase;
        const uint32_t code = (packed[b >> 2] >> (6u - 2u * (uint32_t)(b & 3))) & 3u;
        ascii[i] = (uint8_t)((0x54474341u >> (8u * code)) & 0xFFu);   // 'A','C','G','T'
    }
}
// off_out[i] = off_in[i] - off_in[0]: a chunk of a larger batch becomes a batch of its own
__global__ __launch_bounds__(256) void rebase_offsets_kernel(const uint64_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] - in[0];
}

// One device-resident batch: records d_off[0..n_records] over the stream at d_bases (ASCII, or packed 2-bit starting
// `phase` bases into the first byte).  Appends the batch's occurrences to the session.
void materialise_plain_records(sylph_sketch* sk) {
    if (sk->n_plain == 0) return;
    sylph_ctx* ctx = sk->ctx;
    sk->recs.grow_keep(sk->n_occ * sizeof(OccRec), 0, ctx->stream);
    hipLaunchKernelGGL(plain_records_kernel, dim3(grid_for(sk->n_plain)), dim3(256), 0, ctx->stream, sk->hash.as<uint64_t>(), sk->n_plain,
                       sk->recs.as<OccRec>());
    SY_HIP(hipGetLastError());
    sk->n_plain = 0;
}

static void process_batch(sylph_sketch* sk, const uint8_t* d_bases, uint32_t phase, const uint64_t* d_off, uint64_t n_records, uint64_t n_bases,
                          int enc) {
    sylph_ctx* ctx = sk->ctx;
    flush_pending_slots(sk);   // a further batch: the previous one's occurrences (still in their slots) go to the dense arrays first
    // short-read batches (mean record length <= 300): one lane per record, seeding + markers fused (reads.hip); it declines
    // (returns false) when some record is longer than its halo, and the position kernel + annotate below take over
    bool done = false;
    const bool short_batch = ctx->seeds_mode == 0 && n_bases < (1ull << 32) - 64 && n_records < (1ull << 31) && n_bases <= 300ull * n_records;
    // Marker-less batch?  Single-end only (a pair's mate-2 rule needs the record ids), everything so far marker-less too, and
    // either --no-dedup or no record of a length that carries a marker (asked of the offsets by a small kernel whose answer
    // comes back with the seeding kernel's count).
    bool plain = !short_batch && !sk->paired && sk->n_plain == sk->n_occ && ctx->plain_records;
    uint32_t* d_marked = sk->counters.as<uint32_t>() + 6;
    if (plain && !sk->no_dedup) {
        SY_HIP(hipMemsetAsync(d_marked, 0, 4, ctx->stream));
        if (n_records) {
            hipLaunchKernelGGL(marker_lengths_kernel, dim3((uint32_t)std::min<uint64_t>(1024, (n_records + 255) / 256)), dim3(256), 0, ctx->stream,
                               d_off, n_records, d_marked);
            SY_HIP(hipGetLastError());   // a launch that failed would leave d_marked at 0: the batch would pass for marker-less
        }
    }
    if (!plain) materialise_plain_records(sk);
    if (short_batch) done = push_short_reads(sk, d_bases, phase, d_off, n_records, n_bases, enc);
    if (!done && enc == SYLPH_ENC_2BIT) {   // the position kernel and the marker loads of annotate read ASCII
        sk->batch_ascii.reserve(n_bases + 64);
        if (n_bases)
            hipLaunchKernelGGL(unpack_2bit_kernel, dim3((uint32_t)std::min<uint64_t>((n_bases + 255) / 256, 65536)), dim3(256), 0, ctx->stream,
                               d_bases, phase, n_bases, sk->batch_ascii.as<uint8_t>());
        SY_HIP(hipMemsetAsync(sk->batch_ascii.as<uint8_t>() + n_bases, 0, 64, ctx->stream));
        d_bases = sk->batch_ascii.as<uint8_t>();
    }
    uint32_t* d_count = sk->counters.as<uint32_t>();
    // K1 loads 16 B per lane: start it at the aligned address below d_bases and subtract the bias afterwards (a device
    // pointer into the middle of a larger buffer, e.g. the second batch of a sample, need not be aligned)
    const uint32_t bias = (uint32_t)((uintptr_t)d_bases & 15);
    const uint32_t n = done ? 0 : seeds_sorted_by_pos(ctx, d_bases - bias, n_bases + bias, sk->c, sk->k, d_count);
    if (plain && !sk->no_dedup && !done) {
        uint32_t marked = 0;
        ctx->read_back(&marked, d_marked, 4);        // (the stream is idle: seeds_sorted_by_pos has read its count back)
        if (marked) { plain = false; materialise_plain_records(sk); }
    }
    if (n) {
        HostPhase ph(ctx, "push: grow + annotate");
        const uint64_t need = sk->n_occ + n;
        const size_t keep = sk->n_occ * 8;
        sk->hash.grow_keep(need * 8, keep, ctx->stream);
        if (!plain) sk->recs.grow_keep(need * sizeof(OccRec), sk->n_occ * sizeof(OccRec), ctx->stream);
        const uint32_t n_coarse = (uint32_t)(n_bases >> COARSE_SHIFT) + 2;
        ctx->scratch[5].reserve((size_t)n_coarse * 4);
        ScopedKernelTimer t(ctx, "annotate");
        hipLaunchKernelGGL(coarse_index_kernel, dim3(grid_for(n_coarse)), dim3(256), 0, ctx->stream, d_off, n_records, n_coarse,
                           ctx->scratch[5].as<uint32_t>());
        if (plain)
            hipLaunchKernelGGL(annotate_reads_kernel<true>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_bases, d_off, n_records,
                               ctx->scratch[2].as<uint32_t>(), ctx->scratch[3].as<uint64_t>(), n, bias, sk->k, sk->avx2_compat,
                               sk->paired, 0, sk->rec_base, ctx->scratch[5].as<uint32_t>(), sk->hash.as<uint64_t>() + sk->n_occ,
                               (OccRec*)nullptr);
        else
            hipLaunchKernelGGL(annotate_reads_kernel<false>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_bases, d_off, n_records,
                               ctx->scratch[2].as<uint32_t>(), ctx->scratch[3].as<uint64_t>(), n, bias, sk->k, sk->avx2_compat,
                               sk->paired, sk->no_dedup ? 0 : 1, sk->rec_base, ctx->scratch[5].as<uint32_t>(),
                               sk->hash.as<uint64_t>() + sk->n_occ, sk->recs.as<OccRec>() + sk->n_occ);
        SY_HIP(hipGetLastError());
        if (plain) sk->n_plain = need;
        sk->n_occ = need;
    }
    sk->rec_base += n_records;
    SY_REQUIRE(sk->rec_base <= RID_MASK, "more than 2^42 records in one sample");
}

// A deferred batch whose verdict came back bad (a record too long for the short-read kernel, a block that overflowed its slots):
// the session goes back to where it was before that push and the batch — its memory is still the caller's, that was the deal —
// runs through the checked push (which falls back to the position kernel / spill regions as needed).
void redo_deferred_batch(sylph_sketch* sk) {
    const PendingSlots d = sk->pend;
    sk->pend = PendingSlots{};
    sk->a10_state = 0;                          // whatever the filter pass marked in the slots is gone with them (a10.hip)
    sk->rec_base -= d.n_records;
    if (sk->ctx->profile) sk->ctx->stats["deferred_redo"].launches++;
    const bool borrow = sk->borrow_until_finish;
    sk->borrow_until_finish = false;
    try { process_batch(sk, d.bases, d.phase, d.off, d.n_records, d.n_bases, d.enc); }
    catch (...) { sk->borrow_until_finish = borrow; throw; }
    sk->borrow_until_finish = borrow;
}

// Host batches (SYLPH_MEM_HOST / SYLPH_MEM_HOST_PINNED) are cut into chunks of whole records (whole pairs) that travel on a
// COPY stream into two device slots while the compute stream works on the previous chunk: the PCIe transfer (1 B per base,
// or 1/4 B packed) is the long pole of a host-fed sample — 18 ms per Gbp against ~1 ms of kernels — and everything else
// hides under it.  Page-locked caller memory is DMA'd in place; pageable memory goes through the ctx's pinned staging pair.
static void push_host_batch(sylph_sketch* sk, const uint8_t* bases, const uint64_t* rec_off, uint64_t n_records, int mem, int enc) {
    sylph_ctx* ctx = sk->ctx;
    const bool pinned = mem == SYLPH_MEM_HOST_PINNED;
    const uint64_t bpb = enc == SYLPH_ENC_2BIT ? 4 : 1;                         // bases per byte
    if (!ctx->copy_stream) {
        SY_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) SY_HIP(hipEventCreateWithFlags(&ctx->copy_ev[i], hipEventDisableTiming));
    }
    if (!pinned) {   // make sure the staging pair exists (ctx->h2d creates it lazily)
        uint8_t dummy = 0;
        ctx->counters.reserve(64);
        ctx->h2d(ctx->counters.as<uint8_t>() + 32, &dummy, 1);
    }
    // chunk limits: page-locked input is only bounded by the device slots; pageable input must fit a 32 MiB staging buffer
    // (bases + offsets)
    const uint64_t max_bytes = std::min<uint64_t>(ctx->push_chunk_bytes, pinned ? (1ull << 31) : (24ull << 20));
    const uint64_t max_recs = pinned ? (4ull << 20) : ((sylph_ctx::STAGE_BYTES - (24ull << 20)) / 8 - 2);
    struct Chunk { uint64_t r0, r1, b0, b1; };
    auto next_chunk = [&](uint64_t r0) {
        const uint64_t b0 = rec_off[r0];
        uint64_t hi = std::min(n_records, r0 + max_recs);
        const uint64_t limit = b0 + std::min<uint64_t>(max_bytes * bpb, 1ull << 31) - 8;
        if (rec_off[hi] > limit) hi = (uint64_t)(std::upper_bound(rec_off + r0, rec_off + hi + 1, limit) - rec_off) - 1;
        if (sk->paired) hi -= (hi - r0) & 1;
        if (hi <= r0) hi = std::min(n_records, r0 + (sk->paired ? 2 : 1));     // a single record longer than a chunk
        return Chunk{r0, hi, b0, rec_off[hi]};
    };
    auto enqueue_copy = [&](const Chunk& c, int slot) {
        // device slot: [bases (from the byte that holds base b0) | pad] and [raw offsets r0..r1]
        const uint64_t byte0 = c.b0 / bpb, byte1 = (c.b1 + bpb - 1) / bpb, nb = byte1 - byte0, no = (c.r1 - c.r0 + 1) * 8;
        sk->slot_bases[slot].reserve(nb + 64);
        sk->slot_off[slot].reserve(no * 2);
        const uint8_t* src_b = bases + byte0;
        const uint64_t* src_o = rec_off + c.r0;
        if (!pinned && nb + no + 8 > sylph_ctx::STAGE_BYTES) {   // one record longer than a staging buffer: the plain staged copy
            SY_HIP(hipStreamSynchronize(ctx->copy_stream));       // (it reuses the staging pair: nothing of ours may still be in flight)
            ctx->h2d(sk->slot_bases[slot].p, src_b, nb);
            SY_HIP(hipMemsetAsync(sk->slot_bases[slot].as<uint8_t>() + nb, 0, 64, ctx->stream));
            ctx->h2d(sk->slot_off[slot].p, src_o, no);
            SY_HIP(hipEventRecord(ctx->copy_ev[slot], ctx->stream));
            return;
        }
        if (!pinned) {
            SY_HIP(hipEventSynchronize(ctx->stage_ev[slot]));                   // the previous copy out of this staging buffer finished
            uint8_t* stg = (uint8_t*)ctx->stage[slot];
            if (nb) memcpy(stg, src_b, nb);
            memcpy(stg + ((nb + 7) & ~7ull), src_o, no);
            src_b = stg;
            src_o = reinterpret_cast<const uint64_t*>(stg + ((nb + 7) & ~7ull));
        }
        if (nb) SY_HIP(hipMemcpyAsync(sk->slot_bases[slot].p, src_b, nb, hipMemcpyHostToDevice, ctx->copy_stream));
        SY_HIP(hipMemsetAsync(sk->slot_bases[slot].as<uint8_t>() + nb, 0, 64, ctx->copy_stream));
        SY_HIP(hipMemcpyAsync(sk->slot_off[slot].p, src_o, no, hipMemcpyHostToDevice, ctx->copy_stream));
        SY_HIP(hipEventRecord(ctx->copy_ev[slot], ctx->copy_stream));
        if (!pinned) SY_HIP(hipEventRecord(ctx->stage_ev[slot], ctx->copy_stream));
    };
    // the device slots may still be read by kernels of the previous push (spill redo, annotate): order the copy stream behind them
    SY_HIP(hipStreamSynchronize(ctx->stream));
    Chunk cur = next_chunk(0);
    enqueue_copy(cur, 0);
    for (int j = 0;; j++) {
        const int slot = j & 1;
        Chunk nxt{0, 0, 0, 0};
        const bool more = cur.r1 < n_records;
        if (more) { nxt = next_chunk(cur.r1); enqueue_copy(nxt, slot ^ 1); }    // travels while this chunk is processed
        SY_HIP(hipStreamWaitEvent(ctx->stream, ctx->copy_ev[slot], 0));
        const uint64_t n_rec = cur.r1 - cur.r0;
        uint64_t* d_raw = sk->slot_off[slot].as<uint64_t>();
        uint64_t* d_off = d_raw + (n_rec + 1);
        hipLaunchKernelGGL(rebase_offsets_kernel, dim3(grid_for(n_rec + 1)), dim3(256), 0, ctx->stream, d_raw, n_rec + 1, d_off);
        {   // (the device slots are overwritten by the copy after next: nothing here may be deferred to finish)
            const bool borrow = sk->borrow_until_finish;
            sk->borrow_until_finish = false;
            try { process_batch(sk, sk->slot_bases[slot].as<uint8_t>(), (uint32_t)(cur.b0 % bpb), d_off, n_rec, cur.b1 - cur.b0, enc); }
            catch (...) { sk->borrow_until_finish = borrow; throw; }
            sk->borrow_until_finish = borrow;
        }
        // every kernel that reads this slot must be done before the copy after next overwrites it
        SY_HIP(hipStreamSynchronize(ctx->stream));
        if (!more) break;
        cur = nxt;
    }
}

static void sketch_push_impl(sylph_sketch* sk, const uint8_t* bases, const uint64_t* rec_off, uint64_t n_records, int mem,
                             const uint64_t* n_bases_hint = nullptr, int enc = SYLPH_ENC_ASCII) {
    sylph_ctx* ctx = sk->ctx;
    SY_REQUIRE(!sk->finished, "sylph_sketch_push after finish");
    SY_REQUIRE(mem == SYLPH_MEM_HOST || mem == SYLPH_MEM_DEVICE || mem == SYLPH_MEM_HOST_PINNED, "bad mem kind %d", mem);
    SY_REQUIRE(enc == SYLPH_ENC_ASCII || enc == SYLPH_ENC_2BIT, "bad encoding %d", enc);
    SY_REQUIRE(!sk->paired || (n_records % 2 == 0), "paired batches must hold an even number of records");
    if (n_records == 0) return;
    SY_REQUIRE(rec_off, "null rec_off");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard dg(ctx->device);
    HostPhase ph_total(ctx, "push: total");
    if (mem != SYLPH_MEM_DEVICE) {
        SY_REQUIRE(rec_off[0] == 0, "rec_off[0] must be 0");
        SY_REQUIRE(bases || rec_off[n_records] == 0, "null bases");
        SY_REQUIRE(!n_bases_hint || *n_bases_hint == rec_off[n_records], "n_bases does not match rec_off[n_records]");
        push_host_batch(sk, bases, rec_off, n_records, mem, enc);
        return;
    }
    uint64_t n_bases;
    if (n_bases_hint) n_bases = *n_bases_hint;
    else ctx->read_back(&n_bases, rec_off + n_records, 8);
    SY_REQUIRE(bases || n_bases == 0, "null bases");
    SY_REQUIRE(enc == SYLPH_ENC_ASCII || ((uintptr_t)bases & 3) == 0, "a packed device stream must be 4-byte aligned");
    process_batch(sk, bases, 0, rec_off, n_records, n_bases, enc);
}

// Device-wide replay of dup_removal_lsh_full_exact over `n_all` occurrences (hash = sort key, INVALID_HASH entries ignored;
// recs in file order): stable radix sort by hash, then per-occurrence kernels and scans (see the file header).  Writes the
// (k-mer, count) table in ascending k-mer order.  Used for whole samples (finish = generic, c < 2) and, by the bucket path,
// for the occurrences of buckets that do not fit in LDS.
void generic_replay(sylph_ctx* ctx, const uint64_t* d_hash, const OccRec* d_recs, uint32_t n_all, bool paired, int dedup,
                    DevBuf& out_k, DevBuf& out_c, uint64_t& n_out, uint64_t& removed_out) {
    const bool no_dedup = dedup == DEDUP_NONE, filter = dedup == DEDUP_FILTER;
    uint32_t nv = 0;
    n_out = 0;
    removed_out = 0;
    // (buffers come from the ctx pool, not from ctx->scratch: the bucket path calls this for the occurrences of its
    //  overflowing buckets while its own scratch arrays are still live)
    DevBuf b_idx(ctx), b_hs(ctx), b_perm(ctx), b_cnt(ctx);
    b_cnt.reserve(64);
    if (n_all) {
        // stable sort by hash; invalid occurrences carry ~0 and end up behind the nv valid ones
        HostPhase ph(ctx, "finish: sort by hash");
        b_idx.reserve((size_t)n_all * 4);
        b_hs.reserve((size_t)n_all * 8);
        b_perm.reserve((size_t)n_all * 4);
        hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n_all)), dim3(256), 0, ctx->stream, b_idx.as<uint32_t>(), n_all);
        sort_pairs_u64_u32(ctx, d_hash, b_hs.as<uint64_t>(), b_idx.as<uint32_t>(), b_perm.as<uint32_t>(),
                           n_all, 0, 64);
        uint32_t* d_nv = b_cnt.as<uint32_t>();
        hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(1), 0, ctx->stream, b_hs.as<uint64_t>(), n_all, d_nv);
        ctx->read_back(&nv, d_nv, 4);
    }
    if (nv) {
        HostPhase ph_replay(ctx, "finish: replay");
        DevBuf b_rid(ctx), b_m0(ctx), b_m1(ctx), b_u32(ctx), b_fl(ctx);
        const size_t nv1 = (size_t)nv + 1;
        b_rid.reserve((size_t)nv * 8);
        b_m0.reserve((size_t)nv * 8);
        b_m1.reserve((size_t)nv * 8);
        b_u32.reserve(nv1 * 4 * 6);          // head | headidx->Eu | seg_start | seg_id | u->counted | Ec/start
        b_fl.reserve((size_t)nv * 2);        // skip | flags
        uint32_t* head = b_u32.as<uint32_t>();
        uint32_t* headidx = head + nv1;      // reused as Eu (exclusive sum of u) once seg_start exists
        uint32_t* seg_start = headidx + nv1;
        uint32_t* seg_id = seg_start + nv1;
        uint32_t* uc = seg_id + nv1;         // u, then counted
        uint32_t* Ec = uc + nv1;             // exclusive sum of counted
        uint8_t* skip = b_fl.as<uint8_t>();
        uint8_t* flags = skip + nv;
        const uint64_t* hs = b_hs.as<uint64_t>();
        {
            ScopedKernelTimer t(ctx, "replay");
            hipLaunchKernelGGL(gather_heads_kernel, dim3(grid_for(nv)), dim3(256), 0, ctx->stream, hs, b_perm.as<uint32_t>(),
                               d_recs, nv,
                               b_rid.as<uint64_t>(), b_m0.as<uint64_t>(), b_m1.as<uint64_t>(), head, headidx);
        }
        inclusive_max_u32(ctx, headidx, seg_start, nv);
        exclusive_sum_u32(ctx, head, seg_id, nv);
        unsigned int* d_skipped = b_cnt.as<unsigned int>() + 4;
        SY_HIP(hipMemsetAsync(d_skipped, 0, 4, ctx->stream));
        uint32_t* Eu = headidx;
        {
            ScopedKernelTimer t(ctx, "replay");
            const uint8_t* skip_arg = nullptr;
            if (paired) {
                hipLaunchKernelGGL(skip_kernel, dim3(grid_for(nv)), dim3(256), 0, ctx->stream, b_rid.as<uint64_t>(),
                                   seg_start, nv, skip, d_skipped);
                skip_arg = skip;
            }
            if (!no_dedup || paired) {
                // (with --no-dedup only the skip flags matter, :852 applies regardless; the marker flags are ignored by
                //  would_count_kernel, but computing them keeps one code path)
                DevBuf b_key(ctx), b_key2(ctx), b_ent(ctx), b_hit(ctx);
                const size_t ne = (size_t)nv * 2;
                b_key.reserve(ne * 8);
                b_key2.reserve(ne * 8);
                b_ent.reserve(ne * 4 * 4);       // ent_in | ent_mid -> ent_out | segkey_in | segkey_out
                b_hit.reserve((size_t)nv + 4);
                uint32_t* ent_in = b_ent.as<uint32_t>();
                uint32_t* ent_mid = ent_in + ne;
                uint32_t* sk_in = ent_mid + ne;
                uint32_t* sk_out = sk_in + ne;
                uint32_t* ent_out = ent_in;      // ent_in is dead after the first sort
                SY_HIP(hipMemsetAsync(b_hit.p, 0, nv, ctx->stream));
                if (filter)        // sketch.rs:733-769: the marker test is the filter's answer (a10.hip), no sorting of marker entries
                    hipLaunchKernelGGL(filter_hits_kernel, dim3(grid_for(nv)), dim3(256), 0, ctx->stream, b_rid.as<uint64_t>(), nv, b_hit.as<uint8_t>());
                else {
                hipLaunchKernelGGL(marker_entries_kernel, dim3(grid_for(ne)), dim3(256), 0, ctx->stream, b_rid.as<uint64_t>(),
                                   b_m0.as<uint64_t>(), b_m1.as<uint64_t>(), skip_arg, nv, b_key.as<uint64_t>(), ent_in);
                sort_pairs_u64_u32(ctx, b_key.as<uint64_t>(), b_key2.as<uint64_t>(), ent_in, ent_mid, ne, 0, 64);
                hipLaunchKernelGGL(marker_segkey_kernel, dim3(grid_for(ne)), dim3(256), 0, ctx->stream, ent_mid, b_rid.as<uint64_t>(),
                                   seg_start, skip_arg, nv, sk_in);
                sort_pairs_u32_u32(ctx, sk_in, sk_out, ent_mid, ent_out, ne, 0, std::max(1, bit_length(nv)));
                hipLaunchKernelGGL(marker_hits_kernel, dim3(grid_for(ne)), dim3(256), 0, ctx->stream, ent_out, sk_out,
                                   b_m0.as<uint64_t>(), b_m1.as<uint64_t>(), nv, b_hit.as<uint8_t>());
                }
                const uint32_t* Eproc = nullptr;
                if (skip_arg) {   // exclusive count of processed occurrences (uc / Ec are free until would_count_kernel)
                    hipLaunchKernelGGL(processed_kernel, dim3(grid_for(nv1)), dim3(256), 0, ctx->stream, skip_arg, nv, uc);
                    exclusive_sum_u32(ctx, uc, Ec, nv1);
                    Eproc = Ec;
                }
                hipLaunchKernelGGL(dup_flags_kernel, dim3(grid_for(nv)), dim3(256), 0, ctx->stream, b_rid.as<uint64_t>(),
                                   b_m0.as<uint64_t>(), b_m1.as<uint64_t>(), seg_start, skip_arg, Eproc, b_hit.as<uint8_t>(), nv, filter ? 1 : 0, flags);
            } else
                SY_HIP(hipMemsetAsync(flags, 0, nv, ctx->stream));
            hipLaunchKernelGGL(would_count_kernel, dim3(grid_for(nv1)), dim3(256), 0, ctx->stream, flags, nv, no_dedup ? 1 : 0, uc);
        }
        exclusive_sum_u32(ctx, uc, Eu, nv1);
        {
            ScopedKernelTimer t(ctx, "replay");
            hipLaunchKernelGGL(counted_kernel, dim3(grid_for(nv1)), dim3(256), 0, ctx->stream, flags, Eu, seg_start, nv,
                               no_dedup ? 1 : 0, paired ? 0u : 4u /* MAX_DEDUP_COUNT, constants.rs:14 */, uc);
        }
        exclusive_sum_u32(ctx, uc, Ec, nv1);
        uint32_t tail[4] = {0, 0, 0, 0};   // seg_id, head of the last occurrence; total counted; skipped
        SY_HIP(hipMemcpyAsync(ctx->pinned, seg_id + (nv - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
        SY_HIP(hipMemcpyAsync((uint8_t*)ctx->pinned + 4, head + (nv - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
        SY_HIP(hipMemcpyAsync((uint8_t*)ctx->pinned + 8, Ec + nv, 4, hipMemcpyDeviceToHost, ctx->stream));
        SY_HIP(hipMemcpyAsync((uint8_t*)ctx->pinned + 12, d_skipped, 4, hipMemcpyDeviceToHost, ctx->stream));
        SY_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(tail, ctx->pinned, 16);
        const uint32_t n_seg = tail[0] + tail[1];
        const unsigned long long removed = (unsigned long long)nv - tail[3] - tail[2];   // processed - counted
        out_k.reserve((size_t)n_seg * 8);
        out_c.reserve((size_t)n_seg * 4);
        {
            ScopedKernelTimer t(ctx, "replay");
            uint32_t* start = seg_start;   // seg_start is dead now: reuse as start[n_seg+1]
            hipLaunchKernelGGL(seg_starts_kernel, dim3(grid_for(nv)), dim3(256), 0, ctx->stream, head, seg_id, nv, n_seg, start);
            hipLaunchKernelGGL(emit_table_kernel, dim3(grid_for(n_seg)), dim3(256), 0, ctx->stream, hs, start, Ec, n_seg,
                               out_k.as<uint64_t>(), out_c.as<uint32_t>());
            SY_HIP(hipGetLastError());
        }
        n_out = n_seg;
        removed_out = removed;
    }
}


static void sketch_finish_impl(sylph_sketch* sk) {
    if (sk->finished) return;
    sylph_ctx* ctx = sk->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard dg(ctx->device);
    HostPhase ph_total(ctx, "finish: total incl. readback");
    SY_REQUIRE(sk->n_occ + sk->pend.n < (1ull << 32) - 1, "more than 2^32-2 seed occurrences in one sample");
    a10_mark(sk);                                // the reference's default dedup for pairs (no-op otherwise): the filter's answers go into the records first
    // fast path: bucket partition + in-LDS replay (replay_lds.hip); falls through to the device-wide sort path
    // below when a bucket does not fit in LDS (some k-mer with thousands of occurrences)
    if (ctx->finish_mode != 1) {
        if (finish_bucketed(sk)) { sk->finished = true; return; }
        SY_REQUIRE(ctx->finish_mode != 2, "bucket finish overflowed and finish=bucket forbids the fallback");
    }
    a10_settle(sk);            // (the partitioned filter pass's verdict, if finish_bucketed left before reading it)
    flush_pending_slots(sk);   // the device-wide path works on the dense file-order arrays ...
    if (sk->filter_dedup() && sk->a10_state != 2) a10_settle(sk);   // (... and a deferred batch that was redone just now is marked again)
    materialise_plain_records(sk);   // ... and on occurrence records
    sk->n_out = 0;
    sk->dup_removed = 0;
    generic_replay(ctx, sk->hash.as<uint64_t>(), sk->recs.as<OccRec>(), (uint32_t)sk->n_occ, sk->paired, sk->dedup_mode(), sk->out_k,
                   sk->out_c, sk->n_out, sk->dup_removed);
    sk->finished = true;
}

// ---- genome side -------------------------------------------------------------------------------------------

struct ContigSeeds { std::vector<uint32_t> contig; std::vector<uint64_t> pos, hash; };

static void seeds_positions_impl(sylph_ctx* ctx, const uint8_t* bases, const uint64_t* contig_off, uint64_t n_contigs,
                                 uint32_t c, uint32_t k, int seed_mode, ContigSeeds& out) {
    SY_REQUIRE(c >= 1, "c must be >= 1");
    SY_REQUIRE(seed_mode == SYLPH_SEED_SCALAR || seed_mode == SYLPH_SEED_AVX2_COMPAT, "bad seed_mode %d", seed_mode);
    SY_REQUIRE(k == 21 || k == 31, "k must be 21 or 31 (avx2_seeding.rs:46-52)");
    if (n_contigs == 0) return;
    SY_REQUIRE(contig_off && contig_off[0] == 0, "bad contig offsets");
    SY_REQUIRE(bases || contig_off[n_contigs] == 0, "null bases");
    SY_REQUIRE(n_contigs < (1ull << 32), "too many contigs");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard dg(ctx->device);
    const uint64_t n_bases = contig_off[n_contigs];
    SY_REQUIRE(n_bases < (1ull << 32), "genome larger than 2^32-1 bases: split by contig");
    DevBuf d_bases(ctx), d_off(ctx);
    d_bases.reserve(n_bases + 64);
    d_off.reserve((n_contigs + 1) * 8);
    ctx->h2d(d_bases.p, bases, n_bases);
    ctx->h2d(d_off.p, contig_off, (n_contigs + 1) * 8);
    ctx->counters.reserve(64);
    const uint32_t n = seeds_sorted_by_pos(ctx, d_bases.as<uint8_t>(), n_bases, c, k, ctx->counters.as<uint32_t>());
    if (!n) return;
    DevBuf o_contig(ctx), o_pos(ctx), o_hash(ctx);
    o_contig.reserve((size_t)n * 4);
    o_pos.reserve((size_t)n * 8);
    o_hash.reserve((size_t)n * 8);
    {
        ScopedKernelTimer t(ctx, "annotate");
        hipLaunchKernelGGL(annotate_contigs_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_off.as<uint64_t>(),
                           n_contigs, ctx->scratch[2].as<uint32_t>(), ctx->scratch[3].as<uint64_t>(), n, k,
                           seed_mode == SYLPH_SEED_AVX2_COMPAT, o_contig.as<uint32_t>(), o_pos.as<uint64_t>(),
                           o_hash.as<uint64_t>());
        SY_HIP(hipGetLastError());
    }
    std::vector<uint32_t> hc(n);
    std::vector<uint64_t> hp(n), hh(n);
    ctx->d2h(hc.data(), o_contig.p, (size_t)n * 4);
    ctx->d2h(hp.data(), o_pos.p, (size_t)n * 8);
    ctx->d2h(hh.data(), o_hash.p, (size_t)n * 8);
    out.contig.reserve(n); out.pos.reserve(n); out.hash.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (hh[i] == INVALID_HASH) continue;   // straddled a contig boundary / AVX2 tail / short contig
        out.contig.push_back(hc[i]); out.pos.push_back(hp[i]); out.hash.push_back(hh[i]);
    }
}

template <class T>
static T* to_malloc(const std::vector<T>& v) {
    T* p = (T*)malloc(std::max<size_t>(1, v.size()) * sizeof(T));
    if (!p) throw std::bad_alloc();
    if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace sylph

extern "C" {

int sylph_sketch_begin(sylph_ctx* ctx, uint32_t c, uint32_t k, int reads_mode, int no_dedup, int seed_mode,
                       sylph_sketch** out) {
    return guarded([&] {
        SY_REQUIRE(ctx && out, "null argument");
        SY_REQUIRE(c >= 1, "c must be >= 1");
        SY_REQUIRE(k == 21 || k == 31, "k must be 21 or 31 (avx2_seeding.rs:46-52)");
        SY_REQUIRE(reads_mode == SYLPH_READS_SINGLE || reads_mode == SYLPH_READS_PAIRED, "bad reads_mode %d", reads_mode);
        SY_REQUIRE(seed_mode == SYLPH_SEED_SCALAR || seed_mode == SYLPH_SEED_AVX2_COMPAT, "bad seed_mode %d", seed_mode);
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard dg(ctx->device);
        sylph_sketch* sk = new sylph_sketch(ctx);
        sk->c = c; sk->k = k;
        sk->paired = reads_mode == SYLPH_READS_PAIRED;
        sk->no_dedup = no_dedup != 0;
        sk->avx2_compat = seed_mode == SYLPH_SEED_AVX2_COMPAT;
        try {
            sk->counters.reserve(64);     // (every word of it is cleared by the path that uses it, right before: no memset dispatch per sample)
        } catch (...) { delete sk; throw; }
        ctx->refs++;
        *out = sk;
    });
}

int sylph_sketch_push(sylph_sketch* sk, const uint8_t* bases, const uint64_t* rec_off, uint64_t n_records, int mem) {
    return guarded([&] {
        SY_REQUIRE(sk, "null session");
        sketch_push_impl(sk, bases, rec_off, n_records, mem);
    });
}

int sylph_sketch_push_n(sylph_sketch* sk, const uint8_t* bases, const uint64_t* rec_off, uint64_t n_records, uint64_t n_bases,
                        int mem) {
    return guarded([&] {
        SY_REQUIRE(sk, "null session");
        sketch_push_impl(sk, bases, rec_off, n_records, mem, &n_bases);
    });
}

int sylph_sketch_push_enc(sylph_sketch* sk, const uint8_t* bases, const uint64_t* rec_off, uint64_t n_records, uint64_t n_bases, int mem,
                          int enc) {
    return guarded([&] {
        SY_REQUIRE(sk, "null session");
        sketch_push_impl(sk, bases, rec_off, n_records, mem, &n_bases, enc);
    });
}

// BYTE_TO_SEQ (types.rs:50-59) + packing, on the host: what a feed does before a SYLPH_ENC_2BIT push.  out holds (n + 3) / 4
// bytes; base i lands in bits 7-2(i%4) .. 6-2(i%4) of byte i/4.
int sylph_pack_2bit(const uint8_t* ascii, uint64_t n, uint8_t* out) {
    return guarded([&] {
        SY_REQUIRE((ascii && out) || n == 0, "null argument");
        static uint8_t lut[256];
        static std::once_flag once;
        std::call_once(once, [] {
            memset(lut, 0, sizeof(lut));
            lut[1] = 1; lut[2] = 2; lut[3] = 3;
            const char* s4 = "AaCcGgTtUu";
            const uint8_t v[10] = {0, 0, 1, 1, 2, 2, 3, 3, 3, 3};
            for (int i = 0; i < 10; i++) lut[(uint8_t)s4[i]] = v[i];
        });
        uint64_t i = 0;
        for (; i + 4 <= n; i += 4)
            out[i >> 2] = (uint8_t)((lut[ascii[i]] << 6) | (lut[ascii[i + 1]] << 4) | (lut[ascii[i + 2]] << 2) | lut[ascii[i + 3]]);
        if (i < n) {
            uint8_t b = 0;
            for (uint64_t j = i; j < n; j++) b |= (uint8_t)(lut[ascii[j]] << (6 - 2 * (j - i)));
            out[i >> 2] = b;
        }
    });
}

int sylph_sketch_finish_device(sylph_sketch* sk, const uint64_t** dev_kmers, const uint32_t** dev_counts, uint64_t* out_n,
                               uint64_t* out_dup_removed) {
    return guarded([&] {
        SY_REQUIRE(sk && dev_kmers && dev_counts && out_n, "null argument");
        sketch_finish_impl(sk);
        *dev_kmers = sk->out_k.as<uint64_t>();
        *dev_counts = sk->out_c.as<uint32_t>();
        *out_n = sk->n_out;
        if (out_dup_removed) *out_dup_removed = sk->dup_removed;
    });
}

int sylph_sketch_finish(sylph_sketch* sk, uint64_t** out_kmers, uint32_t** out_counts, uint64_t* out_n,
                        uint64_t* out_dup_removed) {
    return guarded([&] {
        SY_REQUIRE(sk && out_kmers && out_counts && out_n, "null argument");
        sketch_finish_impl(sk);
        const size_t n = sk->n_out;
        uint64_t* hk = (uint64_t*)malloc(std::max<size_t>(1, n) * 8);
        uint32_t* hc = (uint32_t*)malloc(std::max<size_t>(1, n) * 4);
        if (!hk || !hc) { free(hk); free(hc); throw std::bad_alloc(); }
        try {
            if (n) {
                std::lock_guard<std::mutex> lock(sk->ctx->mu);
                DeviceGuard dg(sk->ctx->device);
                sk->ctx->d2h(hk, sk->out_k.p, n * 8);
                sk->ctx->d2h(hc, sk->out_c.p, n * 4);
            }
        } catch (...) { free(hk); free(hc); throw; }
        *out_kmers = hk; *out_counts = hc; *out_n = n;
        if (out_dup_removed) *out_dup_removed = sk->dup_removed;
    });
}

int sylph_sketch_set_option(sylph_sketch* sk, const char* key, const char* value) {
    return guarded([&] {
        SY_REQUIRE(sk && key && value, "null argument");
        std::lock_guard<std::mutex> lock(sk->ctx->mu);
        if (!strcmp(key, "borrow_until_finish")) sk->borrow_until_finish = strtol(value, nullptr, 10) != 0;
        else if (!strcmp(key, "a10")) {             // A/B and test knob: which pass marks the filter's answers (a10.hip)
            if (!strcmp(value, "auto")) sk->a10_force = 0;
            else if (!strcmp(value, "walk")) sk->a10_force = 1;
            else if (!strcmp(value, "part")) sk->a10_force = 2;
            else SY_REQUIRE(false, "a10 must be auto|walk|part");
        }
        else if (!strcmp(key, "dedup_fpr") || !strcmp(key, "dedup_capacity")) {
            SY_REQUIRE_STATE(sk->rec_base == 0 && !sk->finished, "%s must be set before the first push", key);
            if (key[6] == 'f') {
                const double v = strtod(value, nullptr);
                SY_REQUIRE(v >= 0. && v < 1., "dedup_fpr must be in [0, 1) (got %s)", value);
                sk->dedup_fpr = v;
            } else {
                const long long v = strtoll(value, nullptr, 10);
                SY_REQUIRE(v >= 1 && v < (1ll << 31), "dedup_capacity must be in [1, 2^31) (got %s)", value);
                // the filter has the next power of two above capacity / 4 buckets of four entries: a capacity that fills them beyond 80 %
                // (8, 4096: 100 %) makes cuckoo insertions fail, and what a filter answers then depends on its eviction history — which
                // a10.hip does not replay (the reference's 10^7 fills 59.6 %; every doubling keeps the ratio)
                uint64_t nb = 1;
                while (nb * 4 < (uint64_t)v) nb <<= 1;
                SY_REQUIRE((double)v <= 0.8 * 4.0 * (double)nb, "dedup_capacity %lld would fill its %llu buckets to %.0f %%: insertions fail above ~80 %% and "
                           "the result would depend on the eviction order; choose a capacity further from a power of two", v, (unsigned long long)nb,
                           100.0 * (double)v / (4.0 * (double)nb));
                sk->dedup_capacity = (uint64_t)v;
            }
        }
        else SY_REQUIRE(false, "unknown session option %s", key);
    });
}

void sylph_sketch_destroy(sylph_sketch* sk) {
    if (!sk) return;
    sylph_ctx* ctx = sk->ctx;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        delete sk;   // buffers go back to the ctx pool; stream order protects in-flight work
    }
    ctx_unref(ctx);
}

int sylph_seeds_positions(sylph_ctx* ctx, const uint8_t* bases, const uint64_t* contig_off, uint64_t n_contigs, uint32_t c,
                          uint32_t k, int seed_mode, uint32_t** out_contig, uint64_t** out_pos, uint64_t** out_hash,
                          uint64_t* out_n) {
    return guarded([&] {
        SY_REQUIRE(ctx && out_contig && out_pos && out_hash && out_n, "null argument");
        ContigSeeds s;
        seeds_positions_impl(ctx, bases, contig_off, n_contigs, c, k, seed_mode, s);
        *out_contig = to_malloc(s.contig);
        *out_pos = to_malloc(s.pos);
        *out_hash = to_malloc(s.hash);
        *out_n = s.hash.size();
    });
}

int sylph_seeds(sylph_ctx* ctx, const uint8_t* bases, uint64_t len, uint32_t c, uint32_t k, int seed_mode,
                uint64_t** out_hashes, uint64_t* out_n) {
    return guarded([&] {
        SY_REQUIRE(ctx && out_hashes && out_n, "null argument");
        // one record; the read rule for short sequences (k+1, avx2_seeding.rs:42) differs from the contig rule
        // (2k, :160), so run the read flavour through a throw-away session-less path
        SY_REQUIRE(c >= 1, "c must be >= 1");
        SY_REQUIRE(k == 21 || k == 31, "k must be 21 or 31 (avx2_seeding.rs:46-52)");
        SY_REQUIRE(seed_mode == SYLPH_SEED_SCALAR || seed_mode == SYLPH_SEED_AVX2_COMPAT, "bad seed_mode %d", seed_mode);
        std::vector<uint64_t> res;
        if (len) {
            SY_REQUIRE(bases, "null bases");
            SY_REQUIRE(len < (1ull << 32), "sequence longer than 2^32-1 bases");
            std::lock_guard<std::mutex> lock(ctx->mu);
            DeviceGuard dg(ctx->device);
            DevBuf d_bases(ctx);
            d_bases.reserve(len + 64);
            ctx->h2d(d_bases.p, bases, len);
            ctx->counters.reserve(64);
            const uint32_t n = seeds_sorted_by_pos(ctx, d_bases.as<uint8_t>(), len, c, k, ctx->counters.as<uint32_t>());
            if (n) {
                std::vector<uint32_t> hp(n);
                std::vector<uint64_t> hh(n);
                ctx->d2h(hp.data(), ctx->scratch[2].p, (size_t)n * 4);
                ctx->d2h(hh.data(), ctx->scratch[3].p, (size_t)n * 8);
                const uint64_t nk = n_hashed_kmers(len, k, seed_mode == SYLPH_SEED_AVX2_COMPAT, 0);
                for (uint32_t i = 0; i < n; i++)
                    if (hp[i] < nk) res.push_back(hh[i]);
            }
        }
        *out_hashes = to_malloc(res);
        *out_n = res.size();
    });
}

}  // extern "C"
