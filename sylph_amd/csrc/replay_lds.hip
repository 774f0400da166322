// replay_lds.hip — finish() of a read-sketch session without device-wide sorts: bucket partition + in-LDS replay.
//
// FracMinHash survivors are uniformly distributed below the threshold (mm_hash64 is a bijection of canonical
// k-mers), so the top bits of the hash split the sample's occurrences into B buckets of nearly equal size (~128).
// The PARTITION (round 3: hand-written, four small kernels, no library sort, no memset dispatches) only moves 8-byte
// (bucket, occurrence index) pairs, in two levels: tiles of occurrences are histogrammed over <= 512 coarse hash ranges, a
// scan turns the (range x tile) counts into offsets, the pairs are scattered range by range, and one workgroup per coarse
// range finishes with a counting sort by bucket in LDS, which also yields the bucket offsets.  Neither level keeps the file
// order inside a bucket (ranks come from LDS atomics): the occurrence index IS the file order, and the replay workgroup
// re-establishes it for its ~128 occurrences with one rank loop.  The 32 B occurrence records never move: the replay gathers
// them through the permutation — from the session's dense arrays, or straight from the per-block slots the seeding kernel
// left them in when the sample came in one batch (no compaction pass at all).
// One workgroup then owns one bucket entirely in LDS: counting-rank sort by (hash, file order) -> k-mer segments in file
// order -> mate-2 skip, duplicate flags, cut-off and counts (the same data-parallel formulation of
// dup_removal_lsh_full_exact as sketch.hip, see its header) -> distinct (k-mer, count) pairs.  Buckets are ordered
// by hash, so concatenating their outputs gives the table in ascending k-mer order.
// A bucket beyond 1024 occurrences (a k-mer more than ~1000 deep) goes through the device-wide path of sketch.hip, as a
// small sample of its own.
#include "common.h"
#include "device_common.h"
#include "sketch_session.h"
#include "partition.h"

namespace sylph {
namespace {

// Three configurations of the same kernel template.  Buckets of up to CAP_SMALL occurrences (all of them, for ordinary
// samples) run with 10 KiB of LDS per workgroup -> 16 workgroups = 32 wavefronts per CU, which is what hides the latency of
// this barrier- and gather-heavy kernel (with a single 512-slot configuration occupancy was 14 wavefronts and the kernel 1.4x
// slower).  Larger buckets, and buckets that hold a deep k-mer (SEG_LIMIT), are queued for the CAP_MID / CAP_LARGE
// configurations, which replace the scan over a k-mer's earlier occurrences by a hash table in LDS and are launched only
// when something was queued; only beyond CAP_LARGE does a bucket take the device-wide path.
#ifndef SYLPH_REPLAY_TPB
#define SYLPH_REPLAY_TPB 128
#endif
#ifndef SYLPH_REPLAY_TAGS
#define SYLPH_REPLAY_TAGS 1
#endif
constexpr int CAP_SMALL = 256, RTPB_SMALL = SYLPH_REPLAY_TPB;
constexpr int CAP_MID = 512, RTPB_MID = 256;       // hashed marker test, ~31 KiB of LDS: 5 workgroups per CU
constexpr int CAP_LARGE = 1024, RTPB_LARGE = 256;   // hashed marker test, ~59 KiB of LDS: 2 workgroups per CU
constexpr int IDX_BITS = 10;         // arrival index inside a bucket (< CAP_LARGE)
// The marker test of the small configuration looks at every earlier occurrence of the k-mer: quadratic in a k-mer's coverage.
// A bucket holding a k-mer with SEG_LIMIT or more occurrences (a genome at ~100x and above) is handed to the medium / large
// configuration, whose marker test is a hash table in LDS: linear in the bucket size.
constexpr uint32_t SEG_LIMIT = 96;


// 15-bit tag of a dedup marker, never 0 (bit 0 set): what the scan over a k-mer's earlier occurrences compares first
__device__ __forceinline__ uint32_t marker_tag(uint64_t m) { return (uint32_t)((m * 0x9E3779B97F4A7C15ull) >> 49) | 1u; }

// One workgroup = one bucket.  SINGLE_CUTOFF = 4 for single-end (sketch.rs:937), 0 for pairs.
//
// Ordering inside the bucket: the partition hands over the bucket's occurrences in no particular order, but the index an
// occurrence is gathered by (position in the dense arrays / slot number) grows with the file order: a first rank loop over the
// indices gives every occurrence its ARRIVAL number (its place by (record, position) inside the bucket); what is left is a
// sort by (hash, arrival).  Each lane keeps its (up to ITEMS) records in registers, publishes one 64-bit key per record in
// LDS and finds the record's sorted position by counting smaller keys — every lane reads the same LDS word per step (a
// broadcast, no bank conflicts), the loop has no barriers and no dependent LDS round trips, and it is O(n^2 / lanes) with
// n ~ 200.  Keys are unique: the bucket's hashes lie in one narrow range, so key = (hash - lowest hash of the bucket) << 10 |
// arrival number whenever that difference fits in 54 bits (bm.composite, decided by the host; else — tiny samples — the
// two-part comparison is spelled out).  Records are then written straight to their sorted slots.
// Handles buckets with min_n < n <= CAP; larger ones bump `overflow` (when count_overflow) and are left to the caller.
template <int CAP, int RTPB>
__device__ __forceinline__ void replay_bucket(const uint32_t b, const OccRec* __restrict__ recs, const uint32_t* __restrict__ perm,
                                                             const uint32_t* __restrict__ boff,
                                                             const uint32_t* __restrict__ p_nv, int paired, int no_dedup,
                                                             uint32_t cutoff, BucketMap bm, uint64_t* __restrict__ tmp_k,
                                                             uint32_t* __restrict__ tmp_c, uint32_t* __restrict__ n_distinct,
                                                             uint32_t* __restrict__ removed_b,
                                                             uint32_t* __restrict__ overflow, uint32_t* __restrict__ mid_list,
                                                             uint32_t* __restrict__ large_list,
                                                             uint32_t* __restrict__ ovf_list, int dbg_stage) {
    constexpr int ITEMS = CAP / RTPB;     // records per lane
    // `no_dedup` arrives as a DEDUP_* mode.  DEDUP_FILTER (the reference's default for pairs, a10.hip): everything as in the exact
    // mode except the marker test itself, which is the bit a10_mark left in the occurrence's record.
    const bool filter = no_dedup == DEDUP_FILTER;
    if (filter) no_dedup = 0;
    __shared__ uint64_t s_hash[CAP], s_rid[CAP], s_m0[CAP], s_m1[CAP];
    __shared__ __attribute__((aligned(8))) uint16_t s_seg[CAP];   // first sorted position of the k-mer each sorted position belongs to
    __shared__ uint8_t s_fl[CAP];         // bit0 skip, bit1 would-be-dropped
    __shared__ __attribute__((aligned(8))) uint16_t s_ab[2 * (CAP + 2)];   // exclusive counts <= CAP (s_a | s_b); before them: the marker tags
    uint16_t* const s_a = s_ab;
    uint16_t* const s_b = s_ab + (CAP + 2);
    __shared__ uint32_t s_wave[RTPB / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t nv = *p_nv;
    const uint32_t first = boff[b], last = boff[b + 1];
    const uint32_t n = last - first;
    if (last > nv || first > last) { if (tid == 0) atomicAdd(overflow, 1u); return; }   // defensive: inconsistent bounds
    if (n == 0) return;                   // (n_distinct was zeroed by the host)
    // too large for this configuration: queue it for the large one (large_list: [0] = count, [1..] = buckets), or for the
    // host, which sends the occurrences of such buckets through the device-wide path (ovf_list, same layout)
    if (n > CAP) {
        if (tid == 0) {
            uint32_t* list = (CAP < CAP_MID && n <= (uint32_t)CAP_MID) ? mid_list : (CAP < CAP_LARGE && n <= (uint32_t)CAP_LARGE) ? large_list : ovf_list;
            list[1 + atomicAdd(&list[0], 1u)] = b;
        }
        return;
    }
    // ---- gather (through the partition permutation, one 32 B sector per occurrence) + sort by (hash, file order) -------
    uint64_t* s_key = s_m0;               // keys live in s_m0 until the sorted records are written
    const bool composite = bm.composite != 0;
    // lowest hash that maps to this bucket: hs >= ceil(b * 2^32 / mult)  (exact inverse of bucket_of)
    const uint64_t lo_hash = composite ? (((((uint64_t)b << 32) + bm.mult - 1u) / bm.mult) << bm.sh) : 0ull;
    OccRec r[ITEMS];
    uint32_t pidx[ITEMS], rank[ITEMS];
    uint32_t* const s_pidx = reinterpret_cast<uint32_t*>(s_rid);   // (s_rid is free until the sorted records are written)
    const int levels = (int)((n + RTPB - 1) / RTPB);   // lanes of level q hold a record iff q < levels (wave-uniform)
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RTPB;
        pidx[q] = 0xFFFFFFFFu;
        rank[q] = 0;
        if (i < n) {
            pidx[q] = perm[first + i];
            r[q] = recs[pidx[q]];
        }
    }
    if (composite) {
        // Sub-bin sort, linear in the bucket: the bucket's hashes are uniform over its narrow range, so CAP equal sub-ranges hold
        // about half an occurrence each.  Count per sub-range (LDS atomics), scan, drop every (hash, index) into its sub-range
        // (any order), then each occurrence ranks itself among the few members of its own sub-range: sorted position =
        // start of the sub-range + members with a smaller (hash, index).  The occurrences of one k-mer share a sub-range: for
        // them that loop is as long as the k-mer is deep, like the marker test below.  (Rounds 1-2 ranked every occurrence
        // against the whole bucket, n^2 comparisons; with the partition no longer stable a second such loop over the indices
        // would have been needed on top.)
        uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(s_m1);   // CAP + 1 counters (s_m1 is free until the sorted records are written)
        const uint32_t sub_mult = bm.sub_mult[CAP == CAP_SMALL ? 0 : CAP == CAP_MID ? 1 : 2];
        uint32_t sub[ITEMS];
        constexpr int CFG = CAP == CAP_SMALL ? 0 : CAP == CAP_MID ? 1 : 2;
        const int rank_bits = bm.rank_bits[CFG];
        // hashed configurations (deep buckets): lowest and highest hash of every sub-range (s_hash / s_rid are free until the
        // sorted records are written) — a sub-range whose two are equal holds ONE k-mer, see the second level below
        constexpr bool TWO_LEVEL = CAP != CAP_SMALL;
        unsigned long long* const s_min = reinterpret_cast<unsigned long long*>(s_hash);
        unsigned long long* const s_max = reinterpret_cast<unsigned long long*>(s_rid);
        for (uint32_t t = tid; t <= (uint32_t)CAP; t += RTPB) s_cnt[t] = 0;
        if constexpr (TWO_LEVEL)
            for (uint32_t t = tid; t < (uint32_t)CAP; t += RTPB) { s_min[t] = ~0ull; s_max[t] = 0ull; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            const uint32_t i = tid + q * RTPB;
            sub[q] = 0;
            if (i < n) {
                const uint32_t hsres = (uint32_t)((r[q].hash - lo_hash) >> bm.sh);          // < bm.range_hs
                sub[q] = sub_mult ? min(__umulhi(hsres, sub_mult), (uint32_t)CAP - 1u) : min(hsres, (uint32_t)CAP - 1u);
                atomicAdd(&s_cnt[sub[q]], 1u);
                if constexpr (TWO_LEVEL) {
                    atomicMin(&s_min[sub[q]], (unsigned long long)r[q].hash);
                    atomicMax(&s_max[sub[q]], (unsigned long long)r[q].hash);
                }
            }
        }
        __syncthreads();
        {   // exclusive scan of the CAP counters (CAP / RTPB per lane): s_cnt[t] = start of sub-range t, s_cnt[CAP] = n
            constexpr int PER = CAP / RTPB;
            uint32_t v[PER], sum = 0;
#pragma unroll
            for (int e = 0; e < PER; e++) { v[e] = s_cnt[tid * PER + e]; sum += v[e]; }
            uint32_t run = block_excl_sum<RTPB>(sum, s_wave, nullptr);
            bool deep = false;
#pragma unroll
            for (int e = 0; e < PER; e++) { s_cnt[tid * PER + e] = run; run += v[e]; deep |= v[e] >= SEG_LIMIT; }
            if (tid == RTPB - 1) s_cnt[CAP] = run;
            // A k-mer SEG_LIMIT deep fills its sub-range that far (equal hashes share a sub-range): such a bucket is for the
            // hashed marker test of the next configuration — pass it on now, before the placement, the ranking (as long as the k-mer
            // is deep, per occurrence) and the segment scans are spent on it here.  (A sub-range that full without a deep k-mer
            // does not happen with ~0.5 occurrences per sub-range; the next configuration is right for any bucket it can hold.)
            if constexpr (CAP == CAP_SMALL) {
                if (__syncthreads_or(deep && !no_dedup)) {
                    if (tid == 0) mid_list[1 + atomicAdd(&mid_list[0], 1u)] = b;
                    return;
                }
            } else
                __syncthreads();
        }
        // Second level (hashed configurations): the occurrences of a DEEP k-mer all sit in one sub-range, and ranking them among
        // each other by index is quadratic in the depth.  A sub-range that holds one k-mer only (lowest hash = highest hash) and
        // at least DEEP_SUB occurrences is therefore cut once more, by INDEX: cnt equal index ranges own one place each of the
        // sub-range's cnt places (the occurrences of a k-mer are spread over the file like the reads are, so the ranges hold
        // about one each; whatever they hold is ranked inside its range, so any spread is sorted correctly).  Bin of an
        // occurrence = first place of its sub-range (+ its index range): bins are places, they order like (hash, index).
        uint32_t bin[ITEMS];
        const uint32_t* starts = s_cnt;
#pragma unroll
        for (int q = 0; q < ITEMS; q++) bin[q] = sub[q];
        if constexpr (TWO_LEVEL) {
            if (rank_bits) {
                constexpr uint32_t DEEP_SUB = 32;
                uint32_t* const s_c2 = reinterpret_cast<uint32_t*>(s_m0);       // CAP + 1 counters (s_key is not written before the placement)
                uint32_t* const s_s2 = reinterpret_cast<uint32_t*>(s_hash);     // their scan (s_min is done with by then)
                for (uint32_t t = tid; t <= (uint32_t)CAP; t += RTPB) s_c2[t] = 0;
                __syncthreads();
#pragma unroll
                for (int q = 0; q < ITEMS; q++) {
                    const uint32_t i = tid + q * RTPB;
                    if (i < n) {
                        const uint32_t lo = s_cnt[sub[q]], cnt = s_cnt[sub[q] + 1] - lo;
                        const bool pure = cnt >= DEEP_SUB && s_min[sub[q]] == s_max[sub[q]];
                        bin[q] = lo + (pure ? min(cnt - 1u, (uint32_t)(((uint64_t)pidx[q] * cnt) >> rank_bits)) : 0u);
                        atomicAdd(&s_c2[bin[q]], 1u);
                    }
                }
                __syncthreads();
                {
                    constexpr int PER = CAP / RTPB;
                    uint32_t v[PER], sum = 0;
#pragma unroll
                    for (int e = 0; e < PER; e++) { v[e] = s_c2[tid * PER + e]; sum += v[e]; }
                    uint32_t run = block_excl_sum<RTPB>(sum, s_wave, nullptr);
#pragma unroll
                    for (int e = 0; e < PER; e++) { s_s2[tid * PER + e] = run; run += v[e]; }
                    if (tid == RTPB - 1) s_s2[CAP] = run;
                }
                __syncthreads();
                starts = s_s2;
            }
        }
        // place (cursor = a second counter array would cost LDS: take places from the END of each sub-range instead, counting the
        // start words' neighbours down is not possible either — so the places come from s_seg, which is free until the segments)
        uint16_t* const s_fill = s_seg;                               // members placed so far per sub-range (<= CAP: 16 bits do)
        for (uint32_t t = tid; t < (uint32_t)CAP; t += RTPB) s_fill[t] = 0;
        __syncthreads();
        // ranking key of an occurrence inside its sub-range: (hash - a lower bound of the sub-range's hashes, index) in one word
        // when the host found room for both (rank_bits > 0), else the hash with the indices in a second array
        uint64_t rkey[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            const uint32_t i = tid + q * RTPB;
            rkey[q] = 0;
            if (i < n) {
                // 16-bit LDS atomics do not exist: the counter pairs share a word; add 1 or 65536 to the word and take the half
                uint32_t* const w = reinterpret_cast<uint32_t*>(s_fill) + (bin[q] >> 1);
                const uint32_t old = atomicAdd(w, (bin[q] & 1u) ? 65536u : 1u);
                const uint32_t place = starts[bin[q]] + ((bin[q] & 1u) ? (old >> 16) : (old & 0xFFFFu));
                if (rank_bits) {
                    const uint64_t res = (r[q].hash - lo_hash) - ((uint64_t)(sub[q] * bm.sub_width[CFG]) << bm.sh);
                    rkey[q] = (res << rank_bits) | pidx[q];
                    s_key[place] = rkey[q];
                } else {
                    s_key[place] = r[q].hash;
                    s_pidx[place] = pidx[q];
                }
            }
        }
        __syncthreads();
        if (dbg_stage == 1) { if (tid == 0) n_distinct[b] = 0; return; }
        if (rank_bits) {
#pragma unroll
            for (int q = 0; q < ITEMS; q++) {
                const uint32_t i = tid + q * RTPB;
                if (i < n) {
                    const uint32_t lo = starts[bin[q]], hi = starts[bin[q] + 1];
                    uint32_t smaller = 0;
                    for (uint32_t p = lo; p < hi; p++) smaller += s_key[p] < rkey[q] ? 1u : 0u;
                    rank[q] = lo + smaller;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < ITEMS; q++) {
                const uint32_t i = tid + q * RTPB;
                if (i < n) {
                    const uint32_t lo = starts[bin[q]], hi = starts[bin[q] + 1];
                    uint32_t smaller = 0;
                    for (uint32_t p = lo; p < hi; p++) {
                        const uint64_t kp = s_key[p];
                        smaller += (kp < r[q].hash || (kp == r[q].hash && s_pidx[p] < pidx[q])) ? 1u : 0u;
                    }
                    rank[q] = lo + smaller;
                }
            }
        }
    } else {
        // tiny samples (the bucket's hash range does not fit the key): every occurrence against every other, as in rounds 1-2
        uint16_t* const s_arr = s_a;                                  // (free until the counts)
        uint32_t arrival[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            const uint32_t i = tid + q * RTPB;
            arrival[q] = 0;
            if (i < n) s_pidx[i] = pidx[q];
        }
        __syncthreads();
        // arrival number = how many of the bucket's occurrences come earlier in the file (their indices are distinct)
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t pj = s_pidx[j];
#pragma unroll
            for (int q = 0; q < ITEMS; q++)
                if (q < levels) arrival[q] += (pj < pidx[q]) ? 1u : 0u;
        }
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            const uint32_t i = tid + q * RTPB;
            if (i < n) { s_key[i] = r[q].hash; s_arr[i] = (uint16_t)arrival[q]; }
        }
        __syncthreads();
        if (dbg_stage == 1) { if (tid == 0) n_distinct[b] = 0; return; }
        for (uint32_t j = 0; j < n; j++) {
            const uint64_t kj = s_key[j];
#pragma unroll
            for (int q = 0; q < ITEMS; q++)
                if (q < levels && tid + q * RTPB < n)
                    rank[q] += ((kj < r[q].hash) || (kj == r[q].hash && (uint32_t)s_arr[j] < arrival[q])) ? 1u : 0u;
        }
    }
    __syncthreads();                      // every lane is done with s_key (= s_m0), s_pidx (= s_rid), the counters (= s_m1, s_seg, s_a)
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RTPB;
        if (i < n) {
            const uint32_t d = rank[q];
            s_hash[d] = r[q].hash; s_rid[d] = r[q].rid; s_m0[d] = r[q].m0; s_m1[d] = r[q].m1;
        }
    }
    __syncthreads();
    if (dbg_stage == 2) { if (tid == 0) n_distinct[b] = 0; return; }
    // ---- segments ---------------------------------------------------------------------------------------------
    // lane owns `items` contiguous sorted positions; s_seg = running "last head seen" (segmented max-scan)
    const uint32_t items = (n + RTPB - 1) / RTPB;
    const uint32_t j0 = tid * items;
    uint32_t heads = 0, last_head = 0;
    bool has_head = false;
    uint8_t headbits = 0;
    for (uint32_t t = 0; t < items; t++) {
        const uint32_t j = j0 + t;
        if (j >= n) break;
        const bool hd = (j == 0) || (s_hash[j] != s_hash[j - 1]);
        if (hd) { heads++; last_head = j; has_head = true; headbits |= (uint8_t)(1u << t); }
    }
    // inclusive max-scan of last_head over lanes (a lane without a head inherits from the left)
    uint32_t carry = has_head ? last_head + 1 : 0;   // +1 so that 0 means "none"
    {
        const uint32_t lane = tid & 63, wave = tid >> 6;
        uint32_t x = carry;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x = max(x, y);
        }
        __syncthreads();
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        uint32_t left = 0;
        for (uint32_t w = 0; w < wave; w++) left = max(left, s_wave[w]);
        const uint32_t prev = max(left, __shfl_up(x, 1));   // inclusive result of the lane to the left
        carry = (lane == 0) ? left : prev;
    }
    {
        uint32_t cur = carry;   // last head (+1) before this lane's first position
        for (uint32_t t = 0; t < items; t++) {
            const uint32_t j = j0 + t;
            if (j >= n) break;
            if (headbits & (1u << t)) cur = j + 1;
            s_seg[j] = (uint16_t)(cur - 1);
        }
    }
    __syncthreads();
    constexpr bool HASHED = CAP != CAP_SMALL;
    constexpr uint32_t MARKER_TAB = 4 * CAP;    // slots of the marker table: 2 x (2 entries per occurrence)
    if constexpr (!HASHED) {
        // a long k-mer segment of occurrences that carry markers: not for the quadratic marker test below (reads above 400 bases
        // carry none — the test does not run for them, however deep the k-mer)
        if (!no_dedup) {
            __shared__ uint32_t s_longest;
            if (tid == 0) s_longest = 0;
            __syncthreads();
            uint32_t mine = 0;
            for (uint32_t t = 0; t < items; t++) {
                const uint32_t j = j0 + t;
                if (j >= n) break;
                if (s_rid[j] & RID_MARKER_BIT) mine = max(mine, j - (uint32_t)s_seg[j] + 1);
            }
            if (mine >= SEG_LIMIT) atomicMax(&s_longest, mine);
            __syncthreads();
            if (s_longest >= SEG_LIMIT) {
                if (tid == 0) mid_list[1 + atomicAdd(&mid_list[0], 1u)] = b;
                return;
            }
        }
    }
    // DEDUP_FILTER: `*c > 0` (sketch.rs:749, :756) = an occurrence of the k-mer went through the dedup before this one.  The walk hands
    // over a record's seeds in EMISSION order (the rank bits of the rid; lane-interleaved for the AVX2 routine), the segment lists them
    // by position: the first of the walk is the lowest rank among the segment's leading occurrences of the head's record — the head
    // itself unless the read repeats the k-mer (a tandem repeat inside one read).  (None of those is a skipped mate 2: the mate-1
    // occurrence that would make it one belongs to an earlier record.)
    auto walk_first = [&](uint32_t j) {
        const uint32_t s0 = s_seg[j];
        const uint64_t rec0 = s_rid[s0] & RID_MASK, rj = s_rid[j];
        if ((rj & RID_MASK) != rec0) return false;
        const uint64_t rank_j = (rj >> RID_RANK_SHIFT) & RID_RANK_MAX;
        for (uint32_t q = s0; q < n && (uint32_t)s_seg[q] == s0 && (s_rid[q] & RID_MASK) == rec0; q++)
            if (q != j && ((s_rid[q] >> RID_RANK_SHIFT) & RID_RANK_MAX) < rank_j) return false;
        return true;
    };
    uint32_t* const s_tag = reinterpret_cast<uint32_t*>(s_ab);           // CAP words: fits the 2 x (CAP + 2) halfwords of s_a | s_b, which are written later
    (void)s_tag;
    // ---- mate-2 skip (sketch.rs:852) and duplicate flags ----------------------------------------------------------
    for (uint32_t t = 0; t < items; t++) {
        const uint32_t j = j0 + t;
        if (j >= n) break;
        uint8_t fl = 0;
        if (paired) {
            const uint64_t rec = s_rid[j] & RID_MASK;
            if (rec & 1) {
                const uint32_t s0 = s_seg[j];
                for (uint32_t q = j; q > s0;) {
                    q--;
                    const uint64_t rq = s_rid[q] & RID_MASK;
                    if ((rq >> 1) != (rec >> 1)) break;
                    if ((rq & 1) == 0) { fl = 1; break; }
                }
            }
        }
        s_fl[j] = fl;
#if SYLPH_REPLAY_TAGS
        if constexpr (CAP == CAP_SMALL)
            s_tag[j] = (!fl && (s_rid[j] & RID_MARKER_BIT)) ? (marker_tag(s_m0[j]) | (marker_tag(s_m1[j]) << 16)) : 0u;
#endif
    }
    __syncthreads();
    uint32_t my_u = 0;
    uint8_t ubits = 0;
    if constexpr (HASHED) {
        // Marker test through a hash table in LDS.  Every processed occurrence with markers enters both of them under the key
        // (k-mer segment, marker value); a slot belongs to the first entry that claims it (owner entry in the high half of the
        // word, never changes) and keeps the smallest sorted position among the entries with its key in the low half.  An
        // occurrence is a duplicate when one of its two keys was entered from an earlier position (sketch.rs:709-722: markers
        // go into the set whether the occurrence is then counted or dropped), or when its two markers are equal — and it is
        // not the first of its k-mer: the head of a segment is never a skipped mate 2 (the mate-1 occurrence that would make
        // it one precedes it in the segment), so "a processed occurrence precedes j" is simply "j is not the head".
        __shared__ uint32_t s_tab[MARKER_TAB];
        __shared__ uint16_t s_slot[2 * CAP];
        static_assert((MARKER_TAB & (MARKER_TAB - 1)) == 0 && 2 * CAP <= 0xFFFF, "marker table geometry");
        for (uint32_t t = tid; t < MARKER_TAB; t += RTPB) s_tab[t] = 0xFFFFFFFFu;
        __syncthreads();
        auto marker_of = [&](uint32_t e) { return (e & 1u) ? s_m1[e >> 1] : s_m0[e >> 1]; };
        if (!no_dedup && !filter) {
            for (uint32_t t = 0; t < items; t++) {
                const uint32_t j = j0 + t;
                if (j >= n) break;
                if ((s_fl[j] & 1) || !(s_rid[j] & RID_MARKER_BIT)) continue;
                const uint32_t seg = s_seg[j];
                for (uint32_t w = 0; w < 2; w++) {
                    const uint32_t e = 2 * j + w;
                    const uint64_t v = marker_of(e);
                    uint32_t h = (uint32_t)(((v ^ (v >> 31) ^ ((uint64_t)seg << 17)) * 0x9E3779B97F4A7C15ull) >> 40) & (MARKER_TAB - 1);
                    for (;;) {
                        const uint32_t old = atomicCAS(&s_tab[h], 0xFFFFFFFFu, (e << 16) | j);
                        const uint32_t o = old == 0xFFFFFFFFu ? e : old >> 16;
                        if (marker_of(o) == v && s_seg[o >> 1] == seg) {
                            if (old != 0xFFFFFFFFu) atomicMin(&s_tab[h], (o << 16) | j);
                            s_slot[e] = (uint16_t)h;
                            break;
                        }
                        h = (h + 1) & (MARKER_TAB - 1);
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t t = 0; t < items; t++) {
            const uint32_t j = j0 + t;
            if (j >= n) break;
            uint8_t fl = s_fl[j];
            if (!fl && !no_dedup && (s_rid[j] & RID_MARKER_BIT) && (filter || j != (uint32_t)s_seg[j])) {
                if (filter) {      // sketch.rs:747-760: the filter's answers, `*c > 0` = not the first of the k-mer in the walk
                    if ((s_rid[j] & RID_A10_BIT) && !walk_first(j)) fl |= 2;
                } else {
                    const bool hit = (s_tab[s_slot[2 * j]] & 0xFFFFu) < j || (s_tab[s_slot[2 * j + 1]] & 0xFFFFu) < j;
                    if (hit || s_m0[j] == s_m1[j]) fl |= 2;
                }
            }
            const bool u = !(fl & 1) && (no_dedup || !(fl & 2));
            if (u) { my_u++; ubits |= (uint8_t)(1u << t); }
            s_fl[j] = fl;
        }
    } else {
        for (uint32_t t = 0; t < items; t++) {
            const uint32_t j = j0 + t;
            if (j >= n) break;
            uint8_t fl = s_fl[j];
            if (filter) {
                if (!fl && (s_rid[j] & RID_MARKER_BIT) && (s_rid[j] & RID_A10_BIT) && !walk_first(j)) fl |= 2;
            } else if (!fl && !no_dedup && (s_rid[j] & RID_MARKER_BIT)) {
                const uint64_t a = s_m0[j], bb = s_m1[j];
                bool hit = false;
#if SYLPH_REPLAY_TAGS
                // Round 5: the scan over the k-mer's earlier occurrences reads ONE 32-bit word per occurrence — two 15-bit tags of its
                // markers, 0 for an occurrence that put nothing into the set (skipped mate 2, no markers) — and looks at the 16 bytes of
                // markers only where a tag matches (a real duplicate, or 4 x 2^-15 by chance).  Before: flag byte + record id + both
                // markers (25 bytes of LDS, four 64-bit compares) per earlier occurrence: the loop was a third of this kernel for a
                // community with 30x genomes in it.  ("A processed occurrence precedes j" is "j is not the head": the head of a
                // segment is never a skipped mate 2.)
                const uint32_t ta = marker_tag(a) * 0x00010001u, tb = marker_tag(bb) * 0x00010001u;
                for (uint32_t q = s_seg[j]; q < j; q++) {
                    const uint32_t w = s_tag[q], za = w ^ ta, zb = w ^ tb;
                    if ((((za - 0x00010001u) & ~za) | ((zb - 0x00010001u) & ~zb)) & 0x80008000u) {     // a zero halfword in either
                        const uint64_t x = s_m0[q], y = s_m1[q];
                        if (w && (x == a || y == a || x == bb || y == bb)) { hit = true; break; }
                    }
                }
                if (j != (uint32_t)s_seg[j] && (hit || a == bb)) fl |= 2;
#else
                bool any_prev = false;
                for (uint32_t q = s_seg[j]; q < j; q++) {
                    if (s_fl[q] & 1) continue;
                    any_prev = true;
                    if (s_rid[q] & RID_MARKER_BIT) {
                        const uint64_t x = s_m0[q], y = s_m1[q];
                        if (x == a || y == a || x == bb || y == bb) { hit = true; break; }
                    }
                }
                if (any_prev && (hit || a == bb)) fl |= 2;
#endif
            }
            const bool u = !(fl & 1) && (no_dedup || !(fl & 2));
            if (u) { my_u++; ubits |= (uint8_t)(1u << t); }
            s_fl[j] = fl;   // NB: later lanes only read bit0 of earlier positions, which does not change here
        }
    }
    if (dbg_stage == 3) { if (tid == 0) n_distinct[b] = 0; return; }
    // ---- P_i = would-be-counted occurrences before i in its k-mer; counted_i (cut-off rule, sketch.rs:706) ------
    // two block scans in total: (would-count, heads) packed 16+16 bits here, (counted, removed) below; sums <= CAP
    uint32_t tot_uh = 0;
    const uint32_t base_uh = block_excl_sum<RTPB>(my_u | (heads << 16), s_wave, &tot_uh);
    const uint32_t base_u = base_uh & 0xFFFFu, base_h = base_uh >> 16, total_heads = tot_uh >> 16;
    {
        uint32_t run = base_u;
        for (uint32_t t = 0; t < items; t++) {
            const uint32_t j = j0 + t;
            if (j >= n) break;
            s_a[j] = (uint16_t)run;             // Eu[j]
            if (ubits & (1u << t)) run++;
        }
    }
    __syncthreads();
    uint32_t my_c = 0, my_removed = 0;
    uint8_t cbits = 0;
    for (uint32_t t = 0; t < items; t++) {
        const uint32_t j = j0 + t;
        if (j >= n) break;
        const uint8_t fl = s_fl[j];
        if (fl & 1) continue;
        const uint32_t P = (uint32_t)s_a[j] - (uint32_t)s_a[s_seg[j]];
        const bool u = (ubits >> t) & 1;
        const bool c = (cutoff && P >= cutoff) ? true : u;
        if (c) { my_c++; cbits |= (uint8_t)(1u << t); } else my_removed++;
    }
    uint32_t tot_cr = 0;
    const uint32_t base_c = block_excl_sum<RTPB>(my_c | (my_removed << 16), s_wave, &tot_cr) & 0xFFFFu;
    const uint32_t total_removed = tot_cr >> 16;
    {
        uint32_t rc = base_c;
        for (uint32_t t = 0; t < items; t++) {
            const uint32_t j = j0 + t;
            if (j >= n) break;
            s_b[j] = (uint16_t)rc;              // Ec[j]
            if (cbits & (1u << t)) rc++;
        }
        if (j0 < n && j0 + items >= n) s_b[n] = (uint16_t)rc;   // Ec[n], written by the lane that owns the last position
    }
    __syncthreads();
    // heads emit (k-mer, count); the distinct index of a head = number of heads before it
    if constexpr (HASHED) {
        // segment end = position of the next head: the heads publish their positions by distinct index (s_a is free by now)
        uint16_t* const s_headpos = s_a;
        {
            uint32_t rh = base_h;
            for (uint32_t t = 0; t < items; t++) {
                const uint32_t j = j0 + t;
                if (j >= n) break;
                if (headbits & (1u << t)) s_headpos[rh++] = (uint16_t)j;
            }
            if (tid == 0) s_headpos[total_heads] = (uint16_t)n;
        }
        __syncthreads();
        uint32_t rh = base_h;
        for (uint32_t t = 0; t < items; t++) {
            const uint32_t j = j0 + t;
            if (j >= n) break;
            if (headbits & (1u << t)) {
                const uint32_t e = s_headpos[rh + 1];
                tmp_k[first + rh] = s_hash[j];
                tmp_c[first + rh] = (uint32_t)s_b[e] - (uint32_t)s_b[j];
                rh++;
            }
        }
    } else {
        uint32_t rh = base_h;
        const uint32_t out0 = first;
        for (uint32_t t = 0; t < items; t++) {
            const uint32_t j = j0 + t;
            if (j >= n) break;
            if (headbits & (1u << t)) {
                // segment end = next head or n: walk (segments here are shorter than SEG_LIMIT)
                uint32_t e = j + 1;
                while (e < n && s_seg[e] == j) e++;
                tmp_k[out0 + rh] = s_hash[j];
                tmp_c[out0 + rh] = (uint32_t)s_b[e] - (uint32_t)s_b[j];
                rh++;
            }
        }
    }
    // (per-bucket removed counts are summed by a separate kernel: one atomic per workgroup on a single word runs at ~88
    //  atomics/us on this chip and was bounding the whole kernel at ~0.2 ms for 2e4 buckets)
    if (tid == 0) { n_distinct[b] = total_heads; removed_b[b] = total_removed; }
}

template <int CAP, int RTPB>
__global__ __launch_bounds__(RTPB) void bucket_replay_kernel(const OccRec* __restrict__ recs, const uint32_t* __restrict__ perm,
                                                             const uint32_t* __restrict__ boff, const uint32_t* __restrict__ p_nv,
                                                             int paired, int no_dedup, uint32_t cutoff, BucketMap bm,
                                                             uint64_t* __restrict__ tmp_k, uint32_t* __restrict__ tmp_c,
                                                             uint32_t* __restrict__ n_distinct, uint32_t* __restrict__ removed_b,
                                                             uint32_t* __restrict__ overflow, uint32_t* __restrict__ mid_list,
                                                             uint32_t* __restrict__ large_list, uint32_t* __restrict__ ovf_list,
                                                             int dbg_stage) {
    replay_bucket<CAP, RTPB>(blockIdx.x, recs, perm, boff, p_nv, paired, no_dedup, cutoff, bm, tmp_k, tmp_c, n_distinct, removed_b,
                             overflow, mid_list, large_list, ovf_list, dbg_stage);
}

// Marker-less samples (single-end; long reads or --no-dedup: sylph_sketch::n_plain): nothing is ever dropped, the table is the
// histogram of the hashes.  Same bucket, same sub-range sort as replay_bucket, on 8-byte hashes gathered through the permutation
// instead of 32-byte records (the gather is what bounds the replay: 65 B fetched per occurrence there); no records exist at all.
// Buckets above CAP go to the next configuration's list (large_list), above that to ovf_list — the host writes the records of
// the sample then (OccRec{hash, 0, 0, 0}) and sends those buckets the usual way.
template <int CAP, int RTPB>
__device__ __forceinline__ void count_bucket(const uint32_t b, const uint64_t* __restrict__ hash, const uint32_t* __restrict__ perm,
                                             const uint32_t* __restrict__ boff, const uint32_t* __restrict__ p_nv, BucketMap bm,
                                             uint64_t* __restrict__ tmp_k, uint32_t* __restrict__ tmp_c, uint32_t* __restrict__ n_distinct,
                                             uint32_t* __restrict__ removed_b, uint32_t* __restrict__ overflow,
                                             uint32_t* __restrict__ large_list, uint32_t* __restrict__ ovf_list) {
    constexpr int ITEMS = CAP / RTPB;
    __shared__ uint64_t s_key[CAP], s_sorted[CAP];
    __shared__ uint32_t s_cnt[CAP + 1], s_mult[CAP];
    __shared__ __attribute__((aligned(8))) uint16_t s_fill[CAP];
    __shared__ uint32_t s_wave[RTPB / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t nv = *p_nv;
    const uint32_t first = boff[b], last = boff[b + 1];
    const uint32_t n = last - first;
    if (last > nv || first > last) { if (tid == 0) atomicAdd(overflow, 1u); return; }
    if (n == 0) return;
    if (n > (uint32_t)CAP) {
        if (tid == 0) {
            uint32_t* list = (CAP < CAP_LARGE && n <= (uint32_t)CAP_LARGE) ? large_list : ovf_list;
            list[1 + atomicAdd(&list[0], 1u)] = b;
        }
        return;
    }
    const uint64_t lo_hash = ((((uint64_t)b << 32) + bm.mult - 1u) / bm.mult) << bm.sh;      // (bm.composite: checked by the host)
    const uint32_t sub_mult = bm.sub_mult[CAP == CAP_SMALL ? 0 : CAP == CAP_MID ? 1 : 2];
    uint64_t h[ITEMS];
    uint32_t sub[ITEMS], place[ITEMS];
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RTPB;
        h[q] = i < n ? (perm ? hash[perm[first + i]] : hash[first + i]) : 0ull;      // perm == nullptr: `hash` is sorted by bucket already
    }
    for (uint32_t t = tid; t <= (uint32_t)CAP; t += RTPB) s_cnt[t] = 0;
    for (uint32_t t = tid; t < (uint32_t)CAP; t += RTPB) s_fill[t] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RTPB;
        sub[q] = 0;
        if (i < n) {
            const uint32_t hsres = (uint32_t)((h[q] - lo_hash) >> bm.sh);
            sub[q] = sub_mult ? min(__umulhi(hsres, sub_mult), (uint32_t)CAP - 1u) : min(hsres, (uint32_t)CAP - 1u);
            atomicAdd(&s_cnt[sub[q]], 1u);
        }
    }
    __syncthreads();
    {
        uint32_t v[ITEMS], sum = 0;
#pragma unroll
        for (int e = 0; e < ITEMS; e++) { v[e] = s_cnt[tid * ITEMS + e]; sum += v[e]; }
        uint32_t run = block_excl_sum<RTPB>(sum, s_wave, nullptr);
#pragma unroll
        for (int e = 0; e < ITEMS; e++) { s_cnt[tid * ITEMS + e] = run; run += v[e]; }
        if (tid == RTPB - 1) s_cnt[CAP] = run;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RTPB;
        place[q] = 0;
        if (i < n) {
            uint32_t* const w = reinterpret_cast<uint32_t*>(s_fill) + (sub[q] >> 1);
            const uint32_t old = atomicAdd(w, (sub[q] & 1u) ? 65536u : 1u);
            place[q] = s_cnt[sub[q]] + ((sub[q] & 1u) ? (old >> 16) : (old & 0xFFFFu));
            s_key[place[q]] = h[q];
        }
    }
    __syncthreads();
    // sorted position = start of the sub-range + smaller hashes in it + equal hashes placed before; the first of its equals
    // carries the k-mer's multiplicity
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RTPB;
        if (i < n) {
            const uint32_t lo = s_cnt[sub[q]], hi = s_cnt[sub[q] + 1];
            uint32_t less = 0, eq = 0, eq_before = 0;
            for (uint32_t p = lo; p < hi; p++) {
                const uint64_t kp = s_key[p];
                less += kp < h[q] ? 1u : 0u;
                const uint32_t same = kp == h[q] ? 1u : 0u;
                eq += same;
                eq_before += (same && p < place[q]) ? 1u : 0u;
            }
            const uint32_t r = lo + less + eq_before;
            s_sorted[r] = h[q];
            s_mult[r] = eq_before == 0 ? eq : 0u;
        }
    }
    __syncthreads();
    const uint32_t items = (n + RTPB - 1) / RTPB, j0 = tid * items;
    uint32_t heads = 0;
    for (uint32_t t = 0; t < items; t++) {
        const uint32_t j = j0 + t;
        if (j >= n) break;
        heads += s_mult[j] ? 1u : 0u;
    }
    uint32_t total_heads = 0;
    uint32_t rh = block_excl_sum<RTPB>(heads, s_wave, &total_heads);
    for (uint32_t t = 0; t < items; t++) {
        const uint32_t j = j0 + t;
        if (j >= n) break;
        const uint32_t m = s_mult[j];
        if (m) { tmp_k[first + rh] = s_sorted[j]; tmp_c[first + rh] = m; rh++; }
    }
    if (tid == 0) { n_distinct[b] = total_heads; removed_b[b] = 0; }
}
template <int CAP, int RTPB>
__global__ __launch_bounds__(RTPB) void bucket_count_kernel(const uint64_t* __restrict__ hash, const uint32_t* __restrict__ perm,
                                                            const uint32_t* __restrict__ boff, const uint32_t* __restrict__ p_nv, BucketMap bm,
                                                            uint64_t* __restrict__ tmp_k, uint32_t* __restrict__ tmp_c,
                                                            uint32_t* __restrict__ n_distinct, uint32_t* __restrict__ removed_b,
                                                            uint32_t* __restrict__ overflow, uint32_t* __restrict__ large_list,
                                                            uint32_t* __restrict__ ovf_list) {
    count_bucket<CAP, RTPB>(blockIdx.x, hash, perm, boff, p_nv, bm, tmp_k, tmp_c, n_distinct, removed_b, overflow, large_list, ovf_list);
}
template <int CAP, int RTPB>
__global__ __launch_bounds__(RTPB) void bucket_count_list_kernel(const uint64_t* __restrict__ hash, const uint32_t* __restrict__ perm,
                                                                 const uint32_t* __restrict__ boff, const uint32_t* __restrict__ p_nv,
                                                                 BucketMap bm, uint64_t* __restrict__ tmp_k, uint32_t* __restrict__ tmp_c,
                                                                 uint32_t* __restrict__ n_distinct, uint32_t* __restrict__ removed_b,
                                                                 uint32_t* __restrict__ overflow, const uint32_t* __restrict__ my_list,
                                                                 uint32_t* __restrict__ ovf_list) {
    const uint32_t n_listed = my_list[0];
    for (uint32_t i = blockIdx.x; i < n_listed; i += gridDim.x) {
        count_bucket<CAP, RTPB>(my_list[1 + i], hash, perm, boff, p_nv, bm, tmp_k, tmp_c, n_distinct, removed_b, overflow, nullptr, ovf_list);
        __syncthreads();
    }
}

// second configuration: a fixed, small grid walks the (usually empty) list of buckets the first one queued
template <int CAP, int RTPB>
__global__ __launch_bounds__(RTPB) void bucket_replay_list_kernel(const OccRec* __restrict__ recs, const uint32_t* __restrict__ perm,
                                                                  const uint32_t* __restrict__ boff, const uint32_t* __restrict__ p_nv,
                                                                  int paired, int no_dedup, uint32_t cutoff, BucketMap bm,
                                                                  uint64_t* __restrict__ tmp_k, uint32_t* __restrict__ tmp_c,
                                                                  uint32_t* __restrict__ n_distinct, uint32_t* __restrict__ removed_b,
                                                                  uint32_t* __restrict__ overflow, const uint32_t* __restrict__ my_list,
                                                                  uint32_t* __restrict__ large_list, uint32_t* __restrict__ ovf_list,
                                                                  int dbg_stage) {
    const uint32_t n_listed = my_list[0];
    for (uint32_t i = blockIdx.x; i < n_listed; i += gridDim.x) {
        replay_bucket<CAP, RTPB>(my_list[1 + i], recs, perm, boff, p_nv, paired, no_dedup, cutoff, bm, tmp_k, tmp_c, n_distinct,
                                 removed_b, overflow, nullptr, large_list, ovf_list, dbg_stage);
        __syncthreads();   // the LDS arrays are reused by the next bucket
    }
}

// Table offsets in two levels, no library scan: workgroup w scans the 1024 buckets of its chunk — d_loc[b] = table rows of the
// chunk's buckets before b — and leaves the chunk's row total and removed total; the compaction kernel adds the chunk totals
// before w (at most 256 of them: B <= 2^18 on this path).
constexpr uint32_t SCAN_CHUNK = 1024;
__global__ __launch_bounds__(SCAN_CHUNK) void table_scan_kernel(const uint32_t* __restrict__ n_distinct, const uint32_t* __restrict__ removed_b,
                                                                uint32_t B, uint32_t ipt, uint32_t* __restrict__ d_loc,
                                                                uint32_t* __restrict__ chunk_rows, unsigned long long* __restrict__ chunk_removed) {
    // a workgroup scans a chunk of SCAN_CHUNK * ipt buckets, every lane `ipt` consecutive ones (ipt = 1 up to 2^18 buckets; a
    // long-read sample at c = 100 has 4e5: at most 256 chunks whatever B, so that the second level fits one workgroup's LDS)
    __shared__ uint32_t s_wave[SCAN_CHUNK / 64];
    __shared__ unsigned long long s_rem[SCAN_CHUNK / 64];
    const uint32_t b0 = (blockIdx.x * SCAN_CHUNK + threadIdx.x) * ipt, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v = 0;
    unsigned long long rem = 0;
    for (uint32_t i = 0; i < ipt && b0 + i < B; i++) { v += n_distinct[b0 + i]; rem += removed_b[b0 + i]; }
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) rem += __shfl_xor(rem, d);
    if (lane == 63) s_wave[wave] = x;
    if (lane == 0) s_rem[wave] = rem;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < SCAN_CHUNK / 64; w++) { const uint32_t t = s_wave[w]; if (w < wave) base += t; tot += t; }
    uint32_t run = base + x - v;
    for (uint32_t i = 0; i < ipt && b0 + i < B; i++) { d_loc[b0 + i] = run; run += n_distinct[b0 + i]; }
    if (threadIdx.x == 0) {
        unsigned long long r = 0;
        for (uint32_t w = 0; w < SCAN_CHUNK / 64; w++) r += s_rem[w];
        chunk_rows[blockIdx.x] = tot;
        chunk_removed[blockIdx.x] = r;
    }
}
// out[rows before bucket b + i] = tmp[boff[b] + i] for i < n_distinct[b]; a workgroup walks buckets with its four wavefronts
// (one bucket holds ~50 rows).  Also assembles the 48-byte tail block the host reads: {removed u64, overflow u32 (set by the
// replay), n_seg u32, n_ovf u32, n_mid u32, n_large u32, seeding verdict 2 x u32}.
__global__ __launch_bounds__(256) void table_compact_kernel(const uint64_t* __restrict__ tmp_k, const uint32_t* __restrict__ tmp_c,
                                                            const uint32_t* __restrict__ boff, const uint32_t* __restrict__ d_loc,
                                                            const uint32_t* __restrict__ chunk_rows, const unsigned long long* __restrict__ chunk_removed,
                                                            const uint32_t* __restrict__ n_distinct, uint32_t B, uint32_t ipt,
                                                            uint64_t* __restrict__ out_k, uint32_t* __restrict__ out_c,
                                                            const uint32_t* __restrict__ ovf_list, const uint32_t* __restrict__ mid_list,
                                                            const uint32_t* __restrict__ large_list, int skip_if_listed,
                                                            uint32_t* __restrict__ tail, const uint32_t* __restrict__ verdict,
                                                            const uint32_t* __restrict__ a10_words, const uint32_t* __restrict__ p_nv) {
    __shared__ uint32_t s_base[257];
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_rem[4];
    const uint32_t chunk = SCAN_CHUNK * ipt, n_chunks = (B + chunk - 1) / chunk;        // <= 256: one chunk total per lane
    {
        const uint32_t v = threadIdx.x < n_chunks ? chunk_rows[threadIdx.x] : 0u;
        uint32_t tot = 0;
        const uint32_t ex = block_excl_sum<256>(v, s_wave, &tot);
        if (threadIdx.x < n_chunks) s_base[threadIdx.x] = ex;
        if (threadIdx.x == 0) s_base[n_chunks] = tot;
        if (blockIdx.x == 0) {                                                       // the tail block, once
            unsigned long long rem = threadIdx.x < n_chunks ? chunk_removed[threadIdx.x] : 0ull;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) rem += __shfl_xor(rem, d);
            if ((threadIdx.x & 63) == 0) s_rem[threadIdx.x >> 6] = rem;
            __syncthreads();
            if (threadIdx.x == 0) {
                *reinterpret_cast<unsigned long long*>(tail) = s_rem[0] + s_rem[1] + s_rem[2] + s_rem[3];
                tail[3] = tot; tail[4] = ovf_list[0]; tail[5] = mid_list[0]; tail[6] = large_list[0];
                // deferred seeding verdict (reads.hip ReadsState: long_record, overflowing blocks) rides in the same block: one copy
                tail[7] = verdict ? verdict[0] : 0u; tail[8] = verdict ? verdict[1] : 0u;
                // ... and so do the verdict words of the filter dedup's partitioned pass (a10.hip)
                tail[9] = a10_words ? a10_words[0] : 0u; tail[10] = a10_words ? a10_words[1] : 0u;
                tail[11] = *p_nv;            // the occurrences the partition found: what a deferred batch's slots really hold
            }
        }
    }
    __syncthreads();
    // buckets are still waiting for the list-driven configurations: the host will come back after running them (a long-read
    // table has tens of millions of rows: copying it twice would cost more than the configurations themselves)
    if (skip_if_listed && (mid_list[0] | large_list[0])) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        const uint32_t n = n_distinct[b], s0 = boff[b], d = s_base[b / chunk] + d_loc[b];
        for (uint32_t i = lane; i < n; i += 64) { out_k[d + i] = tmp_k[s0 + i]; out_c[d + i] = tmp_c[s0 + i]; }
    }
}
// ---- buckets beyond the large configuration: their occurrences go through the device-wide path as one small sample --------
// sub_off[i] = occurrences of the listed buckets before bucket i (single workgroup; the list is short); sub_off[m] = total
__global__ __launch_bounds__(1024) void ovf_offsets_kernel(const uint32_t* __restrict__ ovf_list, const uint32_t* __restrict__ boff,
                                                           uint32_t* __restrict__ sub_off) {
    __shared__ uint32_t s_part[1024];
    const uint32_t m = ovf_list[0], tid = threadIdx.x;
    const uint32_t per = (m + 1023) / 1024;
    uint32_t sum = 0;
    for (uint32_t i = tid * per; i < min(m, (tid + 1) * per); i++) { const uint32_t b = ovf_list[1 + i]; sum += boff[b + 1] - boff[b]; }
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t t = 0; t < 1024; t++) { const uint32_t v = s_part[t]; s_part[t] = run; run += v; }
        sub_off[m] = run;
    }
    __syncthreads();
    uint32_t run = s_part[tid];
    for (uint32_t i = tid * per; i < min(m, (tid + 1) * per); i++) {
        const uint32_t b = ovf_list[1 + i];
        sub_off[i] = run;
        run += boff[b + 1] - boff[b];
    }
}

// the occurrence indices of listed bucket i -> sub_idx[sub_off[i] ..): sorted afterwards, they are the listed buckets'
// occurrences in FILE order (what the device-wide path expects of its input)
__global__ __launch_bounds__(256) void ovf_indices_kernel(const uint32_t* __restrict__ ovf_list, const uint32_t* __restrict__ sub_off,
                                                          const uint32_t* __restrict__ boff, const uint32_t* __restrict__ perm,
                                                          uint32_t* __restrict__ sub_idx) {
    const uint32_t b = ovf_list[1 + blockIdx.x], first = boff[b], n = boff[b + 1] - first, o = sub_off[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) sub_idx[o + j] = perm[first + j];
}
__global__ __launch_bounds__(256) void ovf_gather_kernel(const uint32_t* __restrict__ sorted_idx, uint32_t n, const OccRec* __restrict__ recs,
                                                         uint64_t* __restrict__ sub_hash, OccRec* __restrict__ sub_recs) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const OccRec r = recs[sorted_idx[j]];
    sub_hash[j] = r.hash;
    sub_recs[j] = r;
}

// the (k-mer, count) rows the device-wide path produced for listed bucket i go to the bucket's slots of the temporary table
__global__ __launch_bounds__(256) void ovf_patch_kernel(const uint32_t* __restrict__ ovf_list, BucketMap bm,
                                                        const uint64_t* __restrict__ sub_k, const uint32_t* __restrict__ sub_c,
                                                        uint32_t n_sub_out, const uint32_t* __restrict__ boff,
                                                        uint64_t* __restrict__ tmp_k, uint32_t* __restrict__ tmp_c,
                                                        uint32_t* __restrict__ n_distinct) {
    const uint32_t b = ovf_list[1 + blockIdx.x];
    auto lower = [&](uint64_t key) {   // first row with k-mer >= key
        uint32_t lo = 0, hi = n_sub_out;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (sub_k[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    // rows of bucket b = rows with k-mer in [smallest hash of bucket b, smallest hash of bucket b + 1) (inverse of bucket_of)
    __shared__ uint32_t s_lo, s_hi;
    if (threadIdx.x == 0) {
        auto lo_hash = [&](uint32_t bb) { return ((((uint64_t)bb << 32) + bm.mult - 1u) / bm.mult) << bm.sh; };
        s_lo = lower(lo_hash(b));
        s_hi = (b + 1 < bm.B) ? lower(lo_hash(b + 1)) : n_sub_out;
    }
    __syncthreads();
    const uint32_t lo = s_lo, n = s_hi - s_lo, d = boff[b];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { tmp_k[d + i] = sub_k[lo + i]; tmp_c[d + i] = sub_c[lo + i]; }
    if (threadIdx.x == 0) n_distinct[b] = n;
}

uint32_t grid_of(uint64_t n, uint32_t tpb = 256) { return (uint32_t)((n + tpb - 1) / tpb); }

}  // namespace

bool finish_bucketed(sylph_sketch* sk) {
    sylph_ctx* ctx = sk->ctx;
    const bool slotted = sk->pend.live;                // the sample's one batch, still in its slots (reads.hip)
    // deferred verdict (sketch_session.h PendingSlots): the number of occurrences is not known on the host — geometry from the
    // expectation, capacities from the upper bound, every count the kernels need from device memory, the verdict read with the tail
    const bool deferred = slotted && sk->pend.deferred;
    const uint32_t n_cap = slotted ? sk->pend.n : (uint32_t)sk->n_occ;      // what the arrays must hold
    const uint32_t n_all = deferred ? std::max<uint32_t>(1, sk->pend.n_expect) : n_cap;
    sk->n_out = 0;
    sk->dup_removed = 0;
    if (n_cap == 0) return true;
    if (sk->c < 2) return false;   // c = 1: valid hashes reach the top bit that marks invalid occurrences
    // bucket geometry: B = n / TARGET equal hash ranges (see BucketMap)
    const uint64_t thr = UINT64_MAX / (uint64_t)sk->c;
    const uint32_t TARGET = ctx->bucket_target;
    const uint32_t B = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, n_all / TARGET), 1u << 24);
    BucketMap bm;
    bm.sh = key_shift(sk->c);
    const uint64_t hs_max = thr >> bm.sh;                              // hashes are < thr
    bm.mult = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, ((uint64_t)B << 32) / (hs_max + 1));
    bm.B = B;
    // widest bucket in hs units is ceil(2^32 / mult) + 1; the key needs (range << sh) to fit in 64 - IDX_BITS bits
    const uint64_t range_hs = (0x100000000ull + bm.mult - 1) / std::max<uint32_t>(1, bm.mult) + 1;
    bm.composite = bm.mult >= 1 && bit_length(range_hs) + bm.sh <= 64 - IDX_BITS;
    bm.range_hs = (uint32_t)std::min<uint64_t>(range_hs, 0xFFFFFFFFull);
    {
        const uint32_t caps[3] = {(uint32_t)CAP_SMALL, (uint32_t)CAP_MID, (uint32_t)CAP_LARGE};
        for (int i = 0; i < 3; i++) bm.sub_mult[i] = range_hs > caps[i] ? (uint32_t)(((uint64_t)caps[i] << 32) / range_hs) : 0u;
        // one-word ranking keys: sub-range s of configuration i (sub = floor(hs * sub_mult / 2^32), sub_mult rounded down) holds
        // hs values from s * width on (width = floor(range_hs / caps[i]) <= 2^32 / sub_mult) and below (s + 1) * 2^32 / sub_mult;
        // the distance between the two grows with s: the last sub-range gives the span every residue stays below
        const uint64_t max_index = slotted ? (uint64_t)sk->pend.n_blk * sk->pend.slot_cap : (uint64_t)n_all;
        const int index_bits = bit_length(max_index | 1);
        for (int i = 0; i < 3; i++) {
            bm.sub_width[i] = bm.sub_mult[i] ? (uint32_t)(range_hs / caps[i]) : 1u;
            uint64_t span_hs = 1;
            if (bm.sub_mult[i]) {
                const uint64_t top = std::min<uint64_t>(range_hs, (((uint64_t)caps[i] << 32) + bm.sub_mult[i] - 1) / bm.sub_mult[i]);
                span_hs = top - (uint64_t)(caps[i] - 1) * bm.sub_width[i] + 1;
            }
            bm.rank_bits[i] = (bm.composite && bit_length(span_hs) + bm.sh + index_bits <= 64) ? index_bits : 0;
        }
    }
    // partition geometry: F = 2^fine_bits buckets per coarse range (about 512 ranges), tiles of occurrences
    const PartGeom geom = part_geometry(B);
    // marker-less sample (sketch_session.h): hashes only, counted without occurrence records; tiny samples whose bucket range does
    // not fit the sub-range arithmetic get their records written and take the usual kernels
    if (!slotted && sk->n_plain && !(sk->n_plain == sk->n_occ && bm.composite)) materialise_plain_records(sk);
    bool plain = !slotted && sk->n_plain != 0;
    PartIn in{};
    in.slotted = slotted ? 1 : 0;
    in.key_sh = bm.sh;
    uint32_t n_tiles;
    const OccRec* recs;
    if (slotted) {
        in.slot_key = sk->slot_key.as<uint32_t>();
        in.n_blk = sk->pend.n_blk;
        in.slot_cap = sk->pend.slot_cap;
        in.blk_count = sk->slot_meta.as<uint32_t>() + (sk->pend.n_blk + 1);   // (layout: reads.hip SlotMeta)
        in.blk_per_tile = BLK_PER_TILE;
        n_tiles = (in.n_blk + BLK_PER_TILE - 1) / BLK_PER_TILE;
        recs = sk->slot_rec.as<OccRec>();
    } else {
        in.hash = sk->hash.as<uint64_t>();
        in.n_dense = n_all;
        in.tile_entries = (uint32_t)std::max<uint64_t>(4096, ((uint64_t)n_all + 65535) / 65536);   // at most 65,536 tiles
        n_tiles = (uint32_t)(((uint64_t)n_all + in.tile_entries - 1) / in.tile_entries);
        recs = sk->recs.as<OccRec>();
    }
    DevBuf &b_hist = ctx->scratch[0], &b_pairs = ctx->scratch[1], &b_perm = ctx->scratch[2], &b_tmpk = ctx->scratch[3],
           &b_tmpc = ctx->scratch[4], &b_small = ctx->scratch[5], &b_bk = ctx->scratch[6];
    b_hist.reserve(part_hist_words(geom, n_tiles) * 4);
    b_pairs.reserve((size_t)n_cap * 8);                                 // (bucket, occurrence index) pairs grouped by coarse range
    if (!plain) b_perm.reserve((size_t)n_cap * 4);
    DevBuf& b_sorted = ctx->scratch[7];                                 // marker-less: the hashes sorted by bucket
    if (plain) b_sorted.reserve((size_t)n_cap * 8);
    uint64_t* sorted_hash = plain ? b_sorted.as<uint64_t>() : nullptr;
    in.carry = plain ? 1 : 0;
    b_tmpk.reserve((size_t)n_cap * 8);
    b_tmpc.reserve((size_t)n_cap * 4);
    b_small.reserve(64);
    // boff | large_list | ovf_list | n_distinct | removed | d_off | mid_list (each B+2) | chunk_rows (264) | chunk_removed (264 u64)
    b_bk.reserve((size_t)(B + 2) * 4 * 7 + 264 * 4 + 264 * 8 + 16);
    uint32_t* hist = b_hist.as<uint32_t>();
    uint2* pairs = b_pairs.as<uint2>();
    uint32_t* boff = b_bk.as<uint32_t>();
    uint32_t* large_list = boff + (B + 2);      // [0] = number of buckets queued for the large configuration, [1..] = ids
    uint32_t* ovf_list = large_list + (B + 2);  // [0] = number of buckets beyond the large configuration, [1..] = ids
    uint32_t* n_distinct = ovf_list + (B + 2);
    uint32_t* removed_b = n_distinct + (B + 2);
    uint32_t* d_off = removed_b + (B + 2);
    uint32_t* mid_list = d_off + (B + 2);       // buckets for the medium configuration (n <= CAP_MID, or a long k-mer segment)
    uint32_t* chunk_rows = mid_list + (B + 2) + ((B & 1u) ? 1 : 0);       // (8-byte aligned: 7 * (B + 2) words is odd for odd B)
    unsigned long long* chunk_removed = reinterpret_cast<unsigned long long*>(chunk_rows + 264);
    unsigned long long* d_removed = b_small.as<unsigned long long>();
    uint32_t* d_overflow = reinterpret_cast<uint32_t*>(b_small.as<uint8_t>() + 8);
    const uint32_t* d_nv = boff + B;            // boff[B] = number of valid occurrences
    const uint32_t n_zero = (B + 2) * 2;                                            // n_distinct and removed (cleared by part_hist_kernel)
    // every launch below takes its sizes from device memory; the host synchronises ONCE, at the end (unless some buckets need
    // the list-driven configurations or the device-wide path)
    const int dbg = getenv("SYLPH_REPLAY_STAGE") ? atoi(getenv("SYLPH_REPLAY_STAGE")) : 0;
    const uint32_t cutoff = sk->paired ? 0u : 4u;   // MAX_DEDUP_COUNT, constants.rs:14
    sk->out_k.reserve((size_t)n_cap * 8);          // upper bound: distinct k-mers <= occurrences
    sk->out_c.reserve((size_t)n_cap * 4);
    {
        HostPhase ph(ctx, "finish(bucket): partition + LDS replay + compact");
        {
            ScopedKernelTimer t(ctx, "sort");   // the partition: what the library's radix sort of (bucket, index) pairs used to do
            launch_partition(ctx, in, bm, geom, n_tiles, n_all, hist, pairs, boff, b_perm.as<uint32_t>(), sorted_hash, n_distinct, n_zero,
                             b_small.as<uint32_t>(), large_list, ovf_list, mid_list);
        }
        {
            ScopedKernelTimer t(ctx, "replay");
            if (plain)
                hipLaunchKernelGGL((bucket_count_kernel<CAP_SMALL, RTPB_SMALL>), dim3(B), dim3(RTPB_SMALL), 0, ctx->stream,
                                   sorted_hash, (const uint32_t*)nullptr, boff, d_nv, bm, b_tmpk.as<uint64_t>(),
                                   b_tmpc.as<uint32_t>(), n_distinct, removed_b, d_overflow, large_list, ovf_list);
            else
                hipLaunchKernelGGL((bucket_replay_kernel<CAP_SMALL, RTPB_SMALL>), dim3(B), dim3(RTPB_SMALL), 0, ctx->stream,
                                   recs, b_perm.as<uint32_t>(), boff, d_nv, sk->paired, sk->dedup_mode(), cutoff, bm,
                                   b_tmpk.as<uint64_t>(), b_tmpc.as<uint32_t>(), n_distinct, removed_b, d_overflow, mid_list, large_list,
                                   ovf_list, dbg);
        }
    }
    // removed counts, table offsets, compaction, and everything the host needs to know in one 28-byte block
    auto close_table = [&](int skip_if_listed) {
        // two levels: chunks of 1024 x ipt buckets (at most 256 of them), then the compaction, which scans the chunk totals itself
        const uint32_t ipt = (B + (1u << 18) - 1) >> 18;
        ScopedKernelTimer t(ctx, "replay");
        hipLaunchKernelGGL(table_scan_kernel, dim3((B + SCAN_CHUNK * ipt - 1) / (SCAN_CHUNK * ipt)), dim3(SCAN_CHUNK), 0, ctx->stream, n_distinct,
                           removed_b, B, ipt, d_off, chunk_rows, chunk_removed);
        hipLaunchKernelGGL(table_compact_kernel, dim3(std::min<uint32_t>((B + 3) / 4, 1u << 15)), dim3(256), 0, ctx->stream,
                           b_tmpk.as<uint64_t>(), b_tmpc.as<uint32_t>(), boff, d_off, chunk_rows, chunk_removed, n_distinct, B, ipt,
                           sk->out_k.as<uint64_t>(), sk->out_c.as<uint32_t>(), ovf_list, mid_list, large_list, skip_if_listed,
                           b_small.as<uint32_t>(),
                           deferred ? sk->slot_meta.as<uint32_t>() + (size_t)(sk->pend.n_blk + 1) * 4 : (const uint32_t*)nullptr,
                           sk->a10_state == 1 ? sk->a10_tail.as<uint32_t>() : (const uint32_t*)nullptr, d_nv);
        SY_HIP(hipGetLastError());
    };
    struct { unsigned long long removed; uint32_t overflow, n_seg, n_ovf, n_mid, n_large; } host{};
    uint32_t verdict[2] = {0, 0};                  // deferred: long_record flag, overflowing blocks of the seeding kernel
    uint32_t a10_words[2] = {0, 0};                // filter dedup, partitioned pass: buckets it could not take, operations it found
    uint32_t n_found = 0;                          // occurrences the partition found
    auto read_tail = [&] {
        SY_HIP(hipMemcpyAsync(ctx->pinned, d_removed, 48, hipMemcpyDeviceToHost, ctx->stream));
        SY_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(&host, ctx->pinned, 28);
        if (deferred) memcpy(verdict, (const char*)ctx->pinned + 28, 8);   // (the two flag words of ReadsState, copied by table_compact_kernel)
        memcpy(a10_words, (const char*)ctx->pinned + 36, 8);
        memcpy(&n_found, (const char*)ctx->pinned + 44, 4);
        if (!ctx->pending.empty()) profile_collect(ctx);
    };
    close_table(1);
    read_tail();
    if (deferred) {
        if (verdict[0] || verdict[1]) {            // not a batch for the short-read kernel after all: the checked push, then from the top
            redo_deferred_batch(sk);
            a10_mark(sk);                          // (filter dedup: the marks went with the slots)
            return finish_bucketed(sk);
        }
        sk->pend.deferred = false;                 // the verdict is in: from here on an ordinary slotted sample ...
        sk->pend.n = n_found;                      // ... whose occurrence count is known (whoever flushes the slots to the dense arrays needs it)
    }
    // filter dedup: were the partitioned pass's marks good (a10.hip)?  If not the phase walk has marked the records again, dense: from the top
    if (!a10_verdict(sk, a10_words)) return finish_bucketed(sk);
    if (host.overflow) return false;             // inconsistent bounds (defensive): the generic path redoes the sample
    if (plain && host.n_ovf) {
        // k-mers more than a thousand deep in a marker-less sample: write the occurrence records after all and take the usual
        // kernels from the start (their overflow path works on records and on the index permutation)
        materialise_plain_records(sk);
        return finish_bucketed(sk);
    }
    if (host.n_mid || host.n_large) {
        // Buckets the 256-slot configuration passed on (more than 256 occurrences, or a k-mer 96+ deep: abundant genomes).  The
        // list-driven configurations are launched only now, with grids that match the lists: launched speculatively with every
        // sample they are two dispatches of large-LDS workgroups that find nothing to do, but cannot even START beside another
        // stream's seeding kernel (which leaves 10 KiB of LDS per CU) — in the pipelined bench they held the stream up for 0.3 ms.
        HostPhase ph(ctx, "finish(bucket): medium / large configurations");
        {
            ScopedKernelTimer t(ctx, "replay");
            if (plain) {
                if (host.n_large)
                    hipLaunchKernelGGL((bucket_count_list_kernel<CAP_LARGE, RTPB_LARGE>), dim3(std::min<uint32_t>(host.n_large, 1536u)),
                                       dim3(RTPB_LARGE), 0, ctx->stream, sorted_hash, (const uint32_t*)nullptr, boff, d_nv, bm,
                                       b_tmpk.as<uint64_t>(), b_tmpc.as<uint32_t>(), n_distinct, removed_b, d_overflow, large_list, ovf_list);
            } else {
            if (host.n_mid)
                hipLaunchKernelGGL((bucket_replay_list_kernel<CAP_MID, RTPB_MID>), dim3(std::min<uint32_t>(host.n_mid, 1280u)),
                                   dim3(RTPB_MID), 0, ctx->stream, recs, b_perm.as<uint32_t>(), boff, d_nv, sk->paired,
                                   sk->dedup_mode(), cutoff, bm, b_tmpk.as<uint64_t>(), b_tmpc.as<uint32_t>(), n_distinct, removed_b,
                                   d_overflow, mid_list, large_list, ovf_list, dbg);
            if (host.n_large)
                hipLaunchKernelGGL((bucket_replay_list_kernel<CAP_LARGE, RTPB_LARGE>), dim3(std::min<uint32_t>(host.n_large, 512u)),
                                   dim3(RTPB_LARGE), 0, ctx->stream, recs, b_perm.as<uint32_t>(), boff, d_nv, sk->paired,
                                   sk->dedup_mode(), cutoff, bm, b_tmpk.as<uint64_t>(), b_tmpc.as<uint32_t>(), n_distinct, removed_b,
                                   d_overflow, large_list, large_list, ovf_list, dbg);
            }
        }
        close_table(0);
        read_tail();
        if (host.overflow) return false;
    }
    unsigned long long removed_extra = 0;
    if (host.n_ovf) {
        // Some buckets exceed even the large configuration (k-mers with thousands of occurrences: low-complexity reads,
        // very abundant genomes).  Only THEIR occurrences go through the device-wide path, as one small sample; its rows
        // are patched into the buckets' slots and the table is compacted again.
        if (ctx->finish_mode == 2 || host.n_ovf > 4096) return false;
        HostPhase ph(ctx, "finish(bucket): overflowing buckets through the device-wide path");
        const uint32_t m = host.n_ovf;
        DevBuf b_so(ctx), b_si(ctx), b_sh(ctx), b_sr(ctx), sub_k(ctx), sub_c(ctx);
        b_so.reserve(((size_t)m + 1) * 4);
        hipLaunchKernelGGL(ovf_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, ovf_list, boff, b_so.as<uint32_t>());
        uint32_t n_sub = 0;
        ctx->read_back(&n_sub, b_so.as<uint32_t>() + m, 4);
        b_si.reserve((size_t)n_sub * 8);            // indices | sorted indices
        b_sh.reserve((size_t)n_sub * 8);
        b_sr.reserve((size_t)n_sub * sizeof(OccRec));
        uint32_t* sub_idx = b_si.as<uint32_t>();
        uint32_t* sub_sorted = sub_idx + n_sub;
        hipLaunchKernelGGL(ovf_indices_kernel, dim3(m), dim3(256), 0, ctx->stream, ovf_list, b_so.as<uint32_t>(), boff,
                           b_perm.as<uint32_t>(), sub_idx);
        sort_keys_u32(ctx, sub_idx, sub_sorted, n_sub, 0, 32);       // ascending index = file order
        {
            ScopedKernelTimer t(ctx, "replay_overflow");   // (family of its own so that tests can see this path was taken)
            hipLaunchKernelGGL(ovf_gather_kernel, dim3(grid_of(n_sub)), dim3(256), 0, ctx->stream, sub_sorted, n_sub, recs,
                               b_sh.as<uint64_t>(), b_sr.as<OccRec>());
        }
        uint64_t n_sub_out = 0, removed_sub = 0;
        generic_replay(ctx, b_sh.as<uint64_t>(), b_sr.as<OccRec>(), n_sub, sk->paired, sk->dedup_mode(), sub_k, sub_c, n_sub_out, removed_sub);
        removed_extra = removed_sub;
        {
            ScopedKernelTimer t(ctx, "replay");
            hipLaunchKernelGGL(ovf_patch_kernel, dim3(m), dim3(256), 0, ctx->stream, ovf_list, bm, sub_k.as<uint64_t>(),
                               sub_c.as<uint32_t>(), (uint32_t)n_sub_out, boff, b_tmpk.as<uint64_t>(), b_tmpc.as<uint32_t>(), n_distinct);
        }
        close_table(0);
        read_tail();
    }
    sk->n_out = host.n_seg;
    sk->dup_removed = host.removed + removed_extra;
    return true;
}

}  // namespace sylph
