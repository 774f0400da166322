// partition.h — the two-level MSD partition by hash range (round 3), shared by the replay (replay_lds.hip: occurrences by k-mer
// hash) and the filter dedup (a10.hip: operations by class hash).  Kernels only move 8-byte pairs; see replay_lds.hip's header.
#pragma once
#include "common.h"
#include "device_common.h"
#include "sketch_session.h"

namespace sylph {
namespace {

// Bucket of a hash: hs = hash >> sh (its 32 most significant bits below the threshold), b = (hs * mult) >> 32 — B
// equal ranges for ANY B (not only powers of two), monotone in the hash.  Inverse used by the replay kernel: the
// smallest hs of bucket b is ceil(b * 2^32 / mult).
// range_hs = widest bucket in hs units; sub_mult[i] = floor(2^32 * CAP_i / range_hs) for the three replay configurations (the
// sub-range of a hash inside its bucket, see replay_bucket), 0 when a bucket is narrower than CAP_i hs units.
// rank_bits[i] > 0: (hash - lowest hash the sub-range can hold) << rank_bits | gather index fits in 64 bits for configuration
// i — the key the occurrences of a sub-range are ranked by with ONE compare; sub_width[i] = floor(range_hs / CAP_i) hs units (a
// lower bound of where sub-range s begins: s * sub_width).
struct BucketMap { int sh; uint32_t mult; uint32_t B; int composite; uint32_t range_hs; uint32_t sub_mult[3]; uint32_t sub_width[3]; int rank_bits[3]; };

__device__ __forceinline__ uint32_t bucket_of_key(uint32_t key, const BucketMap m) { return min(__umulhi(key, m.mult), m.B - 1u); }

// ---- partition ------------------------------------------------------------------------------------------------------------
// Where finish() reads the occurrences from: the dense arrays (hash[i], i < n_dense; INVALID_HASH entries are skipped) or the
// slots of the session's one batch (slot_key[b * slot_cap + i], i < blk_count[b]; key = hash >> key_sh, written by the seeding
// kernel).  The index an occurrence is known by — what the replay gathers its record with — is i, resp. b * slot_cap + i: both
// grow with the file order.
struct PartIn {
    const uint64_t* hash;
    const uint32_t* slot_key;
    const uint32_t* blk_count;
    uint32_t n_dense, n_blk, slot_cap, tile_entries;
    uint32_t blk_per_tile;   // slotted: blocks of the seeding kernel per partition tile (BLK_PER_TILE; fewer for the two-word layout)
    int slotted, key_sh;   // slotted: 0 dense, 1 slots (keys beside them), 2 pairs of 64-bit words in slot layout (carried)
    uint32_t stage_pairs;   // pairs a scatter workgroup groups in LDS (0: STAGE_PAIRS)
    int carry;     // marker-less samples (dense only): the pairs ARE the 64-bit hashes — the partition sorts the hashes themselves by bucket
};
constexpr int PART_TPB = 256;
constexpr uint32_t BLK_PER_TILE = 16;     // slotted: blocks of the seeding kernel per partition tile (~3,000 occurrences)
constexpr uint32_t MAX_COARSE = 4096;     // coarse ranges (LDS counters of the histogram / scatter kernels)
constexpr uint32_t MAX_FINE = 4096;       // buckets per coarse range (LDS counters of the fine kernel)
#ifndef SYLPH_STAGE_PAIRS
#define SYLPH_STAGE_PAIRS 4096
#endif
constexpr uint32_t STAGE_PAIRS = SYLPH_STAGE_PAIRS;    // pairs a scatter workgroup groups in LDS before writing them out in runs

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give every XCD one contiguous eighth of the tiles, so
// that the runs two neighbouring tiles append to the same coarse range — adjacent in memory — meet in the same L2.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t n_tiles) {
    const uint32_t per_xcd = (n_tiles + 7) / 8;
    return (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);      // may be >= n_tiles for the padding of the last XCD's range
}

// f(key, index, hash) for every occurrence of tile t (any order; all threads of the workgroup take part; hash = 0 from slots).  Slotted: eight groups of
// 32 lanes walk eight blocks at a time (independent loads in flight instead of one block after the other).
template <class F>
__device__ __forceinline__ void for_tile_entries(const PartIn& in, uint32_t t, F&& f) {
    if (in.slotted == 1) {
        const uint32_t grp = threadIdx.x >> 5, l = threadIdx.x & 31;
        for (uint32_t bl = grp; bl < in.blk_per_tile; bl += PART_TPB / 32) {
            const uint32_t b = t * in.blk_per_tile + bl;
            if (b >= in.n_blk) break;
            const uint32_t cnt = min(in.blk_count[b], in.slot_cap), g0 = b * in.slot_cap;
            for (uint32_t i = l; i < cnt; i += 32) f(in.slot_key[g0 + i], g0 + i, 0ull);
        }
    } else if (in.slotted == 2) {
        // 64-bit words laid out like the slots, two per slot (a10.hip: the two filter operations of an occurrence); ~0 = none
        const uint32_t grp = threadIdx.x >> 5, l = threadIdx.x & 31;
        for (uint32_t bl = grp; bl < in.blk_per_tile; bl += PART_TPB / 32) {
            const uint32_t b = t * in.blk_per_tile + bl;
            if (b >= in.n_blk) break;
            const uint32_t cnt = 2u * min(in.blk_count[b], in.slot_cap), g0 = 2u * b * in.slot_cap;
            for (uint32_t i = l; i < cnt; i += 32) {
                const uint64_t h = in.hash[g0 + i];
                if (h != INVALID_HASH) f((uint32_t)(h >> in.key_sh), g0 + i, h);
            }
        }
    } else {
        const uint64_t i0 = (uint64_t)t * in.tile_entries;
        for (uint32_t e = threadIdx.x; e < in.tile_entries; e += PART_TPB) {
            const uint64_t i = i0 + e;
            if (i >= in.n_dense) break;
            const uint64_t h = in.hash[i];
            if (h != INVALID_HASH) f((uint32_t)(h >> in.key_sh), (uint32_t)i, h);
        }
    }
}

// hist[c * n_tiles + t] = occurrences of tile t in coarse range c = bucket >> fine_bits
// (also clears what the replay accumulates into — per-bucket counters, the list heads, the tail words — instead of memsets)
__global__ __launch_bounds__(PART_TPB) void part_hist_kernel(PartIn in, BucketMap bm, int fine_bits, uint32_t C, uint32_t n_tiles,
                                                             uint32_t* __restrict__ hist, uint32_t* __restrict__ zero, uint32_t n_zero,
                                                             uint32_t* __restrict__ tail16, uint32_t* __restrict__ list_a,
                                                             uint32_t* __restrict__ list_b, uint32_t* __restrict__ list_c) {
    __shared__ uint32_t s_h[MAX_COARSE];
    const uint32_t t = xcd_tile(n_tiles), gtid = blockIdx.x * PART_TPB + threadIdx.x;
    for (uint32_t i = gtid; i < n_zero; i += gridDim.x * PART_TPB) zero[i] = 0;
    if (tail16 && gtid < 16) tail16[gtid] = 0;
    if (list_a && gtid == 0) { *list_a = 0; *list_b = 0; *list_c = 0; }
    if (t >= n_tiles) return;
    for (uint32_t c = threadIdx.x; c < C; c += PART_TPB) s_h[c] = 0;
    __syncthreads();
    for_tile_entries(in, t, [&](uint32_t key, uint32_t, uint64_t) { atomicAdd(&s_h[bucket_of_key(key, bm) >> fine_bits], 1u); });
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < C; c += PART_TPB) hist[(size_t)c * n_tiles + t] = s_h[c];
}

// exclusive prefix sum of one value per lane across the workgroup; total returned through *total
template <int RTPB>
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t* s_wave, uint32_t* total);

// one workgroup per coarse range: hist row -> exclusive offsets of the tiles inside the range; total[c] = size of the range
// (round 6: rows of up to SCAN_ROW_LDS tiles go through LDS — read and written with consecutive lanes on consecutive words; with every
//  lane walking its own stretch of the row the 12.7 MB matrix of the filter pass's partition cost 44 MB of partial-line writes)
constexpr uint32_t SCAN_ROW_LDS = 4096;
__global__ __launch_bounds__(PART_TPB) void part_scan_kernel(uint32_t* __restrict__ hist, uint32_t n_tiles, uint32_t* __restrict__ total) {
    __shared__ uint32_t s_wave[PART_TPB / 64];
    __shared__ uint32_t s_row[SCAN_ROW_LDS + SCAN_ROW_LDS / 32];      // (a pad word per 32: the lanes' stretches start 13+ words apart)
    uint32_t* row = hist + (size_t)blockIdx.x * n_tiles;
    const uint32_t per = (n_tiles + PART_TPB - 1) / PART_TPB, a = threadIdx.x * per, b = min(n_tiles, a + per);
    const bool staged = n_tiles <= SCAN_ROW_LDS;
    auto at = [](uint32_t i) { return i + (i >> 5); };
    if (staged) {
        for (uint32_t i = threadIdx.x; i < n_tiles; i += PART_TPB) s_row[at(i)] = row[i];
        __syncthreads();
    }
    uint32_t sum = 0;
    for (uint32_t i = a; i < b; i++) sum += staged ? s_row[at(i)] : row[i];
    uint32_t tot = 0;
    uint32_t run = block_excl_sum<PART_TPB>(sum, s_wave, &tot);
    if (staged) {
        for (uint32_t i = a; i < b; i++) { const uint32_t v = s_row[at(i)]; s_row[at(i)] = run; run += v; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_tiles; i += PART_TPB) row[i] = s_row[at(i)];
    } else {
        for (uint32_t i = a; i < b; i++) { const uint32_t v = row[i]; row[i] = run; run += v; }
    }
    if (threadIdx.x == 0) total[blockIdx.x] = tot;
}

// (bucket, index) pairs of tile t -> their coarse ranges.  The workgroup first groups its pairs by range in LDS (count, scan,
// place with an LDS atomic), then writes them out position by position: neighbouring lanes write neighbouring pairs of one
// run (cbase[c] + offset of the tile inside the range + place inside the run) instead of 64 scattered 8-byte words per
// instruction.  The order inside a run is NOT the file order — the replay restores it from the indices.
__global__ __launch_bounds__(PART_TPB) void part_scatter_kernel(PartIn in, BucketMap bm, int fine_bits, uint32_t C, uint32_t n_tiles,
                                                                const uint32_t* __restrict__ offs, const uint32_t* __restrict__ total,
                                                                uint32_t* __restrict__ cbase, uint2* __restrict__ out) {
    extern __shared__ uint32_t s_dyn[];                  // [C] cursor (count -> start -> cursor) | [C] gb | STAGE_PAIRS pairs
    uint32_t* const s_cur = s_dyn;
    uint32_t* const s_gb = s_dyn + C;                    // global position of the range's run minus its start in the tile order
    uint2* const s_stage = reinterpret_cast<uint2*>(s_dyn + 2 * (size_t)C + ((2 * C) & 1u));
    __shared__ uint32_t s_wave[PART_TPB / 64];
    const uint32_t t = xcd_tile(n_tiles), stage = in.stage_pairs ? in.stage_pairs : STAGE_PAIRS;
    if (t >= n_tiles) return;
    uint32_t n_tile = 0;
    {   // tile order: start[c] = exclusive sum of the tile's counts; cbase = exclusive sum of the range sizes.  The tile's count per range is
        // what the histogram found — the next tile's offset inside the range minus this tile's (round 6: the tile was read and counted a
        // second time here before)
        const uint32_t per = (C + PART_TPB - 1) / PART_TPB, a = threadIdx.x * per, b = min(C, a + per);
        uint32_t sum_t = 0, sum_g = 0;
        for (uint32_t c = a; c < b; c++) {
            const uint32_t o0 = offs[(size_t)c * n_tiles + t], tot = total[c], o1 = t + 1 < n_tiles ? offs[(size_t)c * n_tiles + t + 1] : tot;
            s_cur[c] = o1 - o0;
            s_gb[c] = o0;
            sum_t += o1 - o0;
            sum_g += tot;
        }
        uint32_t tot_g = 0;
        uint32_t run_t = block_excl_sum<PART_TPB>(sum_t, s_wave, &n_tile);
        uint32_t run_g = block_excl_sum<PART_TPB>(sum_g, s_wave, &tot_g);
        for (uint32_t c = a; c < b; c++) {
            const uint32_t cnt = s_cur[c];
            if (t == 0) cbase[c] = run_g;
            s_gb[c] = run_g + s_gb[c] - run_t;
            s_cur[c] = run_t;
            run_t += cnt;
            run_g += total[c];
        }
        if (t == 0 && threadIdx.x == 0) cbase[C] = tot_g;
    }
    __syncthreads();
    for_tile_entries(in, t, [&](uint32_t key, uint32_t idx, uint64_t h) {
        const uint32_t b = bucket_of_key(key, bm), c = b >> fine_bits;
        const uint32_t p = atomicAdd(&s_cur[c], 1u);     // place in the tile order
        const uint2 pr = in.carry ? make_uint2((uint32_t)h, (uint32_t)(h >> 32)) : make_uint2(b, idx);
        if (p < stage) s_stage[p] = pr;
        else out[s_gb[c] + p] = pr;                      // (a tile fuller than the stage: the rest goes out directly)
    });
    __syncthreads();
    const uint32_t n_staged = min(n_tile, stage);
    for (uint32_t p = threadIdx.x; p < n_staged; p += PART_TPB) {
        const uint2 v = s_stage[p];
        const uint32_t b = in.carry ? bucket_of_key((uint32_t)((((uint64_t)v.y << 32) | v.x) >> in.key_sh), bm) : v.x;
        out[s_gb[b >> fine_bits] + p] = v;
    }
}

// one workgroup per coarse range: counting sort of its pairs by bucket in LDS -> perm (occurrence indices grouped by bucket) and
// boff[b] = first position of bucket b, boff[B] = number of valid occurrences
template <int TPB>
__global__ __launch_bounds__(TPB) void part_fine_kernel(const uint2* __restrict__ pairs,
                                                             const uint32_t* __restrict__ cbase, int fine_bits, uint32_t C, uint32_t B,
                                                             uint32_t* __restrict__ boff, uint32_t* __restrict__ perm, int carry, int key_sh,
                                                             BucketMap bm, uint64_t* __restrict__ sorted_hash) {
    __shared__ uint32_t s_cnt[MAX_FINE];
    __shared__ uint32_t s_wave[TPB / 64];
    const uint32_t c = blockIdx.x, F = 1u << fine_bits, b0 = c << fine_bits;
    const uint32_t lo = cbase[c], hi = cbase[c + 1];
    for (uint32_t f = threadIdx.x; f < F; f += TPB) s_cnt[f] = 0;
    __syncthreads();
    // (four independent loads in flight per lane: with one, a range of 10^5 pairs is a chain of load latencies)
    for (uint32_t e = lo + threadIdx.x; e < hi; e += 4 * TPB) {
        uint32_t k[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            k[u] = 0xFFFFFFFFu;
            if (e + u * TPB < hi) {
                if (carry) { const uint2 v = pairs[e + u * TPB]; k[u] = bucket_of_key((uint32_t)((((uint64_t)v.y << 32) | v.x) >> key_sh), bm); }
                else k[u] = pairs[e + u * TPB].x;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (k[u] != 0xFFFFFFFFu) atomicAdd(&s_cnt[k[u] - b0], 1u);
    }
    __syncthreads();
    {
        const uint32_t per = (F + TPB - 1) / TPB, a = threadIdx.x * per, b = min(F, a + per);
        uint32_t sum = 0;
        for (uint32_t f = a; f < b; f++) sum += s_cnt[f];
        uint32_t run = lo + block_excl_sum<TPB>(sum, s_wave, nullptr);
        for (uint32_t f = a; f < b; f++) {
            const uint32_t v = s_cnt[f];
            s_cnt[f] = run;                              // becomes the bucket's cursor
            if (b0 + f < B) boff[b0 + f] = run;
            run += v;
        }
    }
    if (c + 1 == C && threadIdx.x == 0) boff[B] = hi;
    __syncthreads();
    for (uint32_t e = lo + threadIdx.x; e < hi; e += 4 * TPB) {
        uint2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = e + u * TPB < hi ? pairs[e + u * TPB] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (e + u * TPB >= hi) continue;
            if (carry) {      // the sorted array holds the hashes themselves (marker-less samples: nothing else is ever looked at)
                const uint64_t h = ((uint64_t)v[u].y << 32) | v[u].x;
                sorted_hash[atomicAdd(&s_cnt[bucket_of_key((uint32_t)(h >> key_sh), bm) - b0], 1u)] = h;
            } else
                perm[atomicAdd(&s_cnt[v[u].x - b0], 1u)] = v[u].y;
        }
    }
}

// exclusive prefix sum of one value per lane across the workgroup (4 waves); total returned through *total
template <int RTPB>
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < RTPB / 64; w++) {
        const uint32_t t = s_wave[w];
        if ((uint32_t)w < wave) base += t;
        tot += t;
    }
    if (total) *total = tot;
    return base + x - v;
}

// F = 2^fine_bits buckets per coarse range (about 512 ranges)
struct PartGeom { int fine_bits; uint32_t C; };
inline PartGeom part_geometry(uint32_t B) {
    PartGeom g{6, 0};
    while ((1u << g.fine_bits) < MAX_FINE && ((B + (1u << g.fine_bits) - 1) >> g.fine_bits) > 512) g.fine_bits++;
    g.C = (B + (1u << g.fine_bits) - 1) >> g.fine_bits;
    SY_REQUIRE(g.C <= MAX_COARSE, "internal: %u coarse ranges", g.C);
    return g;
}
inline size_t part_hist_words(PartGeom g, uint32_t n_tiles) { return (size_t)g.C * n_tiles + 2 * (size_t)g.C + 2; }   // hist (C x n_tiles) | total (C) | cbase (C + 1)

// The four dispatches.  hist: part_hist_words() words; pairs: 8 B per entry; boff: B + 1 words (boff[B] = entries found);
// perm (entry indices grouped by bucket) or, with in.carry, sorted (the 64-bit words themselves grouped by bucket).
// zero / n_zero, tail16, lists: words the first kernel clears for the caller's later kernels (may be null).
inline void launch_partition(sylph_ctx* ctx, const PartIn& in, const BucketMap& bm, PartGeom g, uint32_t n_tiles, uint64_t n_expect,
                             uint32_t* hist, uint2* pairs, uint32_t* boff, uint32_t* perm, uint64_t* sorted, uint32_t* zero, uint32_t n_zero,
                             uint32_t* tail16, uint32_t* list_a, uint32_t* list_b, uint32_t* list_c) {
    uint32_t* ctotal = hist + (size_t)g.C * n_tiles;
    uint32_t* cbase = ctotal + g.C;
    const uint32_t tile_grid = ((n_tiles + 7) / 8) * 8;      // (padded: xcd_tile deals every XCD a contiguous eighth)
    hipLaunchKernelGGL(part_hist_kernel, dim3(tile_grid), dim3(PART_TPB), 0, ctx->stream, in, bm, g.fine_bits, g.C, n_tiles, hist, zero, n_zero,
                       tail16, list_a, list_b, list_c);
    hipLaunchKernelGGL(part_scan_kernel, dim3(g.C), dim3(PART_TPB), 0, ctx->stream, hist, n_tiles, ctotal);
    hipLaunchKernelGGL(part_scatter_kernel, dim3(tile_grid), dim3(PART_TPB), (2 * (size_t)g.C + 2 * (size_t)(in.stage_pairs ? in.stage_pairs : STAGE_PAIRS)) * 4, ctx->stream, in, bm,
                       g.fine_bits, g.C, n_tiles, hist, ctotal, cbase, pairs);
    // (a sample of tens of millions of occurrences — long reads at c = 100 — has ~10^5 pairs per coarse range: 1024
    //  threads walk them instead of 256; c5: 1.27 -> see profiles)
    if (n_expect / g.C > 32768)
        hipLaunchKernelGGL((part_fine_kernel<1024>), dim3(g.C), dim3(1024), 0, ctx->stream, pairs, cbase, g.fine_bits, g.C, bm.B, boff, perm,
                           in.carry, in.key_sh, bm, sorted);
    else
        hipLaunchKernelGGL((part_fine_kernel<PART_TPB>), dim3(g.C), dim3(PART_TPB), 0, ctx->stream, pairs, cbase, g.fine_bits, g.C, bm.B, boff, perm,
                           in.carry, in.key_sh, bm, sorted);
    SY_HIP(hipGetLastError());
}

// One level only (round 6, a10.hip): the first three dispatches with every bucket a coarse range of its own — the 64-bit words end up
// grouped by range in `pairs`, range c at [cbase[c], cbase[c + 1]) with cbase = part_cbase(hist, C, n_tiles); whoever reads the
// ranges resolves the rest in LDS.  bm.B = C <= MAX_COARSE.
inline const uint32_t* part_cbase(const uint32_t* hist, uint32_t C, uint32_t n_tiles) { return hist + (size_t)C * n_tiles + C; }
// have_hist: the caller's own kernel has written hist[c * n_tiles + t] already (a10.hip: while it wrote the words).
inline void launch_partition_coarse(sylph_ctx* ctx, const PartIn& in, const BucketMap& bm, uint32_t n_tiles, uint32_t* hist, uint2* pairs, bool have_hist = false) {
    const uint32_t C = bm.B;
    SY_REQUIRE(C >= 1 && C <= MAX_COARSE, "internal: %u ranges", C);
    uint32_t* ctotal = hist + (size_t)C * n_tiles;
    uint32_t* cbase = ctotal + C;
    const uint32_t tile_grid = ((n_tiles + 7) / 8) * 8;
    if (!have_hist)
        hipLaunchKernelGGL(part_hist_kernel, dim3(tile_grid), dim3(PART_TPB), 0, ctx->stream, in, bm, 0, C, n_tiles, hist, (uint32_t*)nullptr, 0u,
                           (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL(part_scan_kernel, dim3(C), dim3(PART_TPB), 0, ctx->stream, hist, n_tiles, ctotal);
    hipLaunchKernelGGL(part_scatter_kernel, dim3(tile_grid), dim3(PART_TPB), (2 * (size_t)C + 2 * (size_t)(in.stage_pairs ? in.stage_pairs : STAGE_PAIRS)) * 4, ctx->stream, in, bm,
                       0, C, n_tiles, hist, ctotal, cbase, pairs);
    SY_HIP(hipGetLastError());
}

}  // namespace
}  // namespace sylph
