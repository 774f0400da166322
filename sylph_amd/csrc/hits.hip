// hits.hip — from the probe's unsorted hit list to the result block, without a device-wide sort (round 4).
//
// What the caller of the probe half of get_stats wants (contain.rs:632-661) is, per genome, the number of hits and the sample
// counts of those hits in ascending order.  Rounds 1-3 radix-sorted the whole list of (row << cb | count) keys with the
// library's onesweep sort: 12 dispatches (4 kernels + 8 memsets) and a host round trip for the list's length before the first
// of them — 0.11 ms of a sample's 0.30 ms profile stage, 13 of its 19 dispatches.  But nothing needs the rows ordered against
// each other, only each row's values; and a row is never longer than its genome has k-mers (tens of thousands).  So:
//
//   hits_count_kernel    rc[row] += hits of the row.  The ~100 abundant genomes of a community own most hits, and device-wide
//                        atomics on one cache LINE are served one after the other on this chip (eight L2s: such atomics go to the
//                        fabric; ~30-90 per microsecond and line).  One atomic per hit took 0.39 ms (first version, 16 replicated
//                        counters); one per distinct row of a 2048-hit tile still 0.26 ms — the hot genomes have neighbouring
//                        ids, their counters shared 14 lines.  So: a few dozen workgroups, each aggregating a LONG stretch of the
//                        list in an LDS hash table (row -> count), one global atomic per distinct row and workgroup; and the
//                        counter of row r lives at (r x odd constant) mod 2^m, so that neighbouring rows never share a line
//   rows_sum_kernel      per tile of 2048 rows: the rows' totals (+ a snapshot of the list's length / largest value)
//   rows_scan_kernel     cov_off[row] (exclusive scan over the tile totals and inside the tile), contain_count[row]; rc[row] back
//                        to 0 (it becomes the row's cursor); the non-empty rows are listed (<= 64 hits: one wavefront each;
//                        longer: one workgroup)
//   hits_scatter_kernel  the same stretches again: a workgroup takes the places for all its hits of a row with ONE returning atomic
//                        on the row's cursor and hands them out from LDS; values land in their row's segment
//   rows_sort_kernel     every listed row sorted into the result block's narrow coverage values: <= 64 values ranked inside a
//                        wavefront, longer rows histogrammed over the value range in LDS and expanded by output position
//                        (consecutive lanes, consecutive bytes: one lane writing a bin's thousand copies byte by byte took 0.8 ms);
//                        re-zeroes the row's cursor and, at the very end, the probe's hit counter
//
// Every launch takes the list's length from DEVICE memory (the word the probe counted into): the host reads that word with an
// asynchronous copy while these kernels run and only needs it to size the final device -> host copy.  5 dispatches instead of 14,
// no memset, no round trip in front of them, no "last workgroup" tickets (512 workgroups taking a ticket on one word cost 0.06 ms).  Values of 4096 and above (a k-mer seen thousands of times) do not fit the LDS
// histogram: such a batch — decided from the same device word, the same way on both sides — goes through the sorted path of rounds
// 1-3 (finish_hits_sorted, contain.hip), which stays the fallback.
#include "contain_index.h"

namespace sylph {

namespace {

constexpr int ROWS_TPB = 256, ROWS_PER_THREAD = 8, ROWS_TILE = ROWS_TPB * ROWS_PER_THREAD;
constexpr uint32_t SMALL_ROW = 64;            // rows up to one wavefront of values are ranked by shuffles
constexpr uint32_t MAX_VALUE_LDS = 4096;      // rows_sort_kernel's histogram: values below this

struct HitsSrc {                 // where the list's length / largest value come from
    const uint32_t* d_cnt;       // device words [n_hits, max_count] (the probe's counter), or nullptr:
    uint32_t n_imm, max_imm;     // host-known values
    uint32_t cap;                // entries the hit array holds (a probe that overflowed counted more than it stored)
};
__device__ __forceinline__ uint32_t src_n(const HitsSrc& s) { return min(s.d_cnt ? s.d_cnt[0] : s.n_imm, s.cap); }
__device__ __forceinline__ uint32_t src_max(const HitsSrc& s) { return s.d_cnt ? s.d_cnt[1] : s.max_imm; }
__device__ __forceinline__ bool src_overflowed(const HitsSrc& s) { return s.d_cnt && s.d_cnt[0] > s.cap; }

// stretch hash: the hits of a workgroup's stretch of the list in an LDS table of HIT_SLOTS (row, count) entries (open addressing);
// a stretch with more distinct rows than the table may hold sends the surplus hits straight to the global counters
#ifndef SYLPH_HIT_SLOTS_LOG2
#define SYLPH_HIT_SLOTS_LOG2 12
#endif
constexpr uint32_t HIT_SLOTS = 1u << SYLPH_HIT_SLOTS_LOG2, HIT_SLOTS_FULL = HIT_SLOTS * 3 / 4, SLOT_EMPTY_ROW = 0xFFFFFFFFu, NO_SLOT = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t hash_row(uint32_t row) { return (row * 2654435761u) >> (32 - SYLPH_HIT_SLOTS_LOG2); }   // top bits
// -> the row's slot, claiming a free one.  Whether the table still takes NEW rows is decided between rounds of 256 hits, for the
// whole workgroup at once (`frozen`): a row then either owns a slot — and EVERY hit of it in the stretch goes through the slot —
// or never gets one, and every hit of it goes straight to the global counter; a hit-by-hit decision could turn one hit of a row
// away and take the next (a round adds at most 256 rows, and the table freezes a quarter below its capacity)
__device__ __forceinline__ uint32_t table_claim(uint32_t* s_key, uint32_t* s_used, uint32_t row, bool frozen) {
    uint32_t slot = hash_row(row);
    for (uint32_t probes = 0; probes < HIT_SLOTS; probes++) {
        uint32_t prev = s_key[slot];
        if (prev == row) return slot;
        if (prev == SLOT_EMPTY_ROW) {
            if (frozen) return NO_SLOT;
            prev = atomicCAS(&s_key[slot], SLOT_EMPTY_ROW, row);
            if (prev == SLOT_EMPTY_ROW) { atomicAdd(s_used, 1u); return slot; }
            if (prev == row) return slot;
        }
        slot = (slot + 1) & (HIT_SLOTS - 1);
    }
    return NO_SLOT;
}
// -> the row's slot if it has one (read-only: after the table has been built)
__device__ __forceinline__ uint32_t table_find(const uint32_t* s_key, uint32_t row) {
    uint32_t slot = hash_row(row);
    for (uint32_t probes = 0; probes < HIT_SLOTS; probes++) {
        const uint32_t k = s_key[slot];
        if (k == row) return slot;
        if (k == SLOT_EMPTY_ROW) return NO_SLOT;
        slot = (slot + 1) & (HIT_SLOTS - 1);
    }
    return NO_SLOT;
}
// where the counter of a row lives: a bijection of [0, 2^m) that sends neighbouring rows to different cache lines
__device__ __forceinline__ uint32_t rc_at(uint32_t row, uint32_t rc_mask) { return (row * 0x9E3779B1u) & rc_mask; }
// the stretch of workgroup b: whole multiples of 256 hits
__device__ __forceinline__ void stretch_of(uint32_t n, uint32_t& lo, uint32_t& hi) {
    const uint32_t per = ((n + gridDim.x - 1) / gridDim.x + 255u) & ~255u;
    lo = min(n, blockIdx.x * per);
    hi = min(n, lo + per);
}

__global__ __launch_bounds__(256) void hits_count_kernel(const uint64_t* __restrict__ hits, HitsSrc src, uint32_t n_rows, uint32_t rc_mask,
                                                         uint32_t* __restrict__ rc, uint32_t* __restrict__ lists) {
    __shared__ uint32_t s_key[HIT_SLOTS], s_cnt[HIT_SLOTS], s_used;
    if (blockIdx.x == 0 && threadIdx.x < 2) lists[threadIdx.x] = 0;      // (rows_scan_kernel appends to them: stream order)
    const uint32_t n = src_n(src);
    uint32_t lo, hi;
    stretch_of(n, lo, hi);
    if (lo >= hi) return;
    for (uint32_t s = threadIdx.x; s < HIT_SLOTS; s += 256) { s_key[s] = SLOT_EMPTY_ROW; s_cnt[s] = 0; }
    if (threadIdx.x == 0) s_used = 0;
    __syncthreads();
    bool frozen = false;
    for (uint32_t i0 = lo; i0 < hi; i0 += 256) {                       // rounds of 256 hits (lo, hi are multiples of 256 or n)
        const uint32_t i = i0 + threadIdx.x;
        if (i < hi) {
            const uint32_t row = (uint32_t)(hits[i] >> 32);
            if (row < n_rows) {
                const uint32_t slot = table_claim(s_key, &s_used, row, frozen);
                if (slot != NO_SLOT) atomicAdd(&s_cnt[slot], 1u);
                else atomicAdd(&rc[rc_at(row, rc_mask)], 1u);
            }
        }
        __syncthreads();
        frozen = frozen || s_used >= HIT_SLOTS_FULL;                   // (uniform: read between two barriers)
        __syncthreads();
    }
    for (uint32_t s = threadIdx.x; s < HIT_SLOTS; s += 256) {
        const uint32_t row = s_key[s];
        if (row != SLOT_EMPTY_ROW) atomicAdd(&rc[rc_at(row, rc_mask)], s_cnt[s]);
    }
}

__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* s_w) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// snap: [0] = entries of the hit list (clamped to the array), [1] = largest value, [2] = 1: not a batch for this path (a value
// beyond the LDS histogram, or the probe overflowed its array) — what the later kernels read instead of the probe's live counter
__global__ __launch_bounds__(ROWS_TPB) void rows_sum_kernel(const uint32_t* __restrict__ rc, uint32_t n_rows, uint32_t rc_mask,
                                                            uint32_t* __restrict__ tile_sum, HitsSrc src, uint32_t* __restrict__ snap) {
    __shared__ uint32_t s_w[4];
    const uint32_t t = blockIdx.x;
    if (t == 0 && threadIdx.x == 0) {
        snap[0] = src_n(src);
        snap[1] = src_max(src);
        snap[2] = (src_max(src) >= MAX_VALUE_LDS || src_overflowed(src)) ? 1u : 0u;
    }
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; j++) {
        const uint32_t row = t * ROWS_TILE + j * ROWS_TPB + threadIdx.x;
        if (row < n_rows) v += rc[rc_at(row, rc_mask)];
    }
    const uint32_t tot = block_sum_256(v, s_w);
    if (threadIdx.x == 0) tile_sum[t] = tot;
}

// lists: [0] = number of small rows, [1] = number of long rows, [2..] small rows from the front, long rows from the back
__global__ __launch_bounds__(ROWS_TPB) void rows_scan_kernel(uint32_t* __restrict__ rc, uint32_t n_rows, uint32_t rc_mask, uint32_t n_tiles,
                                                             const uint32_t* __restrict__ tile_sum, uint64_t* __restrict__ cov_off,
                                                             uint32_t* __restrict__ contain_count, uint32_t* __restrict__ lists,
                                                             uint32_t list_cap) {
    __shared__ uint32_t s_w[4], s_small[4], s_big[4], s_base[2], s_w2[4];
    const uint32_t t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // hits in the tiles before this one (and, for the last tile, in all of them): every workgroup adds the totals up itself — a
    // batch of 64 samples has 3,500 of them, read from L2
    uint32_t pre = 0, all = 0;
    for (uint32_t i = threadIdx.x; i < n_tiles; i += ROWS_TPB) {
        const uint32_t x = tile_sum[i];
        all += x;
        if (i < t) pre += x;
    }
    const uint32_t tile_base = block_sum_256(pre, s_w2);
    const uint32_t grand = (t == n_tiles - 1) ? block_sum_256(all, s_w2) : 0u;       // (uniform branch: t is per workgroup)
    // a thread owns ROWS_PER_THREAD CONSECUTIVE rows (its scan is a running sum)
    const uint32_t row0 = t * ROWS_TILE + threadIdx.x * ROWS_PER_THREAD;
    uint32_t cnt[ROWS_PER_THREAD];
    uint32_t mine = 0, n_small = 0, n_big = 0;
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; j++) {
        const uint32_t row = row0 + j;
        uint32_t s = 0;
        if (row < n_rows) { const uint32_t a = rc_at(row, rc_mask); s = rc[a]; if (s) rc[a] = 0; }   // from here on the row's cursor
        cnt[j] = s;
        mine += s;
        n_small += s != 0 && s <= SMALL_ROW;
        n_big += s > SMALL_ROW;
    }
    // workgroup-exclusive scan of `mine`, and the places of this thread's rows in the two lists
    uint32_t incl = mine, is = n_small, ib = n_big;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, d), ys = (uint32_t)__shfl_up((int)is, d), yb = (uint32_t)__shfl_up((int)ib, d);
        if (lane >= (uint32_t)d) { incl += y; is += ys; ib += yb; }
    }
    if (lane == 63) { s_w[wave] = incl; s_small[wave] = is; s_big[wave] = ib; }
    __syncthreads();
    uint32_t before = tile_base, bs = 0, bb = 0, tot_s = 0, tot_b = 0;
    for (uint32_t w = 0; w < 4; w++) {
        if (w < wave) { before += s_w[w]; bs += s_small[w]; bb += s_big[w]; }
        tot_s += s_small[w];
        tot_b += s_big[w];
    }
    if (threadIdx.x == 0) {
        s_base[0] = tot_s ? atomicAdd(&lists[0], tot_s) : 0u;
        s_base[1] = tot_b ? atomicAdd(&lists[1], tot_b) : 0u;
    }
    __syncthreads();
    uint32_t o = before + incl - mine, ps = s_base[0] + bs + is - n_small, pb = s_base[1] + bb + ib - n_big;
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; j++) {
        const uint32_t row = row0 + j;
        if (row < n_rows) {
            cov_off[row] = o;
            contain_count[row] = cnt[j];
            if (cnt[j] != 0 && cnt[j] <= SMALL_ROW) lists[2 + ps++] = row;
            else if (cnt[j] > SMALL_ROW) lists[2 + list_cap - 1 - pb++] = row;
            o += cnt[j];
        }
    }
    if (t == n_tiles - 1 && threadIdx.x == 0) cov_off[n_rows] = grand;
}

// value of every hit -> its row's segment of `vals`.  The workgroup walks its stretch twice: first the table (row -> hits in the
// stretch), then ONE returning atomic per distinct row takes the places of all of them from the row's cursor, then every hit
// takes its place from LDS (segment start + the stretch's first place + an LDS counter)
__global__ __launch_bounds__(256) void hits_scatter_kernel(const uint64_t* __restrict__ hits, const uint32_t* __restrict__ snap, uint32_t n_rows,
                                                           uint32_t rc_mask, uint32_t* __restrict__ rc, const uint64_t* __restrict__ cov_off,
                                                           uint32_t* __restrict__ vals) {
    __shared__ uint32_t s_key[HIT_SLOTS], s_cnt[HIT_SLOTS], s_used;
    if (snap[2]) return;                                              // the sorted path takes this batch (uniform over the grid)
    const uint32_t n = snap[0];
    uint32_t lo, hi;
    stretch_of(n, lo, hi);
    if (lo >= hi) return;
    for (uint32_t s = threadIdx.x; s < HIT_SLOTS; s += 256) { s_key[s] = SLOT_EMPTY_ROW; s_cnt[s] = 0; }
    if (threadIdx.x == 0) s_used = 0;
    __syncthreads();
    bool frozen = false;
    for (uint32_t i0 = lo; i0 < hi; i0 += 256) {                       // (rounds and freezing as in hits_count_kernel)
        const uint32_t i = i0 + threadIdx.x;
        if (i < hi) {
            const uint32_t row = (uint32_t)(hits[i] >> 32);
            if (row < n_rows) {
                const uint32_t slot = table_claim(s_key, &s_used, row, frozen);
                if (slot != NO_SLOT) atomicAdd(&s_cnt[slot], 1u);
            }
        }
        __syncthreads();
        frozen = frozen || s_used >= HIT_SLOTS_FULL;
        __syncthreads();
    }
    for (uint32_t s = threadIdx.x; s < HIT_SLOTS; s += 256) {        // s_cnt[s]: hits of the row in the stretch -> the next free place
        const uint32_t row = s_key[s];
        if (row != SLOT_EMPTY_ROW) s_cnt[s] = (uint32_t)cov_off[row] + atomicAdd(&rc[rc_at(row, rc_mask)], s_cnt[s]);
    }
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        const uint64_t h = hits[i];
        const uint32_t row = (uint32_t)(h >> 32);
        if (row >= n_rows) continue;
        const uint32_t slot = table_find(s_key, row);
        const uint32_t at = slot != NO_SLOT ? atomicAdd(&s_cnt[slot], 1u)
                                            : (uint32_t)cov_off[row] + atomicAdd(&rc[rc_at(row, rc_mask)], 1u);   // (a row the full table turned away)
        vals[at] = (uint32_t)h;
    }
}

__device__ __forceinline__ void store_cov(void* covs, uint32_t width, uint32_t at, uint32_t v) {
    if (width == 1) ((uint8_t*)covs)[at] = (uint8_t)v;
    else if (width == 2) ((uint16_t*)covs)[at] = (uint16_t)v;
    else ((uint32_t*)covs)[at] = v;
}

// want_narrow: 1 = values stored with the narrowest of 1 / 2 / 4 bytes that holds the batch's largest one (the host applies the
// same rule to the same word: finish_hits), 0 = always u32
__global__ __launch_bounds__(256) void rows_sort_kernel(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ snap, uint32_t rc_mask,
                                                        uint32_t* __restrict__ rc, const uint64_t* __restrict__ cov_off,
                                                        const uint32_t* __restrict__ contain_count, const uint32_t* __restrict__ lists,
                                                        uint32_t list_cap, int want_narrow, void* __restrict__ covs,
                                                        uint32_t* __restrict__ probe_counter) {
    __shared__ uint32_t s_bins[MAX_VALUE_LDS + 1];
    __shared__ uint32_t s_w[4];
    const uint32_t mx = snap[1];
    if (snap[2]) return;                                              // (the host clears what this path would have cleaned)
    // the probe's counter goes back to zero for the next batch: its value is in the snapshot, and on its way to the host already
    // (the copy was queued in front of these kernels)
    if (probe_counter && blockIdx.x == 0 && threadIdx.x < 2) probe_counter[threadIdx.x] = 0;
    const uint32_t n_small = lists[0], n_big = lists[1];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t width = 4;
    if (want_narrow) width = mx < 256u ? 1u : mx < 65536u ? 2u : 4u;
    // ---- rows of at most one wavefront of values: one wavefront each, every value ranked against the row's others
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t e = blockIdx.x * 4 + wave; e < n_small; e += n_waves) {
        const uint32_t row = lists[2 + e];
        const uint32_t n = contain_count[row], off = (uint32_t)cov_off[row];
        const uint32_t v = lane < n ? vals[off + lane] : 0xFFFFFFFFu;
        uint32_t rank = 0;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t u = (uint32_t)__shfl((int)v, (int)i);
            rank += (u < v) || (u == v && i < lane);
        }
        if (lane < n) store_cov(covs, width, off + rank, v);
        if (lane == 0) rc[rc_at(row, rc_mask)] = 0;
    }
    // ---- longer rows: one workgroup each; histogram over the value range in LDS, exclusive scan, expansion by output place
    const uint32_t n_bins = mx + 1;                                   // <= MAX_VALUE_LDS
    for (uint32_t e = blockIdx.x; e < n_big; e += gridDim.x) {
        const uint32_t row = lists[2 + list_cap - 1 - e];
        const uint32_t n = contain_count[row], off = (uint32_t)cov_off[row];
        for (uint32_t b = threadIdx.x; b <= n_bins; b += 256) s_bins[b] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 256) atomicAdd(&s_bins[min(vals[off + i], mx)], 1u);
        __syncthreads();
        // in-place exclusive scan of the bins: each thread takes a run of consecutive bins
        const uint32_t per = (n_bins + 255) / 256;
        const uint32_t b0 = min(threadIdx.x * per, n_bins), b1 = min(b0 + per, n_bins);
        uint32_t mine = 0;
        for (uint32_t b = b0; b < b1; b++) mine += s_bins[b];
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= (uint32_t)d) incl += y;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t o = incl - mine;
        for (uint32_t w = 0; w < wave; w++) o += s_w[w];
        for (uint32_t b = b0; b < b1; b++) { const uint32_t c = s_bins[b]; s_bins[b] = o; o += c; }   // (a thread only touches its own run)
        if (threadIdx.x == 255) s_bins[n_bins] = n;                   // sentinel: every place lies below the "start" of bin n_bins
        __syncthreads();
        // value v fills places [start[v], start[v + 1]): a wavefront per bin, consecutive lanes write consecutive places (a binary
        // search per place was a chain of dependent LDS reads: 0.02 ms for a 17,000-hit row)
        for (uint32_t b = wave; b < n_bins; b += 4) {
            const uint32_t p0 = s_bins[b], p1 = s_bins[b + 1];
            for (uint32_t p = p0 + lane; p < p1; p += 64) store_cov(covs, width, off + p, b);
        }
        if (threadIdx.x == 0) rc[rc_at(row, rc_mask)] = 0;
        __syncthreads();
    }
}

}  // namespace

// Launches the five kernels for the hit list in db->hits.  d_cnt != nullptr: the list's length and largest value are the two
// device words the probe counted into (the caller learns them from its asynchronous copy); else n_imm / max_imm.  Returns false —
// having launched nothing — when the row path cannot take the batch for a reason the HOST can see (host-known largest value
// beyond the LDS histogram).  Output: d_res filled per ResultLayout(n_rows, ., width, .) — cov_off, contain_count, and the
// coverage values at their fixed offset — for any n_hits.
bool launch_row_assembly(sylph_db* db, const uint32_t* d_cnt, uint32_t n_imm, uint32_t max_imm, uint32_t cap, uint64_t n_rows, int want_narrow,
                         char* d_res, size_t covs_offset, size_t ccount_offset) {
    sylph_ctx* ctx = db->ctx;
    if (!d_cnt && max_imm >= MAX_VALUE_LDS) return false;
    if (n_rows == 0 || n_rows >= (1ull << 31)) return false;
    const uint32_t R = (uint32_t)n_rows;
    uint32_t R2 = 1024;                                        // counters: a power of two (rc_at is a bijection of [0, R2))
    while (R2 < R) R2 <<= 1;
    const uint32_t n_tiles = (R + ROWS_TILE - 1) / ROWS_TILE;
    const uint32_t list_cap = (uint32_t)std::min<uint64_t>(R, std::max<uint32_t>(cap, 1));
    const size_t rc_bytes = (size_t)R2 * 4, meta_bytes = ((size_t)n_tiles + 4 + 2 + list_cap) * 4 + 64;
    // rc must be all zero on entry.  The kernels leave it that way; after a failure, a fallback batch or a reallocation it is cleared here.
    if (db->row_counters.cap < rc_bytes || db->rc_rows != R2) { db->row_counters.reserve(rc_bytes); db->rc_dirty = true; }
    if (db->row_meta.cap < meta_bytes) db->row_meta.reserve(meta_bytes);
    uint32_t* rc = db->row_counters.as<uint32_t>();
    uint32_t* tile_sum = db->row_meta.as<uint32_t>();          // n_tiles
    uint32_t* snap = tile_sum + n_tiles;                       // 4 words
    uint32_t* lists = snap + 4;                                // [0], [1] lengths, then list_cap entries
    if (db->rc_dirty) SY_HIP(hipMemsetAsync(rc, 0, rc_bytes, ctx->stream));
    db->rc_dirty = true;                                       // until the chain below has been queued completely
    db->rc_rows = R2;
    db->hits_sorted.reserve((size_t)std::max<uint32_t>(cap, 1) * 4);   // the values, grouped by row, before they are sorted
    uint32_t* vals = db->hits_sorted.as<uint32_t>();
    const HitsSrc src{d_cnt, n_imm, max_imm, cap};
    // a few dozen long stretches: the fewer workgroups, the fewer atomics per row counter (one per workgroup and distinct row)
    const uint64_t n_guess = d_cnt ? (uint64_t)cap / 2 : n_imm;
    const uint32_t hit_grid = (uint32_t)std::max<uint64_t>(16, std::min<uint64_t>(n_guess / 6144, 768));
    uint64_t* cov_off = reinterpret_cast<uint64_t*>(d_res);
    uint32_t* ccount = reinterpret_cast<uint32_t*>(d_res + ccount_offset);
    const uint64_t* hits = db->hits.as<uint64_t>();
    hipLaunchKernelGGL(hits_count_kernel, dim3(hit_grid), dim3(256), 0, ctx->stream, hits, src, R, R2 - 1, rc, lists);
    hipLaunchKernelGGL(rows_sum_kernel, dim3(n_tiles), dim3(ROWS_TPB), 0, ctx->stream, rc, R, R2 - 1, tile_sum, src, snap);
    hipLaunchKernelGGL(rows_scan_kernel, dim3(n_tiles), dim3(ROWS_TPB), 0, ctx->stream, rc, R, R2 - 1, n_tiles, tile_sum, cov_off, ccount, lists, list_cap);
    hipLaunchKernelGGL(hits_scatter_kernel, dim3(hit_grid), dim3(256), 0, ctx->stream, hits, snap, R, R2 - 1, rc, cov_off, vals);
    hipLaunchKernelGGL(rows_sort_kernel, dim3(512), dim3(256), 0, ctx->stream, vals, snap, R2 - 1, rc, cov_off, ccount, lists, list_cap, want_narrow,
                       (void*)(d_res + covs_offset), const_cast<uint32_t*>(d_cnt));
    SY_HIP(hipGetLastError());
    db->rc_dirty = false;
    return true;
}

uint32_t row_assembly_max_value() { return MAX_VALUE_LDS; }

}  // namespace sylph
