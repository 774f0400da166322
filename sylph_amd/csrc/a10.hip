// a10.hip — the reference's DEFAULT paired-end dedup: dup_removal_lsh_full (sketch.rs:733-769) over a scalable cuckoo filter
// (built at sketch.rs:796-804: initial capacity 10^7, false-positive probability --fpr; cmdline.rs:77).
//
// What the filter changes against the exact set of dup_removal_lsh_full_exact is ONE thing: `contains((k-mer, markers))` may
// answer true for a pair that was never inserted.  A cuckoo filter stores, per inserted item, a fingerprint f in one of the two
// buckets {i1, i1 ^ d(f)}; evictions move a fingerprint between ITS two buckets only.  So, as long as no insertion fails (the
// filters run at <= 60 % load: four entries per bucket, the bucket count the next power of two above capacity / 4),
//
//     contains(x)  <=>  some item y inserted earlier has the same reduced key  g(x) = (f(x), min(i1(x), i2(x)))
//
// whatever the eviction history was.  The sequential walk of sketch.rs:733-769 — test, insert when absent — therefore reduces
// to "x is contained iff it is not the FIRST operation of its reduced-key class", over ALL operations of the sample in file
// order (two per processed occurrence: (k-mer, markers.0) then (k-mer, markers.1); classes span k-mers — that is what a false
// positive is).  That is order-free except for "first", and "first" is a minimum: one pass enters every operation into a
// device-wide table keyed by g with an atomic minimum of the operation index, a second pass compares.  The occurrence whose
// test came back "contained" gets RID_A10_BIT in its record; the replay (replay_lds.hip, sketch.hip) reads that bit where the
// exact path compares markers, and applies the same `*c > 0` rule (:749, :756).
//
// Growth (the "scalable" part): filter j holds cap0 * 2^j items at fpr * 0.9^j; the insert that finds it full opens filter
// j + 1, `contains` asks every filter.  The operation that opens filter j + 1 is the one with cap_j inserting operations of
// phase j before it: phases are resolved one after the other (table of the phase, count of the inserting operations per tile,
// the cut found on the host), each a pair of passes over the operations behind the cut.  A 1 Gbp sample (8 M operations)
// never leaves filter 0 and takes the two passes with no host round trip.
//
// The crate (scalable_cuckoo_filter 0.2.4) is not in /root/reference: fingerprint width, bucket count, growth rule follow its
// documentation, the hash bits are this repository's (the ScalableCuckoo model of the tests' CPU checker; the
// two agree bit for bit, tests/test_gpu_parity.py).  Which pairs collide therefore differs from a run of the reference;
// how many do, and what a collision does, does not.  DESIGN.md §1.
#include <cmath>
#include <memory>

#include "common.h"
#include "device_common.h"
#include "sketch_session.h"
#include "partition.h"

namespace sylph {
namespace {

constexpr uint64_t FX_K = 0x517cc1b727220a95ull;        // rustc-hash 1.x
constexpr uint64_t GOLD = 0x9E3779B97F4A7C15ull;
constexpr int MAX_FILTERS = 8;                           // capacities cap0 * (2^8 - 1): 2.5e9 items for the default cap0
constexpr uint64_t NO_OP = ~0ull;
constexpr uint32_t TILE_OCC = 1024;                      // occurrences per tile of the inserting-operation count
constexpr int A10_TPB = 256;

__host__ __device__ __forceinline__ uint64_t fx_add(uint64_t h, uint64_t w) { return (((h << 5) | (h >> 59)) ^ w) * FX_K; }
// FxHasher over the tuple (u64, [u32; 2]), mixed once more (the model's choice of 64 hash bits)
__device__ __forceinline__ uint64_t item_hash(uint64_t km, uint64_t marker) {
    return fx_add(fx_add(fx_add(0, km), marker & 0xffffffffull), marker >> 32) * GOLD;
}
// The place of an operation in the walk of sketch.rs:806-867: records in file order (mate 1 of a pair, then mate 2), inside a record
// the order extract_markers emitted its seeds in (the rank the seeding kernels left in the rid, sketch_session.h), markers.0 before
// markers.1.  One 64-bit word, smaller = earlier.
__device__ __forceinline__ uint64_t op_key(uint64_t rid, uint32_t w) {
    return ((rid & RID_MASK) << 21) | (((rid >> RID_RANK_SHIFT) & RID_RANK_MAX) << 1) | w;
}

struct alignas(16) Ent { unsigned long long key; unsigned long long inv_op; };   // inv_op = NO_OP - earliest operation of the class (0: none yet)
static_assert(sizeof(Ent) == 16, "table entry");

struct Filter {
    Ent* tab;
    uint32_t tab_shift;       // slot of a key = (key * GOLD) >> tab_shift
    uint32_t tab_mask;
    uint32_t fp_mask;         // fingerprint bits of the filter
    uint32_t nb_mask;         // buckets - 1
    uint64_t cut;             // operations >= cut did not insert into this filter (NO_OP while it is the current one)
};
struct Filters { Filter f[MAX_FILTERS]; int n; uint64_t begin, end; };   // operations in [begin, end) are this phase's

__device__ __forceinline__ uint64_t reduced_key(const Filter& d, uint64_t h) {
    uint32_t f = (uint32_t)(h >> 32) & d.fp_mask;
    if (!f) f = 1u;
    const uint32_t i1 = (uint32_t)h & d.nb_mask;
    const uint32_t i2 = (i1 ^ (uint32_t)(fx_add(0, f) >> 11)) & d.nb_mask;
    return ((uint64_t)f << 32) | min(i1, i2);
}
// Lookups run in kernels of their own, behind the one that filled the table (or on closed tables): one plain 16-byte load per slot.
__device__ __forceinline__ uint64_t table_first_op(const Filter& d, uint64_t g) {
    uint32_t s = (uint32_t)((g * GOLD) >> d.tab_shift);
    for (;;) {
        const ulonglong2 e = *reinterpret_cast<const ulonglong2*>(&d.tab[s]);
        if (e.x == g) return NO_OP - e.y;
        if (e.x == 0) return NO_OP;
        s = (s + 1) & d.tab_mask;
    }
}
// The slot's key is read first (an agent-scope load: the table is written with device-scope atomics, which act behind the XCDs'
// L2s) and claimed only when it is empty; a later operation of a class whose earlier one is already in place leaves without the
// second atomic (threads run roughly in file order).  (Claiming first — one compare-and-swap that also tells whose slot it is —
// measured 6 % slower for the whole sample: a failed compare-and-swap costs more than the load it replaces.)
__device__ __forceinline__ void table_enter(const Filter& d, uint64_t g, uint64_t op) {
    uint32_t s = (uint32_t)((g * GOLD) >> d.tab_shift);
    const unsigned long long mine = NO_OP - op;
    for (;;) {
        unsigned long long k = __hip_atomic_load(&d.tab[s].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == 0) k = atomicCAS(&d.tab[s].key, 0ull, (unsigned long long)g);
        if (k == 0 || k == g) {
            if (k == 0 || __hip_atomic_load(&d.tab[s].inv_op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(&d.tab[s].inv_op, mine);
            return;
        }
        s = (s + 1) & d.tab_mask;
    }
}

struct Occs { const uint64_t* hash; OccRec* recs; uint32_t n; const uint8_t* part; };   // part[i]: occurrence i takes part (a10_part_kernel)

// Occurrence i takes part in the filter's bookkeeping iff it is valid, its pair has markers (sketch.rs:745) and it is not a
// mate-2 occurrence of a k-mer that mate 1 of the same pair produced too (:852).  The dense arrays are in file order record by
// record: the pair's mate-1 occurrences lie right before its mate-2 occurrences.
// (The walk is linear in the pair's seeds in front of i — quadratic per pair, which only shows for pairs of long reads at c = 1: it
//  is done ONCE per sample, by a10_part_kernel; every other kernel reads its verdict.)
__device__ __forceinline__ bool walk_takes_part(const Occs& o, uint32_t i, uint64_t& km, uint64_t& rid, uint64_t& m0, uint64_t& m1) {
    // (every output is assigned before the walk below, whatever the verdict: with the assignments behind the loop hipcc 7.2 zeroed
    //  m0 for the lanes that had walked — seen in the ISA of the lookup kernels, and in half of the mate-2 lookups missing)
    const OccRec r = o.recs[i];
    km = o.hash[i];
    rid = r.rid;
    m0 = r.m0;
    m1 = r.m1;
    if (km == INVALID_HASH || !(r.rid & RID_MARKER_BIT)) return false;
    const uint64_t rec = r.rid & RID_MASK;
    bool mate1_has_it = false;
    if (rec & 1) {
        for (uint32_t j = i; j > 0 && !mate1_has_it;) {
            j--;
            const uint64_t hj = o.hash[j];
            if (hj == INVALID_HASH) continue;
            const uint64_t rj = o.recs[j].rid & RID_MASK;
            if ((rj >> 1) != (rec >> 1)) break;
            mate1_has_it = !(rj & 1) && hj == km;
        }
    }
    return !mate1_has_it;
}
__global__ __launch_bounds__(A10_TPB) void a10_part_kernel(Occs o, uint8_t* __restrict__ part) {
    const uint32_t i = blockIdx.x * A10_TPB + threadIdx.x;
    if (i >= o.n) return;
    uint64_t km, rid, m0, m1;
    part[i] = walk_takes_part(o, i, km, rid, m0, m1) ? 1 : 0;
    if (rid & RID_A10_BIT) o.recs[i].rid = rid & ~RID_A10_BIT;     // a mark the partitioned pass left before it was found unfit (a10_verdict)
}
__device__ __forceinline__ bool takes_part(const Occs& o, uint32_t i, uint64_t& km, uint64_t& rid, uint64_t& m0, uint64_t& m1) {
    const OccRec r = o.recs[i];
    km = r.hash;
    rid = r.rid;
    m0 = r.m0;
    m1 = r.m1;
    return o.part[i] != 0;
}
// true: an earlier, closed filter holds the operation's reduced key
__device__ __forceinline__ bool in_closed_filters(const Filters& F, uint64_t h) {
    for (int q = 0; q + 1 < F.n; q++)
        if (table_first_op(F.f[q], reduced_key(F.f[q], h)) < F.f[q].cut) return true;
    return false;
}

__global__ __launch_bounds__(A10_TPB) void a10_enter_kernel(Occs o, Filters F) {
    const uint32_t i = blockIdx.x * A10_TPB + threadIdx.x;
    if (i >= o.n) return;
    uint64_t km, rid, m[2];
    if (!takes_part(o, i, km, rid, m[0], m[1])) return;
    const Filter& cur = F.f[F.n - 1];
    for (uint32_t w = 0; w < 2; w++) {
        const uint64_t op = op_key(rid, w);
        if (op < F.begin) continue;
        const uint64_t h = item_hash(km, m[w]);
        if (!in_closed_filters(F, h)) table_enter(cur, reduced_key(cur, h), op);
    }
}
// 0: contained (an earlier operation of its class, or a closed filter);  1: this operation inserts into the current filter
__device__ __forceinline__ int op_inserts(const Filters& F, uint64_t h, uint64_t op) {
    if (in_closed_filters(F, h)) return 0;
    const Filter& cur = F.f[F.n - 1];
    return table_first_op(cur, reduced_key(cur, h)) == op ? 1 : 0;
}
__global__ __launch_bounds__(A10_TPB) void a10_flag_kernel(Occs o, Filters F) {
    const uint32_t i = blockIdx.x * A10_TPB + threadIdx.x;
    if (i >= o.n) return;
    uint64_t km, rid, m[2];
    if (!takes_part(o, i, km, rid, m[0], m[1])) return;
    bool contained = false;
    for (uint32_t w = 0; w < 2; w++) {
        const uint64_t op = op_key(rid, w);
        if (op < F.begin || op >= F.end) continue;
        if (!op_inserts(F, item_hash(km, m[w]), op)) contained = true;
    }
    if (contained) o.recs[i].rid = rid | RID_A10_BIT;      // (the thread's own record; other threads read it through RID_MASK)
}
// inserting operations of the phase per tile of TILE_OCC occurrences
__global__ __launch_bounds__(A10_TPB) void a10_count_kernel(Occs o, Filters F, uint32_t* __restrict__ tile_count) {
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t e = threadIdx.x; e < TILE_OCC; e += A10_TPB) {
        const uint32_t i = blockIdx.x * TILE_OCC + e;
        if (i >= o.n) continue;
        uint64_t km, rid, m[2];
        if (!takes_part(o, i, km, rid, m[0], m[1])) continue;
        for (uint32_t w = 0; w < 2; w++) {
            const uint64_t op = op_key(rid, w);
            if (op >= F.begin) mine += (uint32_t)op_inserts(F, item_hash(km, m[w]), op);
        }
    }
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = s_n;
}
// The operation that opens the next filter: the one with `rank` inserting operations of the phase before it, counted from the
// start of tile `tile`.  Tiles follow the dense arrays, which are in file order record by record but not inside a record (see
// op_key): the workgroup finds the occurrence that holds inserting operation `rank` in ARRAY order; the operation wanted is in
// the same record — the records before it hold the same inserting operations in either order — and one lane picks it among the
// record's inserting operations by their keys.
__global__ __launch_bounds__(A10_TPB) void a10_find_kernel(Occs o, Filters F, uint32_t tile, uint32_t rank, unsigned long long* __restrict__ run_keys,
                                                            uint32_t run_cap, uint64_t* __restrict__ out_op) {
    constexpr uint32_t PER = TILE_OCC / A10_TPB;
    __shared__ uint32_t s_cnt[A10_TPB];
    uint32_t mine = 0;
    uint8_t ins[PER];          // inserting operations (0..2) of the lane's PER consecutive occurrences
    for (uint32_t e = 0; e < PER; e++) {
        const uint32_t i = tile * TILE_OCC + threadIdx.x * PER + e;
        ins[e] = 0;
        if (i >= o.n) continue;
        uint64_t km, rid, m[2];
        if (!takes_part(o, i, km, rid, m[0], m[1])) continue;
        for (uint32_t w = 0; w < 2; w++) {
            const uint64_t op = op_key(rid, w);
            if (op >= F.begin) ins[e] += (uint8_t)op_inserts(F, item_hash(km, m[w]), op);
        }
        mine += ins[e];
    }
    s_cnt[threadIdx.x] = mine;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t t = 0; t < threadIdx.x; t++) before += s_cnt[t];
    // one lane holds inserting operation `rank` in array order: it names the occurrence and its record's run of occurrences
    // [lo, hi) in the dense arrays (invalid entries in between belong to no record: skipped)
    __shared__ uint32_t s_run[3];                         // at, lo, hi
    __shared__ uint32_t s_t;                              // the record's inserting operations before number `rank` in array order
    if (rank >= before && rank < before + mine) {
        uint32_t at = 0, r = before;
        for (uint32_t e = 0; e < PER; e++) {
            if (rank < r + ins[e]) { at = tile * TILE_OCC + threadIdx.x * PER + e; break; }
            r += ins[e];
        }
        const uint64_t rec = o.recs[at].rid & RID_MASK;
        uint32_t lo = at, hi = at + 1;
        while (lo > 0 && (o.hash[lo - 1] == INVALID_HASH || (o.recs[lo - 1].rid & RID_MASK) == rec)) lo--;
        while (hi < o.n && (o.hash[hi] == INVALID_HASH || (o.recs[hi].rid & RID_MASK) == rec)) hi++;
        s_run[0] = at; s_run[1] = lo; s_run[2] = hi;
        s_t = rank - r;                                   // (those of `at` itself that come before number `rank`)
    }
    __syncthreads();
    const uint32_t at = s_run[0], lo = s_run[1], hi = s_run[2];
    // The operation wanted is the record's inserting operation with exactly t of the record's inserting operations before it BY
    // KEY, t = those before number `rank` in array order (the records in front hold the same inserting operations in either
    // order).  The record's keys go to a scratch array (a record of a long read pair at c = 1 has thousands).
    if (2 * (hi - lo) > run_cap) return;                  // (out_op stays unset: the host reports it)
    for (uint32_t i = lo + threadIdx.x; i < hi; i += A10_TPB) {
        uint64_t km, rid, m[2];
        const bool part = takes_part(o, i, km, rid, m[0], m[1]);
        for (uint32_t w = 0; w < 2; w++) {
            const uint64_t op = op_key(rid, w);
            const bool in = part && op >= F.begin && op_inserts(F, item_hash(km, m[w]), op);
            run_keys[2 * (i - lo) + w] = in ? op : NO_OP;
            if (in && i < at) atomicAdd(&s_t, 1u);
        }
    }
    __syncthreads();                                      // (one workgroup: its own global writes are visible to it behind the barrier)
    const uint32_t t = s_t, n_keys = 2 * (hi - lo);
    for (uint32_t e = threadIdx.x; e < n_keys; e += A10_TPB) {
        const unsigned long long key = run_keys[e];
        if (key == NO_OP) continue;
        uint32_t smaller = 0;
        for (uint32_t q = 0; q < n_keys; q++) smaller += run_keys[q] < key ? 1u : 0u;
        if (smaller == t) *out_op = key;
    }
}

#ifdef SYLPH_A10_DEBUG   // (not in the product build: `make CXXFLAGS+=-DSYLPH_A10_DEBUG` for tools/a10_debug.py / a10_dump.py)
// debug aid (SYLPH_HIP_A10_TRACE): [0] occurrences taking part, [1] operations of the phase, [2] inserting, [3] first_op < op, [4] first_op > op
__global__ __launch_bounds__(A10_TPB) void a10_debug_kernel(Occs o, Filters F, unsigned long long* __restrict__ out) {
    const uint32_t i = blockIdx.x * A10_TPB + threadIdx.x;
    if (i >= o.n) return;
    uint64_t km, rid, m[2];
    if (!takes_part(o, i, km, rid, m[0], m[1])) return;
    atomicAdd(&out[0], 1ull);
    const Filter& cur = F.f[F.n - 1];
    for (uint32_t w = 0; w < 2; w++) {
        const uint64_t op = op_key(rid, w);
        if (op < F.begin) continue;
        atomicAdd(&out[1], 1ull);
        const uint64_t h = item_hash(km, m[w]);
        const uint64_t fo = table_first_op(cur, reduced_key(cur, h));
        atomicAdd(&out[fo == op ? 2 : (fo < op ? 3 : 4)], 1ull);
    }
}
#endif


// ---- the partitioned pass (round 5): one filter, no device-wide atomics -------------------------------------------------------
// While the sample stays inside filter 0 (at most `capacity` operations: 1 Gbp of pairs has 8 M against the reference's 10^7),
// "first operation of its class" needs no table at all: the operations are SORTED by class — the two-level partition that groups the
// occurrences by k-mer hash for the replay (partition.h) moves one 8-byte word per operation, {upper 32 bits of the mixed class key |
// operation id = 2 x slot + marker}, into ~128-operation buckets of equal class-hash ranges — and one workgroup per bucket finds the
// words that share their upper half with another word (exact duplicates: 2-4 % of a sample; chance: 2^-16 per pair of a bucket).
// Only those fetch their occurrence record, compute the exact class key and their place in the walk (op_key), and the ones that are
// not the earliest of their class set RID_A10_BIT — an atomic OR on ~3 % of the records instead of two atomics and a load per
// operation on a 256 MB table.  Skipped mate-2 occurrences (sketch.rs:852) take part here although the walk never sees them: their
// items are those of the mate-1 occurrence of the same k-mer in the same pair, which precedes them — they are never the first of a
// class, change no other operation's answer, and the replay drops them before it looks at the mark.
// What the pass cannot do is tell by itself that one filter was enough: it leaves {buckets too full for the workgroup, operations
// found} in two device words, the caller reads them with whatever it reads anyway (finish_bucketed's tail block), and a sample with
// more operations than the capacity — or thousands of copies of one item in one bucket — is marked again by the phase walk above.
constexpr int RES_CAP = 256, RES_TPB = 128;
constexpr uint32_t OPS_BLK_PER_TILE = 8;           // two words per slot: half the blocks per partition tile

struct OpsIn { const uint64_t* hash; const OccRec* recs; const uint32_t* blk_count; uint32_t n_dense, n_blk, slot_cap; int slotted; };

__device__ __forceinline__ uint64_t op_word(const Filter& d, uint64_t km, uint64_t marker, uint32_t opid) {
    return ((reduced_key(d, item_hash(km, marker)) * GOLD) & 0xFFFFFFFF00000000ull) | opid;
}
__device__ __forceinline__ ulonglong2 op_words(const Filter& d, const OccRec& r, bool valid, uint32_t slot) {
    if (!valid || !(r.rid & RID_MARKER_BIT)) return make_ulonglong2(INVALID_HASH, INVALID_HASH);
    return make_ulonglong2(op_word(d, r.hash, r.m0, 2u * slot), op_word(d, r.hash, r.m1, 2u * slot + 1u));
}
// ops[2 * slot + w] for every occurrence: slots of the session's one batch (one wavefront per block of the seeding kernel), or the dense arrays
__global__ __launch_bounds__(64) void a10_ops_slots_kernel(OpsIn in, Filter d, ulonglong2* __restrict__ ops, uint32_t* __restrict__ tail) {
    if (blockIdx.x == 0 && threadIdx.x < 2) tail[threadIdx.x] = 0;
    const uint32_t b = blockIdx.x, cnt = min(in.blk_count[b], in.slot_cap), g0 = b * in.slot_cap;
    for (uint32_t i = threadIdx.x; i < cnt; i += 64) ops[g0 + i] = op_words(d, in.recs[g0 + i], true, g0 + i);
}
// The same for a whole partition TILE of the slots (in.blk_per_tile blocks of the seeding kernel) per workgroup, counting the words per
// class range on the way: hist[c * n_tiles + t] is what part_hist_kernel would find in a pass of its own over the 64 MB of words
// (round 6, one-level pass only; eight groups of 32 lanes walk eight blocks at a time, as partition.h's for_tile_entries does)
__global__ __launch_bounds__(PART_TPB) void a10_ops_tile_kernel(OpsIn in, Filter d, uint32_t blk_per_tile, BucketMap bm, uint32_t n_tiles, ulonglong2* __restrict__ ops,
                                                                uint32_t* __restrict__ hist, uint32_t* __restrict__ tail) {
    __shared__ uint32_t s_h[MAX_COARSE];
    if (blockIdx.x == 0 && threadIdx.x < 2) tail[threadIdx.x] = 0;
    const uint32_t t = xcd_tile(n_tiles), C = bm.B;
    if (t >= n_tiles) return;
    for (uint32_t c = threadIdx.x; c < C; c += PART_TPB) s_h[c] = 0;
    __syncthreads();
    const uint32_t grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    for (uint32_t bl = grp; bl < blk_per_tile; bl += PART_TPB / 32) {
        const uint32_t b = t * blk_per_tile + bl;
        if (b >= in.n_blk) break;
        const uint32_t cnt = min(in.blk_count[b], in.slot_cap), g0 = b * in.slot_cap;
        for (uint32_t i = l; i < cnt; i += 32) {
            const ulonglong2 w = op_words(d, in.recs[g0 + i], true, g0 + i);
            ops[g0 + i] = w;
            if (w.x != INVALID_HASH) {
                atomicAdd(&s_h[bucket_of_key((uint32_t)(w.x >> 32), bm)], 1u);
                atomicAdd(&s_h[bucket_of_key((uint32_t)(w.y >> 32), bm)], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < C; c += PART_TPB) hist[(size_t)c * n_tiles + t] = s_h[c];
}
__global__ __launch_bounds__(256) void a10_ops_dense_kernel(OpsIn in, Filter d, ulonglong2* __restrict__ ops, uint32_t* __restrict__ tail) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2) tail[i] = 0;
    if (i >= in.n_dense) return;
    ops[i] = op_words(d, in.recs[i], in.hash[i] != INVALID_HASH, i);
}

// One workgroup = one bucket of class hashes.  Sub-range counting sort as in the replay (the words are uniform over the bucket's
// range: RES_CAP sub-ranges hold half a word each, equal upper halves share one); candidates = words with an equal upper half in
// their sub-range; those resolve through their records.
__global__ __launch_bounds__(RES_TPB) void a10_resolve_kernel(const uint64_t* __restrict__ sorted, const uint32_t* __restrict__ boff, BucketMap bm,
                                                              Filter d, OccRec* __restrict__ recs, uint32_t* __restrict__ tail) {
    constexpr int ITEMS = RES_CAP / RES_TPB;
    __shared__ uint64_t s_w[RES_CAP], s_g[RES_CAP], s_k[RES_CAP];
    __shared__ uint32_t s_cnt[RES_CAP + 1];
    __shared__ __attribute__((aligned(8))) uint16_t s_fill[RES_CAP];
    __shared__ uint32_t s_wave[RES_TPB / 64];
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    const uint32_t first = boff[b], n = boff[b + 1] - first;
    if (b == 0 && tid == 0) tail[1] = boff[bm.B];             // operations found (every valid word of the layout)
    if (n == 0) return;
    if (n > (uint32_t)RES_CAP) { if (tid == 0) atomicAdd(&tail[0], 1u); return; }
    const uint32_t lo_key = (uint32_t)((((uint64_t)b << 32) + bm.mult - 1u) / bm.mult);      // smallest 32-bit key of the bucket
    const uint32_t sub_mult = bm.sub_mult[0];
    uint64_t h[ITEMS];
    uint32_t sub[ITEMS], place[ITEMS];
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RES_TPB;
        h[q] = i < n ? sorted[first + i] : 0ull;
    }
    for (uint32_t t = tid; t <= (uint32_t)RES_CAP; t += RES_TPB) s_cnt[t] = 0;
    for (uint32_t t = tid; t < (uint32_t)RES_CAP; t += RES_TPB) s_fill[t] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RES_TPB;
        sub[q] = 0;
        if (i < n) {
            const uint32_t res = (uint32_t)(h[q] >> 32) - lo_key;
            sub[q] = sub_mult ? min(__umulhi(res, sub_mult), (uint32_t)RES_CAP - 1u) : min(res, (uint32_t)RES_CAP - 1u);
            atomicAdd(&s_cnt[sub[q]], 1u);
        }
    }
    __syncthreads();
    {
        uint32_t v[ITEMS], sum = 0;
#pragma unroll
        for (int e = 0; e < ITEMS; e++) { v[e] = s_cnt[tid * ITEMS + e]; sum += v[e]; }
        uint32_t run = block_excl_sum<RES_TPB>(sum, s_wave, nullptr);
#pragma unroll
        for (int e = 0; e < ITEMS; e++) { s_cnt[tid * ITEMS + e] = run; run += v[e]; }
        if (tid == RES_TPB - 1) s_cnt[RES_CAP] = run;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RES_TPB;
        place[q] = 0;
        if (i < n) {
            uint32_t* const w = reinterpret_cast<uint32_t*>(s_fill) + (sub[q] >> 1);
            const uint32_t old = atomicAdd(w, (sub[q] & 1u) ? 65536u : 1u);
            place[q] = s_cnt[sub[q]] + ((sub[q] & 1u) ? (old >> 16) : (old & 0xFFFFu));
            s_w[place[q]] = h[q];
        }
    }
    __syncthreads();
    bool cand[ITEMS];
    uint64_t g[ITEMS], key[ITEMS];
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const uint32_t i = tid + q * RES_TPB;
        cand[q] = false;
        g[q] = ~0ull;
        key[q] = 0;
        if (i < n) {
            const uint32_t lo = s_cnt[sub[q]], hi = s_cnt[sub[q] + 1], up = (uint32_t)(h[q] >> 32);
            for (uint32_t p = lo; p < hi; p++) cand[q] |= p != place[q] && (uint32_t)(s_w[p] >> 32) == up;
            if (cand[q]) {
                const uint32_t opid = (uint32_t)h[q];
                const OccRec r = recs[opid >> 1];
                g[q] = reduced_key(d, item_hash(r.hash, (opid & 1u) ? r.m1 : r.m0));      // < 2^63: never the ~0 of a word that is no candidate
                key[q] = op_key(r.rid, opid & 1u);
            }
            s_g[place[q]] = g[q];
            s_k[place[q]] = key[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        if (!cand[q]) continue;
        const uint32_t lo = s_cnt[sub[q]], hi = s_cnt[sub[q] + 1];
        bool contained = false;
        for (uint32_t p = lo; p < hi; p++) contained |= s_g[p] == g[q] && s_k[p] < key[q];
        // (the two operations of an occurrence may both be contained, from two workgroups: the OR commutes; readers of the rid in
        //  this kernel look at its record and rank bits only)
        if (contained) atomicOr(reinterpret_cast<uint32_t*>(&recs[(uint32_t)h[q] >> 1].rid) + 1, (uint32_t)(RID_A10_BIT >> 32));
    }
}

// ---- one partition level (round 6) ------------------------------------------------------------------------------------------
// The fine level above costs a read and a write of every word (part_fine) before the resolve kernel reads them a third time.
// Here the words stay where part_scatter left them, grouped by coarse range (~8,000 words), and the words that share their upper half
// with another one are found WITHOUT sorting them.  The upper halves are uniform over the range, so `(upper half - lowest of the
// range) x slots / width` is a direct-mapped slot of a bit table in LDS:
//   pass 1  every word sets its slot's bit; a word that finds it set is a SECOND ARRIVAL (every duplicate but one of each group, and by
//           chance words / 2 slots = 1.5 % of the rest): its upper half goes into a small open-addressing set;
//   pass 2  every word asks the set for its upper half: the ones found are the candidates (all words with an equal upper half — true
//           duplicates and the filter's cross-item collisions — plus the chance arrivals), ~7 % of the words;
//   then    candidates fetch their record, compute the exact class key and their place in the walk, and meet in a table of {class key,
//           earliest place} that overlays the bit table (64-bit LDS compare-and-swap + minimum); the ones that are not the earliest
//           of their class set RID_A10_BIT.
// `split` workgroups share a range: each reads all of its words and keeps the ones whose slot falls into its part of the slot space
// (2^17 slots per workgroup) — twice the reads, from L2, for half the LDS.  The footprint is what decides in the pipeline: the next
// samples' seeding kernels keep every CU full (5 workgroups x ~30 KB of LDS), and a kernel of this pass gets a workgroup in only
// where one of theirs retires.  Measured (profiles/r06_ab_a10.txt): 1024 threads + 68 KB per workgroup was 60 us faster than round 5's
// two levels alone on the GPU and 2 % SLOWER in the pipeline; 256 threads + 48 KB 2 % faster; holding the words in registers between
// the passes (112 VGPRs) slower again.
// A workgroup with more second arrivals or candidates than its lists take (heavily duplicated samples) cuts its part into P slices,
// resolved one after the other (equal classes share a slot); a slice that is still too full — ~a thousand copies of one item —
// bumps the verdict word as the two-level pass does.
constexpr uint32_t RNG_SLOT_BITS = 17, RNG_SLOTS = 1u << RNG_SLOT_BITS;     // per workgroup, one bit each: 16 KB
constexpr uint32_t RNG_CAND = 512, RNG_TAB = 1024, RNG_TAB_BITS = 10;       // candidates per slice, table entries (both tables)
constexpr uint32_t RNG_WORDS = 8192;                                       // words per range aimed for
constexpr uint32_t RNG_FREE = 0xFFFFFFFFu;
static_assert(RNG_TAB * 16 <= RNG_SLOTS / 8 && RNG_TAB == 1u << RNG_TAB_BITS, "the class table overlays the bits");

template <int TPB>
__global__ __launch_bounds__(TPB) void a10_range_kernel(const uint64_t* __restrict__ words, const uint32_t* __restrict__ cbase, uint32_t C, uint32_t mult,
                                                        uint32_t slot_mult, uint32_t split, Filter d, OccRec* __restrict__ recs, uint32_t* __restrict__ tail) {
    constexpr int Q = RNG_CAND / TPB, U = 4;
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[RNG_SLOTS / 32];
    __shared__ uint32_t s_sec[RNG_CAND], s_up[RNG_TAB], s_list[RNG_CAND];
    __shared__ uint32_t s_n[2];
    // Workgroups go round-robin to the 8 XCDs (each with its own L2): the `split` workgroups of a range are 8 apart in the grid — the
    // same XCD, one right behind the other — so that the second one's reads of the range's words are L2 hits
    const uint32_t tid = threadIdx.x, c = (blockIdx.x / (8u * split)) * 8u + (blockIdx.x & 7u), part = (blockIdx.x >> 3) % split;
    if (blockIdx.x == 0 && tid == 0) tail[1] = cbase[C];      // operations found
    if (c >= C) return;                                       // (padding of the grid to whole groups of 8 ranges)
    const uint32_t lo = cbase[c], hi = cbase[c + 1];
    if (hi == lo) return;
    const uint32_t lo_key = (uint32_t)((((uint64_t)c << 32) + mult - 1u) / mult);      // smallest upper half of the range
    const uint32_t all_slots = split << RNG_SLOT_BITS;
    unsigned long long* const t_key = reinterpret_cast<unsigned long long*>(s_bits);              // [RNG_TAB] class keys (~0: free)
    unsigned long long* const t_min = reinterpret_cast<unsigned long long*>(s_bits) + RNG_TAB;    // [RNG_TAB] earliest place of the class
    // appends v to list[0 .. RNG_CAND) behind *counter: one returning LDS atomic for the few lanes that take (3-7 % of them: a ballot +
    // prefix count per visit, for every lane, was a third of the kernel's 45 VALU + 48 SALU instructions per word visited)
    auto append = [&](bool take, uint32_t v, uint32_t* list, uint32_t* counter) {
        if (take) {
            const uint32_t at = atomicAdd(counter, 1u);
            if (at < RNG_CAND) list[at] = v;
        }
    };
    // body(valid, word) for every word of the range, U loads in flight per lane
    auto for_words = [&](auto&& body) {
        for (uint32_t e0 = lo; e0 < hi; e0 += U * TPB) {
            uint64_t w[U];
#pragma unroll
            for (int u = 0; u < U; u++) { const uint32_t e = e0 + u * TPB + tid; w[u] = e < hi ? words[e] : 0ull; }
#pragma unroll
            for (int u = 0; u < U; u++) body(e0 + u * TPB + tid < hi, w[u]);
        }
    };
    uint32_t P = 1;
    // slot of a word in this workgroup's table, or RNG_FREE: another workgroup's (or another slice's) word
    auto my_slot = [&](uint32_t res, uint32_t p) {
        const uint32_t s = min(slot_mult ? __umulhi(res, slot_mult) : res, all_slots - 1u);
        if ((s >> RNG_SLOT_BITS) != part) return RNG_FREE;
        const uint32_t l = s & (RNG_SLOTS - 1u);
        return (P == 1 || (uint32_t)(((uint64_t)l * P) >> RNG_SLOT_BITS) == p) ? l : RNG_FREE;
    };
    for (uint32_t p = 0; p < P; p++) {
        for (uint32_t i = tid; i < RNG_SLOTS / 32; i += TPB) s_bits[i] = 0;
        for (uint32_t i = tid; i < RNG_TAB; i += TPB) s_up[i] = RNG_FREE;
        if (tid < 2) s_n[tid] = 0;
        __syncthreads();
        for_words([&](bool valid, uint64_t w) {
            bool second = false;
            const uint32_t res = min((uint32_t)(w >> 32) - lo_key, RNG_FREE - 1u);
            if (valid) {
                const uint32_t l = my_slot(res, p);
                if (l != RNG_FREE) {
                    const uint32_t bit = 1u << (l & 31u);
                    second = atomicOr(&s_bits[l >> 5], bit) & bit;
                }
            }
            append(second, res, s_sec, &s_n[0]);
        });
        __syncthreads();
        const uint32_t n_sec = s_n[0];
        bool over = n_sec > RNG_CAND;
        uint32_t n_over = n_sec * 2u;
        if (!over) {
            for (uint32_t i = tid; i < n_sec; i += TPB) {
                const uint32_t res = s_sec[i];
                uint32_t at = (res * 0x9E3779B1u) >> (32 - RNG_TAB_BITS);
                for (;;) {
                    const uint32_t old = atomicCAS(&s_up[at], RNG_FREE, res);
                    if (old == RNG_FREE || old == res) break;
                    at = (at + 1u) & (RNG_TAB - 1u);
                }
            }
            __syncthreads();
            for_words([&](bool valid, uint64_t w) {
                bool cand = false;
                if (valid) {
                    const uint32_t res = min((uint32_t)(w >> 32) - lo_key, RNG_FREE - 1u);
                    if (my_slot(res, p) != RNG_FREE) {
                        uint32_t at = (res * 0x9E3779B1u) >> (32 - RNG_TAB_BITS);
                        for (;;) {
                            const uint32_t v = s_up[at];
                            if (v == res) { cand = true; break; }
                            if (v == RNG_FREE) break;
                            at = (at + 1u) & (RNG_TAB - 1u);
                        }
                    }
                }
                append(cand, (uint32_t)w, s_list, &s_n[1]);
            });
            __syncthreads();
            over = s_n[1] > RNG_CAND;
            n_over = s_n[1];
        }
        if (over) {
            // slices of about half the lists (the slots are uniform: 11 sigma of room), from the top; a slice that is still too full holds
            // many copies of ONE item — narrower slices twice more (marks set so far are set again: the OR is idempotent), then the verdict
            if (P < 1024u) {
                P = P == 1 ? min((n_over + RNG_CAND / 2 - 1) / (RNG_CAND / 2), 4096u) : min(P * 8u, 4096u);
                p = ~0u;                                                   // (p++ makes it 0)
                __syncthreads();
                continue;
            }
            if (tid == 0) atomicAdd(&tail[0], 1u);                         // more copies of one item than the lists take: the walk takes the sample
            return;
        }
        const uint32_t nc = s_n[1];
        for (uint32_t i = tid; i < RNG_TAB; i += TPB) { t_key[i] = ~0ull; t_min[i] = ~0ull; }
        __syncthreads();
        uint64_t key[Q];
        uint32_t opid[Q], at[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const uint32_t i = tid + q * TPB;
            key[q] = 0; opid[q] = 0; at[q] = 0;
            if (i < nc) {
                opid[q] = s_list[i];
                const OccRec r = recs[opid[q] >> 1];
                const uint64_t g = reduced_key(d, item_hash(r.hash, (opid[q] & 1u) ? r.m1 : r.m0));      // < 2^63: never the ~0 of a free entry
                key[q] = op_key(r.rid, opid[q] & 1u);
                at[q] = (uint32_t)((g * GOLD) >> 40) & (RNG_TAB - 1u);
                for (;;) {
                    const unsigned long long old = atomicCAS(&t_key[at[q]], ~0ull, (unsigned long long)g);
                    if (old == ~0ull || old == g) break;
                    at[q] = (at[q] + 1u) & (RNG_TAB - 1u);
                }
                atomicMin(&t_min[at[q]], (unsigned long long)key[q]);
            }
        }
        __syncthreads();
        // (the two operations of an occurrence may both be contained, from two workgroups: the OR commutes; readers of the rid in
        //  this kernel look at its record and rank bits only)
#pragma unroll
        for (int q = 0; q < Q; q++)
            if (tid + q * TPB < nc && t_min[at[q]] < key[q])
                atomicOr(reinterpret_cast<uint32_t*>(&recs[opid[q] >> 1].rid) + 1, (uint32_t)(RID_A10_BIT >> 32));
        __syncthreads();
    }
}

}  // namespace

static Filter filter_geometry(const sylph_sketch* sk, int j) {
    // (as CuckooFilter::init / ScalableCuckoo::grow of the tests' CPU checker)
    Filter f{};
    const uint64_t cap = sk->dedup_capacity << j;
    const double fpr = sk->dedup_fpr * std::pow(0.9, (double)j);
    int fp_bits = (int)std::ceil(std::log2(1.0 / fpr) + std::log2(8.0));
    fp_bits = std::min(31, std::max(1, fp_bits));
    uint64_t n_buckets = 1;
    while (n_buckets * 4 < cap) n_buckets <<= 1;
    SY_REQUIRE(n_buckets <= (1ull << 31), "filter of %llu buckets", (unsigned long long)n_buckets);
    f.fp_mask = (uint32_t)((1ull << fp_bits) - 1);
    f.nb_mask = (uint32_t)(n_buckets - 1);
    f.cut = NO_OP;
    return f;
}

// The partitioned pass over the session's occurrences where they lie (slots of its one batch, or the dense arrays).
static void a10_mark_partitioned(sylph_sketch* sk) {
    sylph_ctx* ctx = sk->ctx;
    const bool slotted = sk->pend.live;
    const bool deferred = slotted && sk->pend.deferred;
    const uint64_t n_slots = slotted ? (uint64_t)sk->pend.n_blk * sk->pend.slot_cap : sk->n_occ;
    const uint64_t n_expect = deferred ? std::max<uint32_t>(1, sk->pend.n_expect) : (slotted ? sk->pend.n : sk->n_occ);
    HostPhase ph(ctx, "finish: a10 filter marks (partitioned)");
    sk->a10_tail.reserve(16);
    uint32_t* tail = sk->a10_tail.as<uint32_t>();
    DevBuf b_ops(ctx), b_pairs(ctx), b_sorted(ctx), b_hist(ctx), b_boff(ctx);
    b_ops.reserve(n_slots * 16);
    b_pairs.reserve(n_slots * 16);
    const Filter f0 = filter_geometry(sk, 0);
    // buckets: equal ranges of the words' upper 32 bits, ~bucket_target operations each
    static const int env_levels = [] { const char* e = getenv("SYLPH_HIP_A10_LEVELS"); return e ? atoi(e) : 1; }();
    const bool one_level = env_levels != 2;
    static const uint32_t env_words = [] { const char* e = getenv("SYLPH_HIP_A10_RANGE_WORDS"); return e ? (uint32_t)std::max(512, std::min(1 << 20, atoi(e))) : RNG_WORDS; }();
    const uint32_t B = one_level ? (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, 2 * n_expect / env_words), MAX_COARSE)
                                 : (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, 2 * n_expect / ctx->bucket_target), 1u << 24);
    BucketMap bm{};
    bm.sh = 32;
    bm.mult = B;                                                   // (B << 32) / 2^32
    bm.B = B;
    bm.composite = 1;
    const uint64_t range = (0x100000000ull + B - 1) / B + 1;       // widest bucket in key units
    bm.range_hs = (uint32_t)std::min<uint64_t>(range, 0xFFFFFFFFull);
    bm.sub_mult[0] = range > (uint64_t)RES_CAP ? (uint32_t)(((uint64_t)RES_CAP << 32) / range) : 0u;
    if (!one_level) b_sorted.reserve(n_slots * 16);
    const PartGeom geom = one_level ? PartGeom{0, B} : part_geometry(B);
    OpsIn oin{};
    PartIn in{};
    in.key_sh = 32;
    in.carry = 1;
    uint32_t n_tiles;
    if (slotted) {
        const uint32_t* blk_count = sk->slot_meta.as<uint32_t>() + (sk->pend.n_blk + 1);      // (layout: reads.hip SlotMeta)
        oin.recs = sk->slot_rec.as<OccRec>(); oin.blk_count = blk_count; oin.n_blk = sk->pend.n_blk; oin.slot_cap = sk->pend.slot_cap; oin.slotted = 1;
        static const uint32_t env_bpt = [] { const char* e = getenv("SYLPH_HIP_A10_BLK_PER_TILE"); return e ? (uint32_t)std::max(1, std::min(32, atoi(e))) : OPS_BLK_PER_TILE; }();
        in.slotted = 2; in.blk_count = blk_count; in.n_blk = sk->pend.n_blk; in.slot_cap = sk->pend.slot_cap; in.blk_per_tile = env_bpt;
        n_tiles = (in.n_blk + env_bpt - 1) / env_bpt;
    } else {
        oin.hash = sk->hash.as<uint64_t>(); oin.recs = sk->recs.as<OccRec>(); oin.n_dense = (uint32_t)sk->n_occ;
        in.slotted = 0; in.n_dense = (uint32_t)(2 * sk->n_occ);
        in.tile_entries = (uint32_t)std::max<uint64_t>(4096, (2 * sk->n_occ + 65535) / 65536);
        n_tiles = (uint32_t)((2 * sk->n_occ + in.tile_entries - 1) / in.tile_entries);
    }
    static const uint32_t env_stage = [] { const char* e = getenv("SYLPH_HIP_A10_STAGE_PAIRS"); return e ? (uint32_t)std::max(256, std::min(8192, atoi(e))) : 0u; }();
    if (one_level) in.stage_pairs = env_stage;
    in.hash = b_ops.as<uint64_t>();
    b_hist.reserve(part_hist_words(geom, n_tiles) * 4);
    b_boff.reserve(((size_t)B + 2) * 4);
    ScopedKernelTimer t(ctx, "a10");
    static const bool env_fused_hist = [] { const char* e = getenv("SYLPH_HIP_A10_FUSED_HIST"); return !e || atoi(e) != 0; }();
    const bool fused_hist = slotted && one_level && env_fused_hist;         // the operation words' kernel also counts them per (range, tile)
    if (fused_hist)
        hipLaunchKernelGGL(a10_ops_tile_kernel, dim3(((n_tiles + 7) / 8) * 8), dim3(PART_TPB), 0, ctx->stream, oin, f0, in.blk_per_tile, bm, n_tiles, b_ops.as<ulonglong2>(),
                           b_hist.as<uint32_t>(), tail);
    else if (slotted) hipLaunchKernelGGL(a10_ops_slots_kernel, dim3(oin.n_blk), dim3(64), 0, ctx->stream, oin, f0, b_ops.as<ulonglong2>(), tail);
    else hipLaunchKernelGGL(a10_ops_dense_kernel, dim3((uint32_t)((sk->n_occ + 255) / 256)), dim3(256), 0, ctx->stream, oin, f0, b_ops.as<ulonglong2>(), tail);
    if (one_level) {
        // words grouped by range where the scatter left them; slot of a word inside its range = (upper half - lowest of the range) x slots / width
        launch_partition_coarse(ctx, in, bm, n_tiles, b_hist.as<uint32_t>(), b_pairs.as<uint2>(), fused_hist);
        static const int env_split = [] { const char* e = getenv("SYLPH_HIP_A10_RANGE_SPLIT"); return e ? std::max(1, std::min(8, atoi(e))) : 2; }();
        static const int env_pad = [] { const char* e = getenv("SYLPH_HIP_A10_RANGE_LDS_PAD"); return e ? atoi(e) : 0; }();     // (A/B: LDS footprint)
        const uint32_t split = (uint32_t)env_split;
        const uint64_t all_slots = (uint64_t)split << RNG_SLOT_BITS;
        const uint32_t slot_mult = range > all_slots ? (uint32_t)((all_slots << 32) / range) : 0u;
        hipLaunchKernelGGL(a10_range_kernel<256>, dim3(((B + 7) / 8) * 8 * split), dim3(256), (size_t)env_pad, ctx->stream, b_pairs.as<uint64_t>(),
                           part_cbase(b_hist.as<uint32_t>(), B, n_tiles), B, bm.mult, slot_mult, split, f0, const_cast<OccRec*>(oin.recs), tail);
    } else {
        launch_partition(ctx, in, bm, geom, n_tiles, 2 * n_expect, b_hist.as<uint32_t>(), b_pairs.as<uint2>(), b_boff.as<uint32_t>(), nullptr,
                         b_sorted.as<uint64_t>(), nullptr, 0, nullptr, nullptr, nullptr, nullptr);
        hipLaunchKernelGGL(a10_resolve_kernel, dim3(B), dim3(RES_TPB), 0, ctx->stream, b_sorted.as<uint64_t>(), b_boff.as<uint32_t>(), bm, f0,
                           const_cast<OccRec*>(oin.recs), tail);
    }
    SY_HIP(hipGetLastError());
    sk->a10_state = 1;
    if (ctx->profile) ctx->stats["a10_part"].launches++;              // (tests ask sylph_ctx_kernel_stats which pass ran)
    // (the buffers go back to the pool here; stream order keeps them alive for the kernels queued above)
}

static void a10_mark_walk(sylph_sketch* sk);

// Which pass: the partitioned one wherever the first filter is sure (the occurrence count is known) or expected (deferred
// batches: the verdict words tell) to take every operation; the phase walk for samples that make the filter grow.
void a10_mark(sylph_sketch* sk) {
    if (!sk->filter_dedup() || sk->a10_state != 0) return;
    SY_REQUIRE(sk->dedup_capacity >= 1 && sk->dedup_capacity < (1ull << 31), "dedup_capacity out of range");
    static const int env_force = [] { const char* e = getenv("SYLPH_HIP_A10"); return !e ? 0 : !strcmp(e, "walk") ? 1 : !strcmp(e, "part") ? 2 : 0; }();
    const int force = sk->a10_force ? sk->a10_force : env_force;
    const bool slotted = sk->pend.live, deferred = slotted && sk->pend.deferred;
    const uint64_t n_slots = slotted ? (uint64_t)sk->pend.n_blk * sk->pend.slot_cap : sk->n_occ;
    const uint64_t n_ops = 2 * (deferred ? (uint64_t)sk->pend.n_expect + sk->pend.n_expect / 8 : (slotted ? sk->pend.n : sk->n_occ));
    const bool fits = n_slots < (1ull << 31) && n_ops <= sk->dedup_capacity && sk->n_plain == 0;
    if (n_slots == 0) { sk->a10_state = 2; return; }
    if (force != 1 && fits) a10_mark_partitioned(sk);
    else a10_mark_walk(sk);
}

// The verdict words of the partitioned pass, read by the caller together with its own tail: false = the marks were no good (a
// bucket with more copies of one class than a workgroup takes; more operations than the first filter holds) and the phase walk
// has marked the records again — on the dense arrays: whoever partitioned the slots starts over.
bool a10_verdict(sylph_sketch* sk, const uint32_t words[2]) {
    if (sk->a10_state != 1) return true;
    if (words[0] == 0 && (uint64_t)words[1] <= sk->dedup_capacity) { sk->a10_state = 2; return true; }
    if (sk->ctx->profile) sk->ctx->stats["a10_redo"].launches++;       // (tests ask sylph_ctx_kernel_stats whether this road was taken)
    a10_mark_walk(sk);
    return false;
}

void a10_settle(sylph_sketch* sk) {
    if (!sk->filter_dedup()) return;
    for (int round = 0; round < 4 && sk->a10_state != 2; round++) {
        if (sk->a10_state == 0) { a10_mark(sk); continue; }
        // the verdict of a deferred batch first: a batch that is redone loses its marks (redo_deferred_batch resets a10_state)
        if (sk->pend.live && sk->pend.deferred) { resolve_deferred_slots(sk); if (sk->a10_state != 1) continue; }
        uint32_t words[2] = {0, 0};
        sk->ctx->read_back(words, sk->a10_tail.p, 8);
        a10_verdict(sk, words);
    }
    SY_REQUIRE(sk->a10_state == 2, "internal: the filter marks did not settle");
}

// Marks (RID_A10_BIT) every occurrence of a paired session for which sketch.rs:747 / :754 would have found its (k-mer, markers)
// in the filter.  Works on the dense file-order arrays: occurrences still in their slots are compacted first.
static void a10_mark_walk(sylph_sketch* sk) {
    sylph_ctx* ctx = sk->ctx;
    flush_pending_slots(sk);
    sk->a10_state = 2;
    const uint64_t n = sk->n_occ;
    if (!n) return;
    SY_REQUIRE(n < (1ull << 32), "more than 2^32-1 seed occurrences in one sample");
    HostPhase ph(ctx, "finish: a10 filter marks");
    DevBuf b_part(ctx), b_run(ctx);
    b_part.reserve(n);
    Occs o{sk->hash.as<uint64_t>(), sk->recs.as<OccRec>(), (uint32_t)n, b_part.as<uint8_t>()};
    {
        ScopedKernelTimer t(ctx, "a10");
        hipLaunchKernelGGL(a10_part_kernel, dim3((uint32_t)((n + A10_TPB - 1) / A10_TPB)), dim3(A10_TPB), 0, ctx->stream, o, b_part.as<uint8_t>());
        SY_HIP(hipGetLastError());
    }
#ifdef SYLPH_A10_DEBUG
    if (const char* dump = getenv("SYLPH_HIP_A10_DUMP")) {      // debug aid: the occurrence records the filter pass works on
        std::vector<OccRec> h(n);
        ctx->d2h(h.data(), o.recs, n * sizeof(OccRec));
        std::vector<uint64_t> hh(n);
        ctx->d2h(hh.data(), o.hash, n * 8);
        if (FILE* f = fopen(dump, "wb")) { fwrite(hh.data(), 8, n, f); fwrite(h.data(), sizeof(OccRec), n, f); fclose(f); }
    }
#endif
    const uint64_t n_ops = 2 * n;                 // an upper bound of the operations (occurrences that take part x 2)
    uint64_t ops_before = 0;                      // ... and of those before the current phase: every closed filter took its capacity
    std::vector<std::unique_ptr<DevBuf>> tables;
    Filters F{};
    F.begin = 0;
    const uint32_t grid = (uint32_t)((n + A10_TPB - 1) / A10_TPB), n_tiles = (uint32_t)((n + TILE_OCC - 1) / TILE_OCC);
    DevBuf b_tiles(ctx);
    for (int j = 0;; j++) {
        SY_REQUIRE(j < MAX_FILTERS, "the approximate dedup would need more than %d filters: raise dedup_capacity", MAX_FILTERS);
        const uint64_t cap = sk->dedup_capacity << j;
        const Filter geo = filter_geometry(sk, j);
        // the phase's table: every operation behind the cut may enter it
        const uint64_t remaining = n_ops - std::min(n_ops, ops_before);
        uint64_t slots = 1024;
        while (slots < 2 * remaining) slots <<= 1;
        tables.emplace_back(new DevBuf(ctx));
        tables.back()->reserve(slots * sizeof(Ent));
        Filter& cur = F.f[j];
        cur.tab = tables.back()->as<Ent>();
        cur.tab_mask = (uint32_t)(slots - 1);
        cur.tab_shift = (uint32_t)(64 - (bit_length(slots) - 1));
        cur.fp_mask = geo.fp_mask;
        cur.nb_mask = geo.nb_mask;
        cur.cut = NO_OP;
        F.n = j + 1;
        F.end = NO_OP;
        ScopedKernelTimer t(ctx, "a10");
        SY_HIP(hipMemsetAsync(tables.back()->p, 0, slots * sizeof(Ent), ctx->stream));
        hipLaunchKernelGGL(a10_enter_kernel, dim3(grid), dim3(A10_TPB), 0, ctx->stream, o, F);
#ifdef SYLPH_A10_DEBUG
        if (const char* dump = getenv("SYLPH_HIP_A10_DUMP")) {
            if (j == 0) {
                std::vector<Ent> h(slots);
                ctx->d2h(h.data(), cur.tab, slots * sizeof(Ent));
                std::string fn = std::string(dump) + ".table";
                if (FILE* f = fopen(fn.c_str(), "wb")) { fwrite(h.data(), sizeof(Ent), slots, f); fclose(f); }
            }
        }
        if (getenv("SYLPH_HIP_A10_TRACE")) {
            DevBuf dbg(ctx);
            dbg.reserve(64);
            SY_HIP(hipMemsetAsync(dbg.p, 0, 64, ctx->stream));
            hipLaunchKernelGGL(a10_debug_kernel, dim3(grid), dim3(A10_TPB), 0, ctx->stream, o, F, dbg.as<unsigned long long>());
            unsigned long long hdbg[5];
            ctx->d2h(hdbg, dbg.p, 40);
            fprintf(stderr, "[sylph_hip] a10 phase %d: %llu occurrences take part, %llu operations, %llu inserting, %llu later than their class's first, %llu EARLIER than it\n",
                    j, hdbg[0], hdbg[1], hdbg[2], hdbg[3], hdbg[4]);
        }
#endif
        bool last = remaining <= cap;            // not even every remaining operation inserting would fill the filter
        if (!last) {
            b_tiles.reserve(((size_t)n_tiles + 2) * 4 + 16);
            hipLaunchKernelGGL(a10_count_kernel, dim3(n_tiles), dim3(A10_TPB), 0, ctx->stream, o, F, b_tiles.as<uint32_t>());
            SY_HIP(hipGetLastError());
            std::vector<uint32_t> h_tiles(n_tiles);
            ctx->d2h(h_tiles.data(), b_tiles.p, (size_t)n_tiles * 4);
            uint64_t seen = 0;
            uint32_t tile = n_tiles;
            for (uint32_t q = 0; q < n_tiles; q++) {
                if (seen + h_tiles[q] > cap) { tile = q; break; }      // the inserting operation number `cap` (0-based) lies here
                seen += h_tiles[q];
            }
            if (tile == n_tiles) last = true;
            else {
                uint64_t* d_cut = reinterpret_cast<uint64_t*>(b_tiles.as<uint32_t>() + ((n_tiles + 1) & ~1u));
                SY_HIP(hipMemsetAsync(d_cut, 0xFF, 8, ctx->stream));
                const uint32_t run_cap = (uint32_t)std::min<uint64_t>(2 * n, 1u << 22);     // operations of ONE record the cut can be looked for in
                b_run.reserve((size_t)run_cap * 8);
                hipLaunchKernelGGL(a10_find_kernel, dim3(1), dim3(A10_TPB), 0, ctx->stream, o, F, tile, (uint32_t)(cap - seen),
                                   b_run.as<unsigned long long>(), run_cap, d_cut);
                SY_HIP(hipGetLastError());
                uint64_t cut = 0;
                ctx->read_back(&cut, d_cut, 8);
                SY_REQUIRE(cut != NO_OP && cut > F.begin, "internal: the operation that opens filter %d was not found", j + 1);
                if (getenv("SYLPH_HIP_A10_TRACE"))
                    fprintf(stderr, "[sylph_hip] a10 phase %d: capacity %llu, %d fingerprint bits, %llu buckets; the next filter opens at record %llu, seed %llu, marker %llu (tile %u, %llu inserting operations before the tile)\n",
                            j, (unsigned long long)cap, bit_length(geo.fp_mask), (unsigned long long)geo.nb_mask + 1, (unsigned long long)(cut >> 21),
                            (unsigned long long)((cut >> 1) & RID_RANK_MAX), (unsigned long long)(cut & 1), tile, (unsigned long long)seen);
                cur.cut = cut;
                F.end = cut;
                ops_before += cap;
            }
        }
        hipLaunchKernelGGL(a10_flag_kernel, dim3(grid), dim3(A10_TPB), 0, ctx->stream, o, F);
        SY_HIP(hipGetLastError());
        if (last) break;
        F.begin = F.end;
    }
    // (the tables go back to the pool here; stream order keeps them alive for the kernels queued above)
}

}  // namespace sylph
