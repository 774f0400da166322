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

}  // namespace

// Marks (RID_A10_BIT) every occurrence of a paired session for which sketch.rs:747 / :754 would have found its (k-mer, markers)
// in the filter.  Works on the dense file-order arrays: occurrences still in their slots are compacted first.
void a10_mark(sylph_sketch* sk) {
    sylph_ctx* ctx = sk->ctx;
    flush_pending_slots(sk);
    const uint64_t n = sk->n_occ;
    if (!n) return;
    SY_REQUIRE(sk->dedup_capacity >= 1 && sk->dedup_capacity < (1ull << 31), "dedup_capacity out of range");
    HostPhase ph(ctx, "finish: a10 filter marks");
    DevBuf b_part(ctx), b_run(ctx);
    b_part.reserve(n);
    Occs o{sk->hash.as<uint64_t>(), sk->recs.as<OccRec>(), (uint32_t)n, b_part.as<uint8_t>()};
    {
        ScopedKernelTimer t(ctx, "a10");
        hipLaunchKernelGGL(a10_part_kernel, dim3((uint32_t)((n + A10_TPB - 1) / A10_TPB)), dim3(A10_TPB), 0, ctx->stream, o, b_part.as<uint8_t>());
        SY_HIP(hipGetLastError());
    }
    if (const char* dump = getenv("SYLPH_HIP_A10_DUMP")) {      // debug aid: the occurrence records the filter pass works on
        std::vector<OccRec> h(n);
        ctx->d2h(h.data(), o.recs, n * sizeof(OccRec));
        std::vector<uint64_t> hh(n);
        ctx->d2h(hh.data(), o.hash, n * 8);
        if (FILE* f = fopen(dump, "wb")) { fwrite(hh.data(), 8, n, f); fwrite(h.data(), sizeof(OccRec), n, f); fclose(f); }
    }
    const uint64_t n_ops = 2 * n;                 // an upper bound of the operations (occurrences that take part x 2)
    uint64_t ops_before = 0;                      // ... and of those before the current phase: every closed filter took its capacity
    std::vector<std::unique_ptr<DevBuf>> tables;
    Filters F{};
    F.begin = 0;
    const uint32_t grid = (uint32_t)((n + A10_TPB - 1) / A10_TPB), n_tiles = (uint32_t)((n + TILE_OCC - 1) / TILE_OCC);
    DevBuf b_tiles(ctx);
    for (int j = 0;; j++) {
        SY_REQUIRE(j < MAX_FILTERS, "the approximate dedup would need more than %d filters: raise dedup_capacity", MAX_FILTERS);
        // the filter's geometry (as CuckooFilter::init / ScalableCuckoo::grow of the tests' CPU checker)
        const uint64_t cap = sk->dedup_capacity << j;
        const double fpr = sk->dedup_fpr * std::pow(0.9, (double)j);
        int fp_bits = (int)std::ceil(std::log2(1.0 / fpr) + std::log2(8.0));
        fp_bits = std::min(31, std::max(1, fp_bits));
        uint64_t n_buckets = 1;
        while (n_buckets * 4 < cap) n_buckets <<= 1;
        SY_REQUIRE(n_buckets <= (1ull << 31), "filter of %llu buckets", (unsigned long long)n_buckets);
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
        cur.fp_mask = (uint32_t)((1ull << fp_bits) - 1);
        cur.nb_mask = (uint32_t)(n_buckets - 1);
        cur.cut = NO_OP;
        F.n = j + 1;
        F.end = NO_OP;
        ScopedKernelTimer t(ctx, "a10");
        SY_HIP(hipMemsetAsync(tables.back()->p, 0, slots * sizeof(Ent), ctx->stream));
        hipLaunchKernelGGL(a10_enter_kernel, dim3(grid), dim3(A10_TPB), 0, ctx->stream, o, F);
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
                            j, (unsigned long long)cap, fp_bits, (unsigned long long)n_buckets, (unsigned long long)(cut >> 21),
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
