// contain_index.h — the k-mer -> genomes "line index" of a database (shard) and the database object, shared by contain.hip
// (build, probe, result assembly) and shard.hip (multi-GPU exchange).  Internal, not part of the ABI.
#pragma once
#include "common.h"

namespace sylph {

// Slots per bucket line: 8 (a 64-byte line, lambda = 4 postings aimed at per line) or 4 (round 5: a 32-byte half-line, lambda = 2 —
// the same index bytes, twice the buckets; random 32 B reads come back at 1.8x the rate of random 64 B reads on this chip,
// profiles/r05_random_line_rates.txt).  A compile-time choice: the probe's loads and the slot scan are unrolled over it.
#ifndef SYLPH_LINE_SLOTS
#define SYLPH_LINE_SLOTS 8
#endif
constexpr int LINE_SLOTS = SYLPH_LINE_SLOTS;   // x 8 B per bucket line
static_assert(LINE_SLOTS == 8 || LINE_SLOTS == 4, "a bucket line is 64 or 32 bytes");
constexpr int LINE_QUADS = LINE_SLOTS / 2;     // 16-byte loads per line
constexpr uint32_t DEFAULT_INDEX_LAMBDA = LINE_SLOTS / 2;
constexpr uint64_t SLOT_EMPTY = ~0ull;

// What the kernels need to probe an index (POD, passed by value).
//   bucket(km) = (km - base) / div;   slot = ((km - base) % div) << gshift | flag << (gshift - 1) | genome id
//   descriptor (slot 7 of a bucket with more than 8 postings) = first entry of its overflow run << gshift | flag | min(length of
//   the run, 2^(gshift-1) - 1) in the genome field
struct LineView {
    const uint64_t* lines;   // n_buckets x LINE_SLOTS
    const uint64_t* ovf;     // overflow runs, each closed by SLOT_EMPTY
    uint64_t base, div, magic;   // magic = floor(2^64 / div) for div >= 2
    uint64_t n_ovf;          // entries of ovf (the wavefront-wide walk of a long run reads 64 at a time)
    uint32_t n_buckets;
    int gshift;
};

struct LineIndex {
    DevBuf lines, ovf;
    uint64_t base = 0, div = 1, magic = 0, n_postings = 0, n_ovf = 0;
    uint32_t n_buckets = 0;
    int gshift = 2;
    LineIndex() = default;
    explicit LineIndex(sylph_ctx* c) : lines(c), ovf(c) {}
    LineView view() const { return LineView{lines.as<uint64_t>(), ovf.as<uint64_t>(), base, div, magic, n_ovf, n_buckets, gshift}; }
};

// One sample table of a batch: n (k-mer, count) entries; chunk0 = index of its first 256-entry chunk in the batch.
struct SampleRef {
    const uint64_t* k;
    const uint32_t* c;
    uint64_t n;
    uint32_t chunk0, pad;
};

// Up to REFS_INLINE tables travel in the kernel arguments (no host->device copy of the descriptor array, no wait for it).
constexpr uint32_t REFS_INLINE = 8;
struct RefPack { SampleRef r[REFS_INLINE]; };

#ifdef __HIPCC__
// Calls f(genome id) for every posting of k-mer `km`.  One 64 B line read (4 x global_load_dwordx4); the overflow run of
// a crowded bucket is walked only when the 7 postings kept in the line do not already exceed the remainder looked for.
// LONG_RUN > 0: an overflow run of at least LONG_RUN entries is not walked here; its first entry is returned through
// *long_start (else ~0) with the remainder in *long_rem, for the caller's wavefront-wide walk (probe_long_run).
constexpr uint32_t LONG_RUN_MIN = 48;
template <class F>
__device__ __forceinline__ void for_each_posting(const LineView& v, uint64_t km, F&& f, uint64_t* long_start = nullptr,
                                                 uint64_t* long_rem = nullptr) {
    if (long_start) *long_start = ~0ull;
    if (km < v.base) return;
    const uint64_t x = km - v.base;
    uint64_t q, rem;
    if (v.div == 1) { q = x; rem = 0; }
    else {
        q = __umul64hi(x, v.magic);                       // floor(x * floor(2^64/div) / 2^64) is q or q - 1
        rem = x - q * v.div;
        if (rem >= v.div) { rem -= v.div; q++; }
    }
    if (q >= v.n_buckets) return;
    const uint4* lp = reinterpret_cast<const uint4*>(v.lines + q * LINE_SLOTS);
    uint4 qd[LINE_QUADS];
#pragma unroll
    for (int t = 0; t < LINE_QUADS; t++) qd[t] = lp[t];
    uint64_t s[LINE_SLOTS];
#pragma unroll
    for (int t = 0; t < LINE_QUADS; t++) { s[2 * t] = ((uint64_t)qd[t].y << 32) | qd[t].x; s[2 * t + 1] = ((uint64_t)qd[t].w << 32) | qd[t].z; }
    const uint64_t flag = 1ull << (v.gshift - 1), gmask = flag - 1;
#pragma unroll
    for (int j = 0; j < LINE_SLOTS; j++)
        if ((s[j] >> v.gshift) == rem && !(s[j] & flag)) f((uint32_t)(s[j] & gmask));
    const uint64_t last = s[LINE_SLOTS - 1];
    if ((last & flag) && last != SLOT_EMPTY && (s[LINE_SLOTS - 2] >> v.gshift) <= rem) {   // descriptor: rest of the bucket
        if (long_start && (last & gmask) >= LONG_RUN_MIN) { *long_start = last >> v.gshift; *long_rem = rem; return; }
        for (uint64_t i = last >> v.gshift;; i++) {
            const uint64_t y = v.ovf[i];
            if (y == SLOT_EMPTY) break;
            const uint64_t r = y >> v.gshift;
            if (r > rem) break;                            // the run is sorted by (remainder, genome)
            if (r == rem) f((uint32_t)(y & gmask));
        }
    }
}

// The same in two halves, so that a lane can have the lines of several k-mers in flight before it looks at any of them (the
// probe is latency-bound: random 64 B reads): line_fetch computes the bucket and issues the four loads, line_scan walks the
// slots (and the short overflow run) exactly as for_each_posting does.
struct LineFetch {
    uint4 v[LINE_QUADS];
    uint64_t rem;
    bool live;
};
__device__ __forceinline__ LineFetch line_fetch(const LineView& v, uint64_t km, bool wanted) {
    LineFetch f;
    f.live = false;
    f.rem = 0;
#pragma unroll
    for (int t = 0; t < LINE_QUADS; t++) f.v[t] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (!wanted || km < v.base) return f;
    const uint64_t x = km - v.base;
    uint64_t q, rem;
    if (v.div == 1) { q = x; rem = 0; }
    else {
        q = __umul64hi(x, v.magic);                       // floor(x * floor(2^64/div) / 2^64) is q or q - 1
        rem = x - q * v.div;
        if (rem >= v.div) { rem -= v.div; q++; }
    }
    if (q >= v.n_buckets) return f;
    const uint4* lp = reinterpret_cast<const uint4*>(v.lines + q * LINE_SLOTS);
#pragma unroll
    for (int t = 0; t < LINE_QUADS; t++) f.v[t] = lp[t];
    f.rem = rem;
    f.live = true;
    return f;
}
template <class F>
__device__ __forceinline__ void line_scan(const LineView& v, const LineFetch& l, F&& f, uint64_t* long_start, uint64_t* long_rem) {
    *long_start = ~0ull;
    if (!l.live) return;
    const uint64_t rem = l.rem;
    uint64_t s[LINE_SLOTS];
#pragma unroll
    for (int t = 0; t < LINE_QUADS; t++) { s[2 * t] = ((uint64_t)l.v[t].y << 32) | l.v[t].x; s[2 * t + 1] = ((uint64_t)l.v[t].w << 32) | l.v[t].z; }
    const uint64_t flag = 1ull << (v.gshift - 1), gmask = flag - 1;
#pragma unroll
    for (int j = 0; j < LINE_SLOTS; j++)
        if ((s[j] >> v.gshift) == rem && !(s[j] & flag)) f((uint32_t)(s[j] & gmask));
    const uint64_t last = s[LINE_SLOTS - 1];
    if ((last & flag) && last != SLOT_EMPTY && (s[LINE_SLOTS - 2] >> v.gshift) <= rem) {   // descriptor: rest of the bucket
        if ((last & gmask) >= LONG_RUN_MIN) { *long_start = last >> v.gshift; *long_rem = rem; return; }
        for (uint64_t i = last >> v.gshift;; i++) {
            const uint64_t y = v.ovf[i];
            if (y == SLOT_EMPTY) break;
            const uint64_t r = y >> v.gshift;
            if (r > rem) break;                            // the run is sorted by (remainder, genome)
            if (r == rem) f((uint32_t)(y & gmask));
        }
    }
}
#endif

// Layout of the result block, identical on the device (one buffer, ONE device->host copy) and in pinned host memory:
// [cov_off (R+1) u64 | contain_count R u32 | covs n_hits x width | pad to 8 | kmers_lost G u32 (reassign only, second copy)]
// with R = n_samples x n_genomes rows (row = sample * n_genomes + genome).
struct ResultLayout {
    size_t ccount = 0, covs = 0, lost = 0, end = 0;
    ResultLayout() = default;
    ResultLayout(uint64_t R, uint64_t n_hits, uint32_t width, uint64_t lost_entries)
        : ccount((R + 1) * 8), covs(ccount + R * 4), lost((covs + n_hits * width + 7) & ~(size_t)7), end(lost + lost_entries * 4) {}
};

// A block of page-locked host memory that results are copied into (grow-only).
struct HostBlock {
    void* p = nullptr;
    size_t cap = 0;
    ResultLayout lay;            // layout of the result block it holds
    HostBlock() = default;
    HostBlock(const HostBlock&) = delete;
    HostBlock& operator=(const HostBlock&) = delete;
    ~HostBlock() { if (p) (void)hipHostFree(p); }
    void ensure(size_t need) {
        if (need <= cap) return;
        if (p) SY_HIP(hipHostFree(p));
        p = nullptr;
        cap = 0;
        SY_HIP(hipHostMalloc(&p, need + need / 2, hipHostMallocDefault));
        cap = need + need / 2;
    }
};

void build_line_index(sylph_ctx* ctx, const uint64_t* d_kmers, const uint32_t* d_gid, uint64_t n, uint64_t n_genomes, uint64_t kmer_lo,
                      uint64_t kmer_hi, LineIndex& ix);

}  // namespace sylph

struct sylph_db {
    sylph_ctx* ctx;
    uint64_t n_genomes = 0, n_kmers = 0;     // n_kmers: postings resident on this shard
    uint32_t min_glen = 0;                   // smallest genome (k-mers): the contain.rs:627 test is skipped when nothing can fail it
    sylph::LineIndex kept, tracked;          // genome_kmers; pseudotax_tracked_nonused_kmers (winner table only, types.rs:166)
    sylph::DevBuf glen;
    // k-mer-range sharding (sylph_db_upload_shard): this shard holds k-mers in [bounds[rank], bounds[rank + 1])
    std::vector<uint64_t> bounds;
    uint32_t world = 1, rank = 0;
    // genome sharding (sylph_db_upload_genome_shard): this shard indexes ALL k-mers of the genomes [g_bounds[rank], g_bounds[rank + 1])
    // under their global ids; `bounds` is then {0, ~0, .., ~0} — a table's slice for any shard is the whole table (shard_plan.h)
    bool by_genome = false;
    std::vector<uint64_t> g_bounds;
    uint64_t shard_hit_cap = 1ull << 20;     // hits per rank in the all-gathered block; doubles identically on every rank
    uint64_t x_batches = 0, x_table_bytes = 0, x_hit_bytes = 0;   // exchange totals (sylph_db_exchange_stats): batches, bytes sent to OTHER ranks
    sylph::DevBuf rank_of, ani, lost;        // reassign pass: rank[g] in the passing list (or ~0), ANI per rank, kmers_lost[g]
    // per-query scratch (owned by the db so concurrent dbs on one ctx do not alias)
    sylph::DevBuf q_kmers, q_counts, q_refs, hits, hits_sorted, res, counter;   // res: device copy of the result block
    sylph::DevBuf x_send, x_recv, x_meta;    // shard exchange buffers (shard.hip)
    // row assembly of the hits (hits.hip): replicated row counters + tile offsets / tickets / row lists.  The kernels leave the
    // counters (and the probe's hit counter) zeroed; *_dirty = they must be cleared before the next use (first use, a fallback
    // batch, a failure half-way)
    sylph::DevBuf row_counters, row_meta;
    bool rc_dirty = true, cnt_dirty = true;
    uint32_t rc_rep = 0, rc_rows = 0;
    hipEvent_t ev_cnt = nullptr;             // "the probe's counter words have arrived in pinned host memory"
    sylph::ResultLayout lay;                 // layout of the last result
    uint64_t last_rows = 0;
    sylph::HostBlock h_block;                // pinned host results of the plain entry points (the pipeline brings its own blocks)
    explicit sylph_db(sylph_ctx* cx)
        : ctx(cx), kept(cx), tracked(cx), glen(cx), rank_of(cx), ani(cx), lost(cx), q_kmers(cx), q_counts(cx), q_refs(cx), hits(cx),
          hits_sorted(cx), res(cx), counter(cx), x_send(cx), x_recv(cx), x_meta(cx), row_counters(cx), row_meta(cx) {}
    ~sylph_db() { if (ev_cnt) (void)hipEventDestroy(ev_cnt); }
};

namespace sylph {
// Probes a batch of device-resident sample tables (refs: host array of n_samples entries with k, c, n filled in) against the
// kept index and leaves unsorted hits ((row << 32) | count, row = sample * n_genomes + genome) in db->hits.
// Returns the number of hits; *max_count = largest count among them.
uint32_t probe_batch(sylph_db* db, std::vector<SampleRef>& refs, double min_number_kmers, uint32_t* max_count);
// The same without waiting for the counts: launches the probe (hit array of *cap entries, grown to fit `want_cap`), queues the copy of
// the two counter words [hits, largest count] into the context's pinned page (+256) and records db->ev_cnt behind it.  Returns
// false when there is nothing to probe (no k-mers, empty index): the counts are 0 then and nothing was launched.
bool probe_batch_async(sylph_db* db, std::vector<SampleRef>& refs, double min_number_kmers, uint64_t want_cap, uint64_t* cap);
// hits.hip: hit list in db->hits -> cov_off / contain_count / per-row ascending coverage values in the device result block, five
// launches, sizes from device memory (d_cnt) or from the host (n_imm, max_imm).  false = not taken (nothing launched).
bool launch_row_assembly(sylph_db* db, const uint32_t* d_cnt, uint32_t n_imm, uint32_t max_imm, uint32_t cap, uint64_t n_rows, int want_narrow,
                         char* d_res, size_t covs_offset, size_t ccount_offset);
uint32_t row_assembly_max_value();
// Probe + assembly of a batch of device-resident tables (refs) into `dst` (nullptr: the database's block) without a host round trip
// between the probe and the assembly.  Returns the number of hits.
uint32_t probe_and_finish(sylph_db* db, std::vector<SampleRef>& refs, double min_number_kmers, uint64_t n_rows, uint32_t* cov_width,
                          HostBlock* dst);
// Sorts n_hits hits of db->hits (rows < n_rows) and assembles + copies out the result block (layout db->lay) into `dst`
// (nullptr: the database's own block).
void finish_hits(sylph_db* db, uint32_t n_hits, uint32_t max_count, uint64_t n_rows, uint32_t* cov_width, bool with_lost,
                 HostBlock* dst = nullptr);
// The bodies of sylph_db_contain_batch / sylph_db_contain_batch_sharded (they take the context lock themselves); the result
// block lands in `dst` (nullptr: the database's own).  Return the number of coverage values.
// `views` (optional): the three result pointers into the block the call filled, taken while the context lock is still held —
// a pipeline's profile thread working on the same database between the lock's release and the caller's own look at
// db->lay / db->h_block would otherwise hand the caller somebody else's layout.
struct ResultViews { const uint64_t* cov_off = nullptr; const uint32_t* contain_count = nullptr; const void* covs = nullptr; };
uint32_t contain_batch_impl(sylph_db* db, const sylph_sample_ref* samples, uint32_t n_samples, int mem, double min_number_kmers,
                            uint32_t* cov_width, HostBlock* dst, ResultViews* views = nullptr);
uint32_t contain_batch_sharded_impl(sylph_db* db, sylph_comm* comm, const sylph_sample_ref* samples, uint32_t n_local, int mem,
                                    double min_number_kmers, uint32_t* cov_width, HostBlock* dst, ResultViews* views = nullptr);
inline void fill_views(const HostBlock& b, ResultViews* v) {
    if (!v) return;
    const char* h = (const char*)b.p;
    v->cov_off = (const uint64_t*)h;
    v->contain_count = (const uint32_t*)(h + b.lay.ccount);
    v->covs = h + b.lay.covs;
}
}  // namespace sylph
