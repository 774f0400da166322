// shard_plan.h — the bookkeeping of the k-mer-range exchange (shard.hip), free of HIP: which slice of which table goes where, where a
// received slice lies, which rank owns a hit, where its group starts.  shard.hip runs these on the gathered size blocks between
// its collectives; tests/test_dist.py compiles the same header with g++ (tests/shard_plan_capi.cpp) and drives it under gloo with
// world sizes 2 and 3, so the arithmetic that has never met more than one physical GPU is at least exercised rank against rank.
#pragma once
#include <cstdint>
#include <algorithm>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#define SY_PLAN_HD __host__ __device__ __forceinline__
#else
#define SY_PLAN_HD inline
#endif

namespace sylph {
namespace shardplan {

constexpr uint32_t MAX_LOCAL = 64;           // samples a rank may contribute to one batch (fixes the size of the meta block)
constexpr uint32_t MAX_WORLD = 64;

// first entry of the ascending table k[0..n) that is >= key (what split_kernel computes per (sample, boundary))
SY_PLAN_HD uint64_t lower_bound_u64(const uint64_t* k, uint64_t n, uint64_t key) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (k[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// owner of global sample s = the rank r with prefix[r] <= s < prefix[r + 1]
SY_PLAN_HD uint32_t owner_of_sample(uint64_t s, const uint64_t* prefix, uint32_t world) {
    uint32_t lo = 0, hi = world;
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (prefix[mid] <= s) lo = mid; else hi = mid;
    }
    return lo;
}

// The all-gathered meta blocks: per rank [n_local | split[MAX_LOCAL][W + 1]] (u64), split[s][j] = first entry of the rank's sample s
// whose k-mer is >= bounds[j].
inline uint64_t meta_words(uint32_t W) { return 1 + (uint64_t)MAX_LOCAL * (W + 1); }
// whole = 1: a database sharded by GENOME (sylph_db_upload_genome_shard; north_star's wording) — every rank probes every sample in
// full, so the "slice" of a table for any destination is the whole table [split[0], split[W]) = [0, n): the all-to-all of the slices
// then IS the all-gather of the tables, and everything downstream (probe, owners, hits, assembly) is the k-mer-range path's.
struct Meta {
    const uint64_t* words;   // W blocks of meta_words(W)
    uint32_t W;
    int whole = 0;
    uint32_t n_loc(uint32_t r) const { return (uint32_t)words[(size_t)r * meta_words(W)]; }
    uint64_t split(uint32_t r, uint32_t s, uint32_t j) const { return words[(size_t)r * meta_words(W) + 1 + (size_t)s * (W + 1) + j]; }
    uint64_t slice_begin(uint32_t r, uint32_t s, uint32_t d) const { return split(r, s, whole ? 0 : d); }          // first entry of (rank r, sample s) that goes to shard d
    uint64_t slice_len(uint32_t r, uint32_t s, uint32_t d) const { return whole ? split(r, s, W) - split(r, s, 0) : split(r, s, d + 1) - split(r, s, d); }   // entries of (rank r, sample s) for shard d
};

// Genome shards: contiguous genome ranges [g_bounds[r], g_bounds[r + 1]) holding about 1 / W of the database's k-mers each
// (off: the genome-major offsets, G + 1 entries) — genomes are the reference's unit of parallelism (contain.rs:284).
inline void genome_bounds(const uint64_t* off, uint64_t G, uint32_t W, uint64_t* g_bounds) {
    const uint64_t total = off[G] - off[0];
    g_bounds[0] = 0;
    for (uint32_t r = 1; r < W; r++) {
        const uint64_t target = off[0] + (uint64_t)((unsigned __int128)total * r / W);
        uint64_t lo = 0, hi = G;                       // first genome whose start offset is >= target
        while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (off[mid] < target) lo = mid + 1; else hi = mid; }
        g_bounds[r] = lo < g_bounds[r - 1] ? g_bounds[r - 1] : lo;
    }
    g_bounds[W] = G;
}

// Step 2: the all-to-all of the table slices.  Block (src -> dst) = [k-mers of slice (s, dst), s = 0.. | counts of the same slices | pad to 8].
struct SlicePlan {
    std::vector<uint64_t> prefix;              // prefix[r] = global index of rank r's first sample; prefix[W] = samples of the step
    std::vector<uint64_t> send_off, recv_off;  // byte offsets of the W blocks in this rank's send / receive buffers (W + 1 entries)
    uint64_t S_total = 0;
    std::string error;                         // non-empty: the step cannot run (the same verdict on every rank: it only reads the gathered meta)
};
inline uint64_t slice_block_bytes(const Meta& m, uint32_t src, uint32_t dst) {
    uint64_t e = 0;
    for (uint32_t s = 0; s < m.n_loc(src); s++) e += m.slice_len(src, s, dst);
    return (e * 12 + 7) & ~7ull;
}
inline SlicePlan plan_slices(const Meta& m, uint32_t me, uint64_t n_genomes) {
    SlicePlan p;
    const uint32_t W = m.W;
    p.prefix.assign(W + 1, 0);
    for (uint32_t r = 0; r < W; r++) {
        if (m.n_loc(r) > MAX_LOCAL) { p.error = "rank " + std::to_string(r) + " announced " + std::to_string(m.n_loc(r)) + " samples"; return p; }
        p.prefix[r + 1] = p.prefix[r] + m.n_loc(r);
    }
    p.S_total = p.prefix[W];
    if (p.S_total * (n_genomes ? n_genomes : 1) >= (1ull << 32) - 1) { p.error = "samples x genomes of one step must stay below 2^32"; return p; }
    p.send_off.assign(W + 1, 0);
    p.recv_off.assign(W + 1, 0);
    for (uint32_t r = 0; r < W; r++) {
        p.send_off[r + 1] = p.send_off[r] + slice_block_bytes(m, me, r);
        p.recv_off[r + 1] = p.recv_off[r] + slice_block_bytes(m, r, me);
    }
    return p;
}
// where slice (sample s of rank `me`, shard d) goes inside this rank's send block for d, and where slice (sample s of rank r)
// lies inside the block received from r: byte offsets relative to the block, k-mers and counts apart
struct SliceAt { uint64_t k_off, c_off, len; };
inline SliceAt slice_in_block(const Meta& m, uint32_t src, uint32_t s, uint32_t dst) {
    uint64_t e = 0, before = 0;
    for (uint32_t t = 0; t < m.n_loc(src); t++) {
        const uint64_t len = m.slice_len(src, t, dst);
        if (t < s) before += len;
        e += len;
    }
    return SliceAt{before * 8, e * 8 + before * 4, m.slice_len(src, s, dst)};
}

// Steps 4-5: the all-gathered size blocks, per rank SZ = W + 3 words: [hits for rank 0..W-1 | largest count | error word | hits of the rank].
struct HitPlan {
    std::vector<uint64_t> send_off, recv_off;  // byte offsets (8 B per hit) of the groups this rank sends / receives
    std::vector<uint32_t> start;               // first entry of the group for rank r in the send buffer (hits, not bytes)
    uint32_t max_mine = 0;                     // largest count among the hits this rank will receive
    uint64_t n_mine = 0;
    uint32_t failed_rank = 0xFFFFFFFFu, failed_class = 0;   // a rank raised its error word: every rank stops with it
    std::string error;                         // bookkeeping that does not add up / a destination beyond 2^32-1 hits
};
inline uint32_t size_words(uint32_t W) { return W + 3; }
inline HitPlan plan_hits(const uint32_t* sizes, uint32_t W, uint32_t me) {
    HitPlan p;
    const uint32_t SZ = size_words(W);
    auto n_from_to = [&](uint32_t src, uint32_t dst) { return (uint64_t)sizes[(size_t)src * SZ + dst]; };
    for (uint32_t r = 0; r < W; r++) {
        const uint32_t err_r = sizes[(size_t)r * SZ + W + 1];
        if (err_r) { p.failed_rank = r; p.failed_class = err_r; return p; }
        uint64_t sent = 0;
        for (uint32_t d = 0; d < W; d++) sent += n_from_to(r, d);
        if (sent != sizes[(size_t)r * SZ + W + 2]) { p.error = "internal: owner counts of rank " + std::to_string(r) + " do not add up"; return p; }
    }
    // (checked for EVERY destination from the gathered matrix, so that all ranks fail together instead of one leaving the others
    //  waiting in the next collective)
    for (uint32_t dst = 0; dst < W; dst++) {
        uint64_t to_dst = 0;
        for (uint32_t src = 0; src < W; src++) to_dst += n_from_to(src, dst);
        if (to_dst >= (1ull << 32)) { p.error = "more than 2^32-1 hits for the samples of rank " + std::to_string(dst) + " in one step: use smaller batches"; return p; }
    }
    p.send_off.assign(W + 1, 0);
    p.recv_off.assign(W + 1, 0);
    p.start.assign(W, 0);
    for (uint32_t r = 0; r < W; r++) {
        p.start[r] = (uint32_t)(p.send_off[r] / 8);
        p.send_off[r + 1] = p.send_off[r] + n_from_to(me, r) * 8;
        p.recv_off[r + 1] = p.recv_off[r] + n_from_to(r, me) * 8;
        if (n_from_to(r, me)) p.max_mine = p.max_mine > sizes[(size_t)r * SZ + W] ? p.max_mine : sizes[(size_t)r * SZ + W];
    }
    p.n_mine = p.recv_off[W] / 8;
    return p;
}
// Step 5 as ONE ALL-GATHER instead of an all-to-all (round 6; north_star's wording: "per-shard containment counts reduced by a single
// RCCL all-gather"): every rank contributes its WHOLE grouped hit buffer, padded to the longest one (pad_bytes), every rank receives
// all W of them and keeps the group meant for it from each: piece r lies at src_off[r] of the gathered buffer (block r + the groups for
// the ranks in front of me), is len[r] bytes long and goes to recv_off[r] of the hit list — the layout the all-to-all would have
// produced.  W x more bytes on the links than the all-to-all (every rank receives what was meant for the others too); one collective
// call, no point-to-point schedule.  A/B-able: sylph_ctx_set_option(ctx, "shard_reduce", "allgather").
struct GatherPlan { uint64_t pad_bytes = 0; std::vector<uint64_t> src_off, len; };
inline GatherPlan plan_hits_gather(const uint32_t* sizes, uint32_t W, uint32_t me) {
    GatherPlan g;
    const uint32_t SZ = size_words(W);
    for (uint32_t r = 0; r < W; r++) g.pad_bytes = std::max<uint64_t>(g.pad_bytes, (uint64_t)sizes[(size_t)r * SZ + W + 2] * 8);
    g.src_off.assign(W, 0);
    g.len.assign(W, 0);
    for (uint32_t r = 0; r < W; r++) {
        uint64_t before = 0;
        for (uint32_t d = 0; d < me; d++) before += sizes[(size_t)r * SZ + d];
        g.src_off[r] = (uint64_t)r * g.pad_bytes + before * 8;
        g.len[r] = (uint64_t)sizes[(size_t)r * SZ + me] * 8;
    }
    return g;
}
// a hit (row << 32 | count) with row = global sample * n_genomes + genome, re-based to its owner's samples
SY_PLAN_HD uint64_t rebase_hit(uint64_t hit, uint64_t owner_first_sample, uint64_t n_genomes) { return hit - ((owner_first_sample * n_genomes) << 32); }

}  // namespace shardplan
}  // namespace sylph
