// C wrapper around sylph_amd/csrc/shard_plan.h for tests/test_dist.py (g++, no HIP): the very header shard.hip includes.
#include <cstring>

#include "../sylph_amd/csrc/shard_plan.h"

using namespace sylph::shardplan;

static void put_err(const std::string& e, char* err, size_t errn) {
    if (err && errn) { strncpy(err, e.c_str(), errn - 1); err[errn - 1] = 0; }
}

extern "C" {
uint32_t sp_max_local() { return MAX_LOCAL; }
uint64_t sp_meta_words(uint32_t W) { return meta_words(W); }
uint32_t sp_size_words(uint32_t W) { return size_words(W); }
uint64_t sp_lower_bound(const uint64_t* k, uint64_t n, uint64_t key) { return lower_bound_u64(k, n, key); }
uint32_t sp_owner(uint64_t s, const uint64_t* prefix, uint32_t W) { return owner_of_sample(s, prefix, W); }
uint64_t sp_rebase(uint64_t hit, uint64_t first, uint64_t G) { return rebase_hit(hit, first, G); }
static int g_whole = 0;                       // 1: genome shards (every slice is the whole table)
void sp_set_whole(int whole) { g_whole = whole; }
uint64_t sp_slice_begin(const uint64_t* meta, uint32_t W, uint32_t r, uint32_t s, uint32_t d) { return Meta{meta, W, g_whole}.slice_begin(r, s, d); }
void sp_genome_bounds(const uint64_t* off, uint64_t G, uint32_t W, uint64_t* out) { genome_bounds(off, G, W, out); }
int sp_plan_slices(const uint64_t* meta, uint32_t W, uint32_t me, uint64_t G, uint64_t* prefix, uint64_t* send_off, uint64_t* recv_off,
                   uint64_t* S_total, char* err, size_t errn) {
    const SlicePlan p = plan_slices(Meta{meta, W, g_whole}, me, G);
    if (!p.error.empty()) { put_err(p.error, err, errn); return 1; }
    memcpy(prefix, p.prefix.data(), (W + 1) * 8);
    memcpy(send_off, p.send_off.data(), (W + 1) * 8);
    memcpy(recv_off, p.recv_off.data(), (W + 1) * 8);
    *S_total = p.S_total;
    return 0;
}
void sp_slice_in_block(const uint64_t* meta, uint32_t W, uint32_t src, uint32_t s, uint32_t dst, uint64_t out[3]) {
    const SliceAt a = slice_in_block(Meta{meta, W, g_whole}, src, s, dst);
    out[0] = a.k_off; out[1] = a.c_off; out[2] = a.len;
}
int sp_plan_hits(const uint32_t* sizes, uint32_t W, uint32_t me, uint64_t* send_off, uint64_t* recv_off, uint32_t* start, uint32_t* max_mine,
                 uint64_t* n_mine, uint32_t* failed_rank, uint32_t* failed_class, char* err, size_t errn) {
    const HitPlan p = plan_hits(sizes, W, me);
    *failed_rank = p.failed_rank;
    *failed_class = p.failed_class;
    if (p.failed_rank != 0xFFFFFFFFu) return 2;
    if (!p.error.empty()) { put_err(p.error, err, errn); return 1; }
    memcpy(send_off, p.send_off.data(), (W + 1) * 8);
    memcpy(recv_off, p.recv_off.data(), (W + 1) * 8);
    memcpy(start, p.start.data(), W * 4);
    *max_mine = p.max_mine;
    *n_mine = p.n_mine;
    return 0;
}
void sp_plan_hits_gather(const uint32_t* sizes, uint32_t W, uint32_t me, uint64_t* pad_bytes, uint64_t* src_off, uint64_t* len) {
    const GatherPlan g = plan_hits_gather(sizes, W, me);
    *pad_bytes = g.pad_bytes;
    memcpy(src_off, g.src_off.data(), W * 8);
    memcpy(len, g.len.data(), W * 8);
}
}
