// shard.hip — one database over several GPUs (north_star: "genome DB ... sharded across the 8 GPUs of one node,
// per-shard containment counts reduced by a single RCCL all-gather over xGMI").  One process per GPU.
//
// The reference has no distributed path (contain.rs:239-291 is a rayon loop over sample chunks x genomes inside one
// process), so nothing here translates one.  The database is sharded by k-mer RANGE (contain.hip, sylph_db_upload_shard):
// sample tables are sorted by k-mer, so what a rank has to probe of any sample is a contiguous slice holding 1/world of it —
// per-rank probe work per sample is constant in `world` (with genome-sharding every rank would probe every sample in full).
// Per batch of samples (every rank contributes the samples it sketched):
//   1. all-gather of the slice boundaries                                  (a few KB, fixed-size block per rank)
//   2. all-to-all of the table slices                                      (each rank receives 1/world of every table)
//   3. one probe launch over all received slices against the resident shard
//   4. the hits are grouped by the rank that owns their sample; all-gather of the group sizes (W + 1 words per rank)
//   5. all-to-all of the hit groups: every rank receives exactly the hits of ITS samples, from every shard
//   6. every rank sorts its hits and assembles counts + coverage vectors (contain.hip, as for an unsharded batch): partial
//      counts of a genome from different shards simply add up, because a k-mer lives on exactly one shard.
// xGMI is point-to-point (7 links per GPU): an all-to-all puts 1/world of the payload on each link at the same time, whereas
// an all-gather of whole tables / whole hit lists would deliver world x the bytes anybody needs (at GTDB scale a step of
// 64 samples produces ~0.9 GB of hits; each rank needs its 1/8).  The two all-gathers carry a few KB: latency-bound.
// north_star words the reduction as "a single RCCL all-gather" of containment counts; that fits per-genome count vectors of a
// genome-sharded database, whose probe work grows with the number of GPUs — the reason for sharding by k-mer range instead.
// All buffers stay on the device; the host only reads the sizes.
//
// The collectives come from a sylph_comm: RCCL (resolved with dlopen at run time, so that the library neither links against a
// second copy of librccl next to the one PyTorch may already have mapped, nor needs RCCL at all on a single GPU), or callbacks
// supplied by the caller (tests run the same code over gloo).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <memory>

#include "contain_index.h"
#include "shard_plan.h"

namespace sylph {
namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy already mapped into the process first (PyTorch ships its own librccl.so), then the ROCm installation
        const char* names[] = {"librccl.so.1", "librccl.so"};
        void* h = nullptr;
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        api.lib = h;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
        api.Send = (decltype(api.Send))dlsym(h, "ncclSend");
        api.Recv = (decltype(api.Recv))dlsym(h, "ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    });
    return api;
}

void rccl_require() {
    RcclApi& a = rccl();
    SY_REQUIRE(a.lib && a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.Send && a.Recv && a.GroupStart && a.GroupEnd,
               "librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "missing symbols");
}

#define SY_NCCL(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t _r = (expr);                                                                                  \
        if (_r != ncclSuccess) {                                                                                   \
            char _b[384];                                                                                          \
            snprintf(_b, sizeof(_b), "RCCL error %d (%s) in %s", (int)_r,                                          \
                     rccl().GetErrorString ? rccl().GetErrorString(_r) : "?", #expr);                              \
            throw ::sylph::ArgError{_b};                                                                           \
        }                                                                                                          \
    } while (0)

}  // namespace
}  // namespace sylph

using namespace sylph;

struct sylph_comm {
    uint32_t rank = 0, world = 1;
    sylph_comm_ops ops{nullptr, nullptr};
    void* user = nullptr;
    ncclComm_t nccl = nullptr;       // RCCL flavour
    int device = -1;

    void all_gather(const void* send, void* recv, uint64_t bytes, hipStream_t s) {
        if (nccl) { SY_NCCL(rccl().AllGather(send, recv, bytes, ncclUint8, nccl, s)); return; }
        SY_REQUIRE(ops.all_gather(user, send, recv, bytes, (void*)s) == 0, "all_gather callback failed");
    }
    void all_to_all(const void* send, const uint64_t* send_off, void* recv, const uint64_t* recv_off, hipStream_t s) {
        if (!nccl) { SY_REQUIRE(ops.all_to_all(user, send, send_off, recv, recv_off, (void*)s) == 0, "all_to_all callback failed"); return; }
        // own block: a device copy; the others: one grouped send/recv pair per peer (xGMI is point-to-point: all links at once)
        const uint64_t mine = send_off[rank + 1] - send_off[rank];
        SY_REQUIRE(mine == recv_off[rank + 1] - recv_off[rank], "internal: self block size mismatch");
        if (mine) SY_HIP(hipMemcpyAsync((char*)recv + recv_off[rank], (const char*)send + send_off[rank], mine, hipMemcpyDeviceToDevice, s));
        SY_NCCL(rccl().GroupStart());
        for (uint32_t r = 0; r < world; r++) {
            if (r == rank) continue;
            const uint64_t ns = send_off[r + 1] - send_off[r], nr = recv_off[r + 1] - recv_off[r];
            if (ns) SY_NCCL(rccl().Send((const char*)send + send_off[r], ns, ncclUint8, (int)r, nccl, s));
            if (nr) SY_NCCL(rccl().Recv((char*)recv + recv_off[r], nr, ncclUint8, (int)r, nccl, s));
        }
        SY_NCCL(rccl().GroupEnd());
    }
};

namespace sylph {
namespace {

using shardplan::MAX_LOCAL;                  // samples a rank may contribute to one batch (shard_plan.h: the exchange's bookkeeping,
using shardplan::MAX_WORLD;                  // host-compilable — tests/test_dist.py drives it under gloo)

// split[s * (W + 1) + j] = first entry of sample s whose k-mer is >= bounds[j]   (j = 0..W; the tables are ascending)
__global__ __launch_bounds__(256) void split_kernel(const SampleRef* __restrict__ refs, uint32_t n_samples, const uint64_t* __restrict__ bounds,
                                                    uint32_t W, uint64_t* __restrict__ split) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_samples * (W + 1)) return;
    const uint32_t s = t / (W + 1), j = t % (W + 1);
    split[t] = shardplan::lower_bound_u64(refs[s].k, refs[s].n, bounds[j]);
}

// segmented copy in 4-byte words: workgroup (x = segment, y strides)
struct Seg { const uint32_t* src; uint32_t* dst; uint64_t words; };
__global__ __launch_bounds__(256) void copy_segments_kernel(const Seg* __restrict__ segs) {
    const Seg sg = segs[blockIdx.x];
    for (uint64_t i = (uint64_t)blockIdx.y * blockDim.x + threadIdx.x; i < sg.words; i += (uint64_t)gridDim.y * blockDim.x) sg.dst[i] = sg.src[i];
}

// owner of a hit = the rank whose samples include row / n_genomes (prefix[r] <= sample < prefix[r + 1])
__device__ __forceinline__ uint32_t owner_of(uint64_t hit, uint64_t n_genomes, const uint64_t* __restrict__ prefix, uint32_t world) {
    return shardplan::owner_of_sample((hit >> 32) / n_genomes, prefix, world);
}
// Wavefront-aggregated "take a slot in bin r": lanes of a wave that want the same bin are served by ONE atomic (hits arrive in
// runs of one sample, i.e. one owner: a plain per-lane atomic would serialise 64 lanes on one LDS word).  Returns the lane's
// slot; every active lane must call it.
__device__ __forceinline__ uint32_t wave_take(uint32_t* bins, uint32_t r, bool valid) {
    uint32_t slot = 0;
    unsigned long long todo = __ballot(valid);
    const uint32_t lane = threadIdx.x & 63;
    while (todo) {
        const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1;
        const uint32_t r0 = (uint32_t)__shfl((int)r, (int)leader);
        const unsigned long long same = __ballot(valid && r == r0) & todo;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&bins[r0], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, (int)leader);
        if (valid && r == r0) slot = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}
// counts[r] = hits owned by rank r (LDS histogram per workgroup, one global atomic per bin and workgroup)
__global__ __launch_bounds__(256) void owner_count_kernel(const uint64_t* __restrict__ hits, uint32_t n, uint64_t n_genomes,
                                                          const uint64_t* __restrict__ prefix, uint32_t world, uint32_t* __restrict__ counts) {
    __shared__ uint32_t bins[MAX_WORLD];
    if (threadIdx.x < MAX_WORLD) bins[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const bool valid = i < n;
        (void)wave_take(bins, valid ? owner_of(hits[i], n_genomes, prefix, world) : 0u, valid);
    }
    __syncthreads();
    if (threadIdx.x < world && bins[threadIdx.x]) atomicAdd(&counts[threadIdx.x], bins[threadIdx.x]);
}
// out[start[r] + k] = k-th hit of owner r (any order inside a group: the owner sorts), rows re-based to the owner's samples.
// A workgroup bins SCATTER_ITEMS x 256 hits in LDS before it takes its ranges with one global atomic per bin: a single word
// sustains only ~90 atomics per microsecond on this chip, and with one atomic per 256 hits the cursors were the whole kernel.
constexpr int SCATTER_ITEMS = 16;
__global__ __launch_bounds__(256) void owner_scatter_kernel(const uint64_t* __restrict__ hits, uint32_t n, uint64_t n_genomes,
                                                            const uint64_t* __restrict__ prefix, uint32_t world, const uint32_t* __restrict__ start,
                                                            uint32_t* __restrict__ cursor, uint64_t* __restrict__ out) {
    __shared__ uint32_t bins[MAX_WORLD], base[MAX_WORLD];
    const uint32_t tile = 256 * SCATTER_ITEMS;
    for (uint32_t i0 = blockIdx.x * tile; i0 < n; i0 += gridDim.x * tile) {
        if (threadIdx.x < MAX_WORLD) bins[threadIdx.x] = 0;
        __syncthreads();
        uint64_t h[SCATTER_ITEMS];
        uint32_t r[SCATTER_ITEMS], slot[SCATTER_ITEMS];
#pragma unroll
        for (int j = 0; j < SCATTER_ITEMS; j++) {
            const uint32_t i = i0 + j * 256 + threadIdx.x;
            const bool valid = i < n;
            h[j] = valid ? hits[i] : 0;
            r[j] = valid ? owner_of(h[j], n_genomes, prefix, world) : 0xFFFFFFFFu;
            slot[j] = wave_take(bins, valid ? r[j] : 0u, valid);
        }
        __syncthreads();
        if (threadIdx.x < world && bins[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], bins[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SCATTER_ITEMS; j++)
            if (r[j] != 0xFFFFFFFFu) out[start[r[j]] + base[r[j]] + slot[j]] = shardplan::rebase_hit(h[j], prefix[r[j]], n_genomes);
        __syncthreads();
    }
}

uint32_t grid_for64(uint64_t n, uint32_t tpb = 256) { return (uint32_t)((n + tpb - 1) / tpb); }

}  // namespace
}  // namespace sylph

extern "C" {

int sylph_comm_rccl_unique_id(uint8_t id[128]) {
    return guarded([&] {
        SY_REQUIRE(id, "null argument");
        rccl_require();
        ncclUniqueId u;
        static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
        SY_NCCL(rccl().GetUniqueId(&u));
        memcpy(id, &u, 128);
    });
}

int sylph_comm_create_rccl(sylph_ctx* ctx, uint32_t rank, uint32_t world, const uint8_t id[128], sylph_comm** out) {
    return guarded([&] {
        SY_REQUIRE(ctx && id && out && world >= 1 && rank < world, "bad argument");
        rccl_require();
        DeviceGuard dg(ctx->device);
        std::unique_ptr<sylph_comm> c(new sylph_comm());
        c->rank = rank; c->world = world; c->device = ctx->device;
        ncclUniqueId u;
        memcpy(&u, id, 128);
        SY_NCCL(rccl().CommInitRank(&c->nccl, (int)world, u, (int)rank));
        *out = c.release();
    });
}

int sylph_comm_create(uint32_t rank, uint32_t world, const sylph_comm_ops* ops, void* user, sylph_comm** out) {
    return guarded([&] {
        SY_REQUIRE(ops && ops->all_gather && ops->all_to_all && out && world >= 1 && rank < world, "bad argument");
        sylph_comm* c = new sylph_comm();
        c->rank = rank; c->world = world; c->ops = *ops; c->user = user;
        *out = c;
    });
}

void sylph_comm_destroy(sylph_comm* comm) {
    if (!comm) return;
    if (comm->nccl) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (comm->device >= 0) (void)hipSetDevice(comm->device);
        (void)rccl().CommDestroy(comm->nccl);
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    delete comm;
}

}  // extern "C"

uint32_t sylph::contain_batch_sharded_impl(sylph_db* db, sylph_comm* comm, const sylph_sample_ref* samples, uint32_t n_local, int mem,
                                           double min_number_kmers, uint32_t* cov_width, HostBlock* dst, ResultViews* views) {
    {
        SY_REQUIRE(db && comm && cov_width, "null argument");
        SY_REQUIRE(n_local == 0 || samples, "null samples");
        SY_REQUIRE(mem == SYLPH_MEM_HOST || mem == SYLPH_MEM_DEVICE, "bad mem kind %d", mem);
        SY_REQUIRE(n_local <= MAX_LOCAL, "at most %u samples per rank and batch", MAX_LOCAL);
        SY_REQUIRE(db->world == comm->world && db->rank == comm->rank, "database shard %u/%u does not match communicator rank %u/%u", db->rank,
                   db->world, comm->rank, comm->world);
        SY_REQUIRE(db->bounds.size() == (size_t)db->world + 1, "database was not uploaded with sylph_db_upload_shard / sylph_db_upload_genome_shard");
        SY_REQUIRE(comm->world <= MAX_WORLD, "at most %u ranks", MAX_WORLD);
        sylph_ctx* ctx = db->ctx;
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard dg(ctx->device);
        const uint32_t W = comm->world, me = comm->rank;
        const uint64_t G = db->n_genomes;
        hipStream_t st = ctx->stream;

        HostPhase ph_all(ctx, "sharded contain: total");
        std::unique_ptr<HostPhase> ph(new HostPhase(ctx, "shard 0: stage tables"));
        // ---- local tables on the device
        std::vector<SampleRef> mine(n_local);
        uint64_t total = 0;
        for (uint32_t s = 0; s < n_local; s++) {
            SY_REQUIRE(samples[s].n == 0 || (samples[s].kmers && samples[s].counts), "null sample %u", s);
            SY_REQUIRE(samples[s].n < (1ull << 32), "sample table larger than 2^32-1 entries");
            total += samples[s].n;
        }
        if (mem == SYLPH_MEM_HOST && total) {
            db->q_kmers.reserve(total * 8);
            db->q_counts.reserve(total * 4);
            uint64_t o = 0;
            for (uint32_t s = 0; s < n_local; s++) {
                const uint64_t n = samples[s].n;
                if (n) {
                    ctx->h2d(db->q_kmers.as<uint64_t>() + o, samples[s].kmers, n * 8);
                    ctx->h2d(db->q_counts.as<uint32_t>() + o, samples[s].counts, n * 4);
                }
                mine[s].k = db->q_kmers.as<uint64_t>() + o; mine[s].c = db->q_counts.as<uint32_t>() + o; mine[s].n = n;
                o += n;
            }
        } else {
            for (uint32_t s = 0; s < n_local; s++) { mine[s].k = samples[s].kmers; mine[s].c = samples[s].counts; mine[s].n = samples[s].n; }
        }
        for (auto& r : mine) { r.chunk0 = 0; r.pad = 0; }

        ph.reset(); ph.reset(new HostPhase(ctx, "shard 1: split + all-gather meta"));
        // ---- 1. slice boundaries of every local table, all-gathered: block = [n_local | split[MAX_LOCAL][W + 1]] u64
        const uint64_t meta_words = shardplan::meta_words(W);
        // x_meta: [my block | gathered blocks (W) | bounds (W + 1) | refs | segs]
        const size_t off_gather = meta_words * 8, off_bounds = off_gather + (size_t)W * meta_words * 8, off_refs = off_bounds + (size_t)(W + 1) * 8;
        const size_t off_segs = off_refs + (size_t)MAX_LOCAL * sizeof(SampleRef);
        db->x_meta.reserve(off_segs + (size_t)MAX_LOCAL * W * 2 * sizeof(Seg) + 64);
        char* xm = db->x_meta.as<char>();
        uint64_t* d_myblock = reinterpret_cast<uint64_t*>(xm);
        uint64_t* d_gather = reinterpret_cast<uint64_t*>(xm + off_gather);
        uint64_t* d_bounds = reinterpret_cast<uint64_t*>(xm + off_bounds);
        SampleRef* d_refs = reinterpret_cast<SampleRef*>(xm + off_refs);
        Seg* d_segs = reinterpret_cast<Seg*>(xm + off_segs);
        SY_HIP(hipMemsetAsync(d_myblock, 0, meta_words * 8, st));
        const uint64_t nl64 = n_local;
        ctx->h2d(d_myblock, &nl64, 8);
        ctx->h2d(d_bounds, db->bounds.data(), (size_t)(W + 1) * 8);
        if (n_local) {
            ctx->h2d(d_refs, mine.data(), (size_t)n_local * sizeof(SampleRef));
            hipLaunchKernelGGL(split_kernel, dim3(grid_for64((uint64_t)n_local * (W + 1))), dim3(256), 0, st, d_refs, n_local, d_bounds, W,
                               d_myblock + 1);
            SY_HIP(hipGetLastError());
        }
        comm->all_gather(d_myblock, d_gather, meta_words * 8, st);
        std::vector<uint64_t> meta((size_t)W * meta_words);
        ctx->d2h(meta.data(), d_gather, meta.size() * 8);
        // every offset below comes from shard_plan.h (the same arithmetic on every rank, from the same gathered block)
        const shardplan::Meta pm{meta.data(), W, db->by_genome ? 1 : 0};
        auto n_loc = [&](uint32_t r) { return pm.n_loc(r); };
        const shardplan::SlicePlan sp = shardplan::plan_slices(pm, me, G);
        SY_REQUIRE(sp.error.empty(), "%s", sp.error.c_str());
        const std::vector<uint64_t>& prefix = sp.prefix;
        const uint64_t S_total = sp.S_total;

        ph.reset(); ph.reset(new HostPhase(ctx, "shard 2: pack + all-to-all slices"));
        // ---- 2. all-to-all of the slices.  Block for rank d: [k-mers of slice (s, d), s = 0.. | counts of slice (s, d) | pad to 8]
        const std::vector<uint64_t>&send_off = sp.send_off, &recv_off = sp.recv_off;
        db->x_send.reserve(send_off[W] + 64);
        db->x_recv.reserve(recv_off[W] + 64);
        if (n_local) {
            std::vector<Seg> segs;
            uint64_t max_words = 1;
            for (uint32_t d = 0; d < W; d++) {
                char* blk = db->x_send.as<char>() + send_off[d];
                for (uint32_t s = 0; s < n_local; s++) {
                    const shardplan::SliceAt at = shardplan::slice_in_block(pm, me, s, d);
                    if (!at.len) continue;
                    const uint64_t a = pm.slice_begin(me, s, d);
                    segs.push_back(Seg{reinterpret_cast<const uint32_t*>(mine[s].k + a), reinterpret_cast<uint32_t*>(blk + at.k_off), at.len * 2});
                    segs.push_back(Seg{mine[s].c + a, reinterpret_cast<uint32_t*>(blk + at.c_off), at.len});
                    max_words = std::max(max_words, at.len * 2);
                }
            }
            if (!segs.empty()) {
                ctx->h2d(d_segs, segs.data(), segs.size() * sizeof(Seg));
                const uint32_t gy = (uint32_t)std::min<uint64_t>(1024, (max_words + 256 * 4 - 1) / (256 * 4));
                hipLaunchKernelGGL(copy_segments_kernel, dim3((uint32_t)segs.size(), std::max(1u, gy)), dim3(256), 0, st, d_segs);
                SY_HIP(hipGetLastError());
            }
        }
        {
            ScopedKernelTimer t(ctx, "exchange");
            comm->all_to_all(db->x_send.p, send_off.data(), db->x_recv.p, recv_off.data(), st);
        }
        db->x_batches++;
        db->x_table_bytes += send_off[W] - (send_off[me + 1] - send_off[me]);

        ph.reset(); ph.reset(new HostPhase(ctx, "shard 3: probe"));
        // ---- 3. probe every received slice against the resident shard; row = global sample index * G + genome
        std::vector<SampleRef> refs(S_total);
        for (uint32_t r = 0; r < W; r++) {
            const char* blk = db->x_recv.as<char>() + recv_off[r];
            for (uint32_t s = 0; s < n_loc(r); s++) {
                const shardplan::SliceAt at = shardplan::slice_in_block(pm, r, s, me);
                SampleRef& f = refs[prefix[r] + s];
                f.k = reinterpret_cast<const uint64_t*>(blk + at.k_off); f.c = reinterpret_cast<const uint32_t*>(blk + at.c_off); f.n = at.len;
            }
        }
        // A failure on ONE rank between two collectives (a batch beyond the probe's limits, a hit buffer that cannot grow, a HIP
        // error) must not leave the others waiting in the next collective: the rank goes on with an empty hit list and raises
        // an error word in the size block every rank is about to receive — all ranks then fail this call together.
        uint32_t max_count = 0, n_hits = 0, local_err = 0;
        std::string local_msg;
        if (S_total) {
            try {
                if (ctx->fail_next_shard_probe) { ctx->fail_next_shard_probe = 0; throw ArgError{"injected failure of the sharded probe (fail_next_shard_probe)"}; }
                n_hits = probe_batch(db, refs, min_number_kmers, &max_count);
            }
            catch (const ArgError& e) { local_err = 1; local_msg = e.msg; }
            catch (const HipError& e) { local_err = 2; local_msg = std::string("HIP error in the probe: ") + hipGetErrorString(e.e); (void)hipGetLastError(); }
            catch (const std::bad_alloc&) { local_err = 3; local_msg = "host allocation failed in the probe"; }
            if (local_err) { n_hits = 0; max_count = 0; }
        }

        ph.reset(); ph.reset(new HostPhase(ctx, "shard 4: owner counts + all-gather sizes"));
        // ---- 4. group the hits by owner rank; all-gather the group sizes: block (SZ = W + 3 words) = [count for rank 0..W-1 |
        // largest count value | error word | number of hits of the rank]
        // x_meta (reused): [prefix (W + 1) u64 | my sizes SZ u32 | cursors (W) u32 | starts (W) u32 | gathered sizes W x SZ u32]
        const uint32_t SZ = shardplan::size_words(W);
        uint64_t* d_prefix = reinterpret_cast<uint64_t*>(xm);
        uint32_t* d_sizes = reinterpret_cast<uint32_t*>(xm + (size_t)(W + 1) * 8);
        uint32_t* d_cursor = d_sizes + SZ;
        uint32_t* d_start = d_cursor + W;
        uint32_t* d_allsizes = d_start + W + ((W & 1) ? 0 : 1);
        ctx->h2d(d_prefix, prefix.data(), (size_t)(W + 1) * 8);
        SY_HIP(hipMemsetAsync(d_sizes, 0, (size_t)(3 * W + 4) * 4, st));
        if (n_hits) {
            hipLaunchKernelGGL(owner_count_kernel, dim3((uint32_t)std::min<uint64_t>(1024, grid_for64(n_hits))), dim3(256), 0, st, db->hits.as<uint64_t>(),
                               n_hits, G, d_prefix, W, d_sizes);
            SY_HIP(hipGetLastError());
        }
        const uint32_t trailer[3] = {max_count, local_err, n_hits};
        ctx->h2d(d_sizes + W, trailer, 12);
        comm->all_gather(d_sizes, d_allsizes, (uint64_t)SZ * 4, st);
        std::vector<uint32_t> sizes((size_t)W * SZ);
        ctx->d2h(sizes.data(), d_allsizes, sizes.size() * 4);
        // every rank looks at every rank's error word and bookkeeping (shard_plan.h): the same verdict everywhere
        const shardplan::HitPlan hp = shardplan::plan_hits(sizes.data(), W, me);
        if (hp.failed_rank != 0xFFFFFFFFu)
            throw ArgError{hp.failed_rank == me ? "sharded containment failed on this rank: " + local_msg
                                                : "sharded containment failed on rank " + std::to_string(hp.failed_rank) + " (error class " +
                                                      std::to_string(hp.failed_class) + "): this rank stops with it"};
        SY_REQUIRE(hp.error.empty(), "%s", hp.error.c_str());
        ph.reset(); ph.reset(new HostPhase(ctx, "shard 5: scatter + all-to-all hits"));
        // ---- 5. all-to-all of the hit groups
        const std::vector<uint64_t>&hs_off = hp.send_off, &hr_off = hp.recv_off;
        const std::vector<uint32_t>& start = hp.start;
        const uint32_t max_mine = hp.max_mine;
        db->x_send.reserve(hs_off[W] + 64);
        if (n_hits) {
            ctx->h2d(d_start, start.data(), (size_t)W * 4);
            hipLaunchKernelGGL(owner_scatter_kernel, dim3((uint32_t)std::min<uint64_t>(1024, grid_for64(n_hits, 256 * SCATTER_ITEMS))), dim3(256), 0, st,
                               db->hits.as<uint64_t>(), n_hits, G, d_prefix, W, d_start, d_cursor, db->x_send.as<uint64_t>());
            SY_HIP(hipGetLastError());
        }
        const uint32_t n_mine = (uint32_t)(hr_off[W] / 8);
        db->hits.reserve((size_t)std::max<uint32_t>(n_mine, 1) * 8);      // (the scatter above has been queued: stream order keeps it safe)
        if (ctx->shard_reduce == 0) {
            ScopedKernelTimer t(ctx, "exchange");
            comm->all_to_all(db->x_send.p, hs_off.data(), db->hits.p, hr_off.data(), st);
            db->x_hit_bytes += hs_off[W] - (hs_off[me + 1] - hs_off[me]);
        } else {
            // "shard_reduce" = "allgather" (round 6): every rank's whole grouped buffer, padded to the longest, to every rank in ONE
            // all-gather; this rank keeps its group from each (shard_plan.h plan_hits_gather: the all-to-all's receive layout)
            ScopedKernelTimer t(ctx, "exchange");
            const shardplan::GatherPlan gp = shardplan::plan_hits_gather(sizes.data(), W, me);
            db->x_send.grow_keep(gp.pad_bytes + 64, hs_off[W], st);      // (the scatter filled hs_off[W] bytes: padded up, contents kept)
            db->x_recv.reserve((size_t)W * gp.pad_bytes + 64);
            if (gp.pad_bytes) comm->all_gather(db->x_send.p, db->x_recv.p, gp.pad_bytes, st);
            for (uint32_t r = 0; r < W; r++)
                if (gp.len[r]) SY_HIP(hipMemcpyAsync((char*)db->hits.p + hr_off[r], (const char*)db->x_recv.p + gp.src_off[r], gp.len[r], hipMemcpyDeviceToDevice, st));
            db->x_hit_bytes += (uint64_t)(W - 1) * gp.pad_bytes;
        }
        ph.reset(); ph.reset(new HostPhase(ctx, "shard 6: sort + assemble + copy out"));
        // ---- 6. sort + assemble this rank's samples
        finish_hits(db, n_mine, max_mine, (uint64_t)n_local * G, cov_width, false, dst);
        fill_views(dst ? *dst : db->h_block, views);
        ph.reset();
        return n_mine;
    }
}

extern "C" {

int sylph_db_exchange_stats(sylph_db* db, uint64_t* batches, uint64_t* table_bytes_sent, uint64_t* hit_bytes_sent, int reset) {
    return guarded([&] {
        SY_REQUIRE(db, "null argument");
        std::lock_guard<std::mutex> lock(db->ctx->mu);
        if (batches) *batches = db->x_batches;
        if (table_bytes_sent) *table_bytes_sent = db->x_table_bytes;
        if (hit_bytes_sent) *hit_bytes_sent = db->x_hit_bytes;
        if (reset) db->x_batches = db->x_table_bytes = db->x_hit_bytes = 0;
    });
}

int sylph_db_contain_batch_sharded(sylph_db* db, sylph_comm* comm, const sylph_sample_ref* samples, uint32_t n_local, int mem,
                                   double min_number_kmers, const uint32_t** contain_count, const uint64_t** cov_off,
                                   const void** covs, uint32_t* cov_width, uint64_t* out_n_covs) {
    return guarded([&] {
        SY_REQUIRE(db && comm && contain_count && cov_off && covs && cov_width, "null argument");
        ResultViews v;   // taken under the context lock inside the call
        const uint32_t n_mine = contain_batch_sharded_impl(db, comm, samples, n_local, mem, min_number_kmers, cov_width, nullptr, &v);
        *cov_off = v.cov_off;
        *contain_count = v.contain_count;
        *covs = v.covs;
        if (out_n_covs) *out_n_covs = n_mine;
    });
}

}  // extern "C"
