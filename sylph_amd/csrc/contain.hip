// contain.hip — sample-vs-database containment on gfx950 (probe half of get_stats, contain.rs:601-656).
//
// The reference walks every genome's k-mers and probes the sample's FxHashMap (contain.rs:632-652): ~1.4e9 random
// probes per sample at GTDB-R220 scale.  Here the database lives in HBM as ONE postings array sorted by k-mer,
//     db_kmer[N] ascending,  db_gid[N] (genome of each posting),  bucket_start[] (radix index on the top bits),
// built once at upload, and the (much smaller, already sorted) sample table is streamed against it: each sample
// k-mer finds its bucket (one 8 B index read), scans the few postings in it, and emits one (genome, count) hit per
// matching posting.  Hits are radix-sorted by (genome, count): that yields contain_count[g] and the per-genome
// coverage vectors already in the ascending order the reference sorts them into (contain.rs:661).
// Same outputs, O(sample) instead of O(database) work per sample.  Pure integer work, HBM/L2-latency bound.
#include <algorithm>
#include <memory>

#include "common.h"
#include "device_common.h"

namespace sylph {
namespace {

// gid[i] = genome of posting i: one workgroup per genome streams its id over the genome's range
__global__ __launch_bounds__(256) void fill_gid_kernel(const uint64_t* __restrict__ genome_off, uint64_t n_genomes,
                                                       uint32_t* __restrict__ gid) {
    for (uint64_t g = blockIdx.x; g < n_genomes; g += gridDim.x) {
        const uint64_t b = genome_off[g], e = genome_off[g + 1];
        for (uint64_t i = b + threadIdx.x; i < e; i += blockDim.x) gid[i] = (uint32_t)g;
    }
}

__global__ __launch_bounds__(256) void genome_len_kernel(const uint64_t* __restrict__ genome_off, uint64_t n_genomes,
                                                         uint32_t* __restrict__ glen) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_genomes) glen[g] = (uint32_t)(genome_off[g + 1] - genome_off[g]);
}

// bucket_start[b] = first posting whose (kmer >> shift) >= b, for b in [0, n_buckets]
__global__ __launch_bounds__(256) void bucket_index_kernel(const uint64_t* __restrict__ keys, uint32_t n, int shift,
                                                           uint32_t n_buckets, uint32_t* __restrict__ bucket_start) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const uint64_t lo = (i == 0) ? 0 : (keys[i - 1] >> shift) + 1;          // first bucket not yet started
    const uint64_t hi = (i == n) ? (uint64_t)n_buckets : (keys[i] >> shift);   // last bucket that starts at i
    for (uint64_t b = lo; b <= hi && b <= n_buckets; b++) bucket_start[b] = i;
}

// Probe: one lane per sample k-mer, persistent workgroups striding over 256-k-mer chunks.  Hits are (gid << 32 | count),
// staged per workgroup in LDS and flushed with one global atomic per ~PROBE_FLUSH hits: with one workgroup per chunk and
// one atomic each (7.7e3 on a single word for a 2 M-entry sample) the atomic unit (~88 single-address atomics/us on this
// chip) was 40 % of the kernel.
constexpr int PROBE_TPB = 256;
constexpr int PROBE_STAGE = 4096;
constexpr int PROBE_FLUSH = 2048;
constexpr int PROBE_GRID = 512;    // measured optimum on MI355X: 256 -> 0.22 ms, 512 -> 0.177, 1024 -> 0.199, one workgroup per chunk -> 0.22

struct HitStage {
    uint64_t stage[PROBE_STAGE];
    uint32_t cnt, base, max_count;   // max_count: largest count among this workgroup's hits since the last flush
};
__device__ __forceinline__ void hit_stage_init(HitStage& st) {
    if (threadIdx.x == 0) { st.cnt = 0; st.max_count = 0; }
    __syncthreads();
}
__device__ __forceinline__ void hit_stage_push(HitStage& st, uint64_t hit, uint64_t* __restrict__ hits, uint32_t hit_cap,
                                               uint32_t* __restrict__ hit_count) {
    const uint32_t slot = atomicAdd(&st.cnt, 1u);
    if ((uint32_t)hit > st.max_count) atomicMax(&st.max_count, (uint32_t)hit);   // (plain read first: almost never true)
    if (slot < PROBE_STAGE) st.stage[slot] = hit;
    else {                                       // a k-mer shared by thousands of genomes: straight to HBM
        const uint32_t o = atomicAdd(hit_count, 1u);
        if (o < hit_cap) hits[o] = hit;
    }
}
// called by all threads after a chunk; flushes when the stage is filling up or `force`
__device__ __forceinline__ void hit_stage_flush(HitStage& st, bool force, uint64_t* __restrict__ hits, uint32_t hit_cap,
                                                uint32_t* __restrict__ hit_count) {
    __syncthreads();
    const uint32_t n = min(st.cnt, (uint32_t)PROBE_STAGE);
    __syncthreads();                                     // everyone has read cnt before anyone pushes again
    if (n == 0 || (!force && n < PROBE_FLUSH)) return;   // uniform: all lanes saw the same cnt
    if (threadIdx.x == 0) {
        st.base = atomicAdd(hit_count, n);
        atomicMax(hit_count + 1, st.max_count);          // [1] = largest count of any hit (sizes the sort keys)
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n; t += PROBE_TPB) {
        const uint32_t o = st.base + t;
        if (o < hit_cap) hits[o] = st.stage[t];
    }
    __syncthreads();
    if (threadIdx.x == 0) st.cnt = 0;
    __syncthreads();
}

__global__ __launch_bounds__(PROBE_TPB) void probe_kernel(const uint64_t* __restrict__ s_kmers,
                                                          const uint32_t* __restrict__ s_counts, uint32_t n_sample,
                                                          const uint64_t* __restrict__ db_kmer,
                                                          const uint32_t* __restrict__ db_gid,
                                                          const uint32_t* __restrict__ bucket_start, int shift,
                                                          uint32_t n_buckets, const uint32_t* __restrict__ glen,
                                                          double min_number_kmers, uint64_t* __restrict__ hits,
                                                          uint32_t hit_cap, uint32_t* __restrict__ hit_count) {
    __shared__ HitStage st;
    hit_stage_init(st);
    const uint32_t n_chunks = (n_sample + PROBE_TPB - 1) / PROBE_TPB;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t i = chunk * PROBE_TPB + threadIdx.x;
        if (i < n_sample) {
            const uint64_t km = s_kmers[i];
            const uint32_t cnt = s_counts[i];
            const uint64_t b = km >> shift;
            if (cnt != 0 && b < n_buckets) {                                     // contain.rs:634
                uint32_t lo = bucket_start[b];
                const uint32_t end = bucket_start[b + 1];
                uint32_t hi = end;
                while (lo < hi) {                                                // lower_bound inside the bucket
                    const uint32_t mid = lo + ((hi - lo) >> 1);
                    if (db_kmer[mid] < km) lo = mid + 1; else hi = mid;
                }
                for (uint32_t j = lo; j < end && db_kmer[j] == km; j++) {
                    const uint32_t g = db_gid[j];
                    if ((double)glen[g] < min_number_kmers) continue;            // contain.rs:627
                    hit_stage_push(st, ((uint64_t)g << 32) | cnt, hits, hit_cap, hit_count);
                }
            }
        }
        hit_stage_flush(st, chunk + gridDim.x >= n_chunks, hits, hit_cap, hit_count);
    }
}

// equal range of `km` in a bucketed postings index: [lo, hi)
__device__ __forceinline__ void posting_range(const uint64_t* __restrict__ kmer, const uint32_t* __restrict__ bucket_start, int shift,
                                              uint32_t n_buckets, uint64_t km, uint32_t& lo_out, uint32_t& hi_out) {
    lo_out = hi_out = 0;
    const uint64_t b = km >> shift;
    if (b >= n_buckets) return;
    uint32_t lo = bucket_start[b];
    const uint32_t end = bucket_start[b + 1];
    uint32_t hi = end;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (kmer[mid] < km) lo = mid + 1; else hi = mid;
    }
    uint32_t e = lo;
    while (e < end && kmer[e] == km) e++;
    lo_out = lo;
    hi_out = e;
}

// Profile reassignment on device: winner_table (contain.rs:410-430) + the second get_stats pass (contain.rs:300-307 with
// the winner map, :637-646).  A k-mer belongs to the passing genome with the highest first-pass ANI among those that
// hold it in genome_kmers or in pseudotax_tracked_nonused_kmers; ties go to the genome that comes first in the passing
// list (the reference replaces only on strictly greater ANI, :417).  For every passing genome, sample k-mers of its
// genome_kmers that it does not own count as kmers_lost, the others yield (genome, count) hits as in the first pass.
__global__ __launch_bounds__(PROBE_TPB) void reassign_kernel(
    const uint64_t* __restrict__ s_kmers, const uint32_t* __restrict__ s_counts, uint32_t n_sample, const uint64_t* __restrict__ db_kmer,
    const uint32_t* __restrict__ db_gid, const uint32_t* __restrict__ bucket_start, int shift, uint32_t n_buckets,
    const uint64_t* __restrict__ t_kmer, const uint32_t* __restrict__ t_gid, const uint32_t* __restrict__ t_bucket_start, int t_shift,
    uint32_t t_n_buckets, const uint32_t* __restrict__ rank, const double* __restrict__ ani, uint32_t* __restrict__ lost,
    uint64_t* __restrict__ hits, uint32_t hit_cap, uint32_t* __restrict__ hit_count) {
    __shared__ HitStage st;
    hit_stage_init(st);
    const uint32_t n_chunks = (n_sample + PROBE_TPB - 1) / PROBE_TPB;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t i = chunk * PROBE_TPB + threadIdx.x;
        const uint32_t cnt = i < n_sample ? s_counts[i] : 0;
        if (cnt != 0) {                                                          // contain.rs:634
            const uint64_t km = s_kmers[i];
            uint32_t a0, a1, t0 = 0, t1 = 0;
            posting_range(db_kmer, bucket_start, shift, n_buckets, km, a0, a1);
            if (a1 > a0) {
                if (t_n_buckets) posting_range(t_kmer, t_bucket_start, t_shift, t_n_buckets, km, t0, t1);
                uint32_t best_rank = 0xFFFFFFFFu;
                double best_ani = -1.0;
                auto consider = [&](uint32_t g) {
                    const uint32_t r = rank[g];
                    if (r == 0xFFFFFFFFu) return;
                    const double a = ani[r];
                    if (a > best_ani || (a == best_ani && r < best_rank)) { best_ani = a; best_rank = r; }
                };
                for (uint32_t j = a0; j < a1; j++) consider(db_gid[j]);
                for (uint32_t j = t0; j < t1; j++) consider(t_gid[j]);
                for (uint32_t j = a0; j < a1; j++) {
                    const uint32_t g = db_gid[j];
                    const uint32_t r = rank[g];
                    if (r == 0xFFFFFFFFu) continue;                              // not in remaining_genomes
                    if (r != best_rank) { atomicAdd(&lost[g], 1u); continue; }   // contain.rs:639-642
                    hit_stage_push(st, ((uint64_t)g << 32) | cnt, hits, hit_cap, hit_count);
                }
            }
        }
        hit_stage_flush(st, chunk + gridDim.x >= n_chunks, hits, hit_cap, hit_count);
    }
}

// cov_off[g] = first sorted hit with genome id >= g; covs[i] = low 32 bits of hit i
__global__ __launch_bounds__(256) void hit_offsets_kernel(const uint64_t* __restrict__ hits, uint32_t n_hits,
                                                          uint32_t n_genomes, uint64_t* __restrict__ cov_off,
                                                          uint32_t* __restrict__ contain_count) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_genomes) return;
    auto lower = [&](uint64_t key) {
        uint32_t lo = 0, hi = n_hits;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (hits[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const uint32_t a = lower((uint64_t)g << 32);
    cov_off[g] = a;
    if (g < n_genomes) contain_count[g] = lower(((uint64_t)g + 1) << 32) - a;
}

// Compact hit keys: (genome << 32 | count) -> 32-bit (genome << cb | count) with cb = bit_length(max count), whenever
// bit_length(G) + cb <= 32 (always at GTDB scale unless a count exceeds 2^15): the radix sort of the hit list then moves
// 4-byte keys through 3-4 passes instead of 8-byte keys through 8.  Otherwise the 64-bit keys are sorted as they are.
__global__ __launch_bounds__(256) void pack_hits32_kernel(const uint64_t* __restrict__ hits, uint32_t n, int cb, uint32_t* __restrict__ k32) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint64_t h = hits[i]; k32[i] = ((uint32_t)(h >> 32) << cb) | (uint32_t)h; }
}
template <class T>
__global__ __launch_bounds__(256) void narrow32_kernel(const uint32_t* __restrict__ k32, uint32_t n, int cb, T* __restrict__ covs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) covs[i] = (T)(k32[i] & ((1u << cb) - 1u));
}
__global__ __launch_bounds__(256) void hit_offsets32_kernel(const uint32_t* __restrict__ k32, uint32_t n_hits, uint32_t n_genomes, int cb,
                                                            uint64_t* __restrict__ cov_off, uint32_t* __restrict__ contain_count) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_genomes) return;
    auto lower = [&](uint64_t key) {   // first hit whose (genome << cb | count) >= key
        uint32_t lo = 0, hi = n_hits;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if ((uint64_t)k32[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const uint32_t a = lower((uint64_t)g << cb);
    cov_off[g] = a;
    if (g < n_genomes) contain_count[g] = lower(((uint64_t)g + 1) << cb) - a;
}
__global__ __launch_bounds__(256) void narrow_kernel(const uint64_t* __restrict__ hits, uint32_t n, uint32_t* __restrict__ covs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) covs[i] = (uint32_t)hits[i];
}

}  // namespace
}  // namespace sylph

using namespace sylph;

// Layout of the result block, identical on the device (one buffer, ONE device->host copy) and in pinned host memory:
// [cov_off (G+1) u64 | contain_count G u32 | covs n_hits x width | pad to 8 | kmers_lost G u32 (reassign only, second copy)].
struct ResultLayout {
    size_t ccount = 0, covs = 0, lost = 0, end = 0;
    ResultLayout() = default;
    ResultLayout(uint64_t G, uint64_t n_hits, uint32_t width, bool reassign)
        : ccount((G + 1) * 8), covs(ccount + G * 4), lost((covs + n_hits * width + 7) & ~(size_t)7), end(lost + (reassign ? G * 4 : 0)) {}
};

struct sylph_db {
    sylph_ctx* ctx;
    uint64_t n_genomes = 0, n_kmers = 0;
    int shift = 0;
    uint32_t n_buckets = 0;
    DevBuf kmer, gid, bucket_start, glen;
    // optional second postings index over pseudotax_tracked_nonused_kmers (types.rs:166), only used by the winner table
    DevBuf t_kmer, t_gid, t_bucket_start;
    int t_shift = 0;
    uint32_t t_n_buckets = 0;
    uint64_t t_n = 0;
    DevBuf rank, ani, lost;        // reassign pass: rank[g] in the passing list (or ~0), ANI per rank, kmers_lost[g]
    // per-query scratch (owned by the db so concurrent dbs on one ctx do not alias)
    DevBuf q_kmers, q_counts, hits, hits_sorted, res, counter;   // res: device copy of the result block
    ResultLayout lay;              // layout of the last result
    void* h_res = nullptr;         // pinned host results: [cov_off (G+1) u64 | contain_count G u32 | covs u32]
    size_t h_res_cap = 0;
    ~sylph_db() { if (h_res) (void)hipHostFree(h_res); }
    explicit sylph_db(sylph_ctx* cx)
        : ctx(cx), kmer(cx), gid(cx), bucket_start(cx), glen(cx), t_kmer(cx), t_gid(cx), t_bucket_start(cx), rank(cx),
          ani(cx), lost(cx), q_kmers(cx), q_counts(cx), hits(cx), hits_sorted(cx),
          res(cx), counter(cx) {}
};

static uint32_t grid_for64(uint64_t n, uint32_t tpb = 256) { return (uint32_t)((n + tpb - 1) / tpb); }

// Builds a postings index (k-mers sorted, genome id per posting, bucket table) from genome-major device arrays.
static void build_postings(sylph_ctx* ctx, const uint64_t* d_kmers_in, const uint64_t* d_off, uint64_t n_genomes, uint64_t n,
                           DevBuf& kmer, DevBuf& gid, DevBuf& bucket_start, int& shift, uint32_t& n_buckets) {
    ScopedKernelTimer t(ctx, "db_index");
    DevBuf gid_in(ctx);
    gid_in.reserve(n * 4);
    kmer.reserve(n * 8);
    gid.reserve(n * 4);
    hipLaunchKernelGGL(fill_gid_kernel, dim3((uint32_t)std::min<uint64_t>(n_genomes, 1u << 20)), dim3(256), 0, ctx->stream, d_off,
                       n_genomes, gid_in.as<uint32_t>());
    sort_pairs_u64_u32(ctx, d_kmers_in, kmer.as<uint64_t>(), gid_in.as<uint32_t>(), gid.as<uint32_t>(), n, 0, 64);
    uint64_t max_key = 0;
    ctx->read_back(&max_key, kmer.as<uint64_t>() + (n - 1), 8);
    // ~2-3 postings per bucket on average, index <= 2^30 entries (measured on MI355X at 1.8e9 postings: 8 per bucket 0.177 ms
    // per probe of a 2 M-entry sample, 2 per bucket 0.166 ms, 32 per bucket 0.20 ms: the dependent loads inside the bucket
    // cost more than the larger table)
    static const uint64_t ppb = getenv("SYLPH_HIP_POSTINGS_PER_BUCKET") ? std::max(1, atoi(getenv("SYLPH_HIP_POSTINGS_PER_BUCKET"))) : 2;
    int b = bit_length(n / ppb);
    b = std::min(getenv("SYLPH_HIP_BUCKET_BITS_MAX") ? atoi(getenv("SYLPH_HIP_BUCKET_BITS_MAX")) : 30, std::max(8, b));
    const int bits = std::max(1, bit_length(max_key));
    shift = std::max(0, bits - b);
    n_buckets = (uint32_t)((max_key >> shift) + 1);
    bucket_start.reserve(((size_t)n_buckets + 1) * 4);
    hipLaunchKernelGGL(bucket_index_kernel, dim3(grid_for64(n + 1)), dim3(256), 0, ctx->stream, kmer.as<uint64_t>(), (uint32_t)n,
                       shift, n_buckets, bucket_start.as<uint32_t>());
    SY_HIP(hipGetLastError());
    SY_HIP(hipStreamSynchronize(ctx->stream));   // the caller's staging buffers / gid_in are released on return
}

// Stages genome-major (k-mers, offsets) on the device if they are host arrays; returns the device pointers and the total.
static uint64_t stage_genome_major(sylph_ctx* ctx, const uint64_t* kmers, const uint64_t* off, uint64_t n_genomes, int mem,
                                   DevBuf& d_off_buf, DevBuf& d_in, const uint64_t*& d_off, const uint64_t*& d_kmers) {
    uint64_t n = 0;
    d_off = nullptr;
    d_kmers = nullptr;
    if (!n_genomes) return 0;
    if (mem == SYLPH_MEM_HOST) {
        SY_REQUIRE(off[0] == 0, "offsets[0] must be 0");
        n = off[n_genomes];
        d_off_buf.reserve((n_genomes + 1) * 8);
        ctx->h2d(d_off_buf.p, off, (n_genomes + 1) * 8);
        d_off = d_off_buf.as<uint64_t>();
        if (n) {
            SY_REQUIRE(kmers, "null kmers");
            d_in.reserve(n * 8);
            ctx->h2d(d_in.p, kmers, n * 8);
            d_kmers = d_in.as<uint64_t>();
        }
    } else {
        ctx->read_back(&n, off + n_genomes, 8);
        d_off = off;
        d_kmers = kmers;
    }
    return n;
}

extern "C" {

int sylph_db_upload(sylph_ctx* ctx, const uint64_t* kmers, const uint64_t* genome_off, uint64_t n_genomes, int mem,
                    sylph_db** out) {
    return guarded([&] {
        SY_REQUIRE(ctx && out, "null argument");
        SY_REQUIRE(mem == SYLPH_MEM_HOST || mem == SYLPH_MEM_DEVICE, "bad mem kind %d", mem);
        SY_REQUIRE(n_genomes < (1ull << 32), "at most 2^32-1 genomes per shard");
        SY_REQUIRE(n_genomes == 0 || genome_off, "null genome_off");
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard dg(ctx->device);
        std::unique_ptr<sylph_db> db(new sylph_db(ctx));
        db->n_genomes = n_genomes;
        db->counter.reserve(64);
        DevBuf d_off_buf(ctx), d_in(ctx);
        const uint64_t* d_off = nullptr;
        const uint64_t* d_kmers_in = nullptr;
        const uint64_t n = stage_genome_major(ctx, kmers, genome_off, n_genomes, mem, d_off_buf, d_in, d_off, d_kmers_in);
        SY_REQUIRE(n < (1ull << 32), "at most 2^32-1 k-mers per shard (got %llu): shard the database", (unsigned long long)n);
        db->n_kmers = n;
        db->glen.reserve(std::max<uint64_t>(1, n_genomes) * 4);
        if (n_genomes)
            hipLaunchKernelGGL(genome_len_kernel, dim3(grid_for64(n_genomes)), dim3(256), 0, ctx->stream, d_off, n_genomes,
                               db->glen.as<uint32_t>());
        if (n) build_postings(ctx, d_kmers_in, d_off, n_genomes, n, db->kmer, db->gid, db->bucket_start, db->shift, db->n_buckets);
        ctx->refs++;
        *out = db.release();
    });
}

int sylph_db_attach_tracked(sylph_db* db, const uint64_t* tracked_kmers, const uint64_t* tracked_off, int mem) {
    return guarded([&] {
        SY_REQUIRE(db && tracked_off, "null argument");
        SY_REQUIRE(mem == SYLPH_MEM_HOST || mem == SYLPH_MEM_DEVICE, "bad mem kind %d", mem);
        sylph_ctx* ctx = db->ctx;
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard dg(ctx->device);
        DevBuf d_off_buf(ctx), d_in(ctx);
        const uint64_t* d_off = nullptr;
        const uint64_t* d_k = nullptr;
        const uint64_t n = stage_genome_major(ctx, tracked_kmers, tracked_off, db->n_genomes, mem, d_off_buf, d_in, d_off, d_k);
        SY_REQUIRE(n < (1ull << 32), "at most 2^32-1 tracked k-mers per shard");
        db->t_n = n;
        db->t_n_buckets = 0;
        if (n) build_postings(ctx, d_k, d_off, db->n_genomes, n, db->t_kmer, db->t_gid, db->t_bucket_start, db->t_shift, db->t_n_buckets);
    });
}

uint64_t sylph_db_n_genomes(const sylph_db* db) { return db ? db->n_genomes : 0; }
uint64_t sylph_db_n_kmers(const sylph_db* db) { return db ? db->n_kmers : 0; }

// Runs the probe and leaves (cov_off, contain_count, covs) in db->h_res (pinned).  Returns the number of hits.
static uint32_t probe_grid() {
    static const uint32_t g = getenv("SYLPH_HIP_PROBE_GRID") ? (uint32_t)atoi(getenv("SYLPH_HIP_PROBE_GRID")) : PROBE_GRID;
    return std::max<uint32_t>(1, g);
}

struct ReassignArgs { const uint32_t* passing_gids; const double* passing_ani; uint32_t n_passing; };

// cov_width: nullptr = coverage values as u32; else in/out — the values are stored with the narrowest of 1, 2 or 4 bytes
// that holds the sample's largest count (7.4 MB -> 1.9 MB over PCIe per sample at GTDB scale) and the width is returned.
static uint32_t contain_impl(sylph_db* db, const uint64_t* sample_kmers, const uint32_t* sample_counts, uint64_t n, int mem,
                             double min_number_kmers, const ReassignArgs* re = nullptr, uint32_t* cov_width = nullptr) {
    SY_REQUIRE(mem == SYLPH_MEM_HOST || mem == SYLPH_MEM_DEVICE, "bad mem kind %d", mem);
    SY_REQUIRE(n < (1ull << 32), "sample table larger than 2^32-1 entries");
    sylph_ctx* ctx = db->ctx;
    const uint64_t G = db->n_genomes;
    uint32_t n_hits = 0, max_count = 0;
    if (re) {   // rank[g] = position of genome g in the passing list (or ~0), ani[rank], lost[g] = 0
        SY_REQUIRE(re->n_passing == 0 || (re->passing_gids && re->passing_ani), "null passing list");
        std::vector<uint32_t> rank(std::max<uint64_t>(1, G), 0xFFFFFFFFu);
        for (uint32_t r = 0; r < re->n_passing; r++) {
            SY_REQUIRE(re->passing_gids[r] < G, "passing genome id %u out of range", re->passing_gids[r]);
            SY_REQUIRE(rank[re->passing_gids[r]] == 0xFFFFFFFFu, "genome %u listed twice", re->passing_gids[r]);
            rank[re->passing_gids[r]] = r;
        }
        db->rank.reserve(rank.size() * 4);
        db->ani.reserve(std::max<size_t>(1, re->n_passing) * 8);
        db->lost.reserve(rank.size() * 4);
        ctx->h2d(db->rank.p, rank.data(), rank.size() * 4);
        if (re->n_passing) ctx->h2d(db->ani.p, re->passing_ani, (size_t)re->n_passing * 8);
        SY_HIP(hipMemsetAsync(db->lost.p, 0, rank.size() * 4, ctx->stream));
    }
    if (n && db->n_kmers && (!re || re->n_passing)) {
        SY_REQUIRE(sample_kmers && sample_counts, "null sample");
        const uint64_t* d_k = sample_kmers;
        const uint32_t* d_c = sample_counts;
        if (mem == SYLPH_MEM_HOST) {
            db->q_kmers.reserve(n * 8);
            db->q_counts.reserve(n * 4);
            ctx->h2d(db->q_kmers.p, sample_kmers, n * 8);
            ctx->h2d(db->q_counts.p, sample_counts, n * 4);
            d_k = db->q_kmers.as<uint64_t>();
            d_c = db->q_counts.as<uint32_t>();
        }
        uint64_t cap = std::max<uint64_t>(n * 2, 1u << 20);
        uint32_t* d_cnt = db->counter.as<uint32_t>();   // [0] = number of hits, [1] = largest count among the hits
        for (int attempt = 0; attempt < 2; attempt++) {
            SY_REQUIRE(cap < (1ull << 32), "more than 2^32-1 hits for one sample");
            db->hits.reserve(cap * 8);
            SY_HIP(hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
            {
                ScopedKernelTimer t(ctx, "probe");
                if (!re)
                    hipLaunchKernelGGL(probe_kernel, dim3(std::min<uint32_t>(grid_for64(n, PROBE_TPB), probe_grid())), dim3(PROBE_TPB), 0, ctx->stream, d_k, d_c,
                                       (uint32_t)n, db->kmer.as<uint64_t>(), db->gid.as<uint32_t>(),
                                       db->bucket_start.as<uint32_t>(), db->shift, db->n_buckets, db->glen.as<uint32_t>(),
                                       min_number_kmers, db->hits.as<uint64_t>(), (uint32_t)cap, d_cnt);
                else {
                    if (attempt) SY_HIP(hipMemsetAsync(db->lost.p, 0, std::max<uint64_t>(1, G) * 4, ctx->stream));
                    hipLaunchKernelGGL(reassign_kernel, dim3(std::min<uint32_t>(grid_for64(n, PROBE_TPB), probe_grid())), dim3(PROBE_TPB), 0, ctx->stream, d_k, d_c,
                                       (uint32_t)n, db->kmer.as<uint64_t>(), db->gid.as<uint32_t>(),
                                       db->bucket_start.as<uint32_t>(), db->shift, db->n_buckets, db->t_kmer.as<uint64_t>(),
                                       db->t_gid.as<uint32_t>(), db->t_bucket_start.as<uint32_t>(), db->t_shift, db->t_n_buckets,
                                       db->rank.as<uint32_t>(), db->ani.as<double>(), db->lost.as<uint32_t>(),
                                       db->hits.as<uint64_t>(), (uint32_t)cap, d_cnt);
                }
                SY_HIP(hipGetLastError());
            }
            uint32_t hc[2] = {0, 0};
            ctx->read_back(hc, d_cnt, 8);
            n_hits = hc[0];
            max_count = hc[1];
            if (n_hits <= cap) break;
            SY_REQUIRE(attempt == 0, "hit buffer overflow persisted");
            cap = n_hits;
        }
    }
    const int cb = std::max(1, bit_length(max_count)), gb = std::max(1, bit_length(G));
    uint32_t width = 4;
    if (cov_width && n_hits && cb + gb <= 32) width = cb <= 8 ? 1 : cb <= 16 ? 2 : 4;
    const ResultLayout lay(G, n_hits, width, re != nullptr);
    db->lay = lay;
    db->res.reserve(lay.lost + 64);
    char* d_res = db->res.as<char>();
    uint64_t* d_cov_off = reinterpret_cast<uint64_t*>(d_res);
    uint32_t* d_ccount = reinterpret_cast<uint32_t*>(d_res + lay.ccount);
    void* d_covs = d_res + lay.covs;
    if (n_hits && cb + gb <= 32) {
        db->hits_sorted.reserve((size_t)n_hits * 8);   // two u32 arrays: packed keys, sorted keys
        uint32_t* k32 = db->hits_sorted.as<uint32_t>();
        uint32_t* k32s = k32 + n_hits;
        hipLaunchKernelGGL(pack_hits32_kernel, dim3(grid_for64(n_hits)), dim3(256), 0, ctx->stream, db->hits.as<uint64_t>(), n_hits, cb,
                           k32);
        sort_keys_u32(ctx, k32, k32s, n_hits, 0, cb + gb);
        if (width == 1) hipLaunchKernelGGL(narrow32_kernel<uint8_t>, dim3(grid_for64(n_hits)), dim3(256), 0, ctx->stream, k32s, n_hits, cb, (uint8_t*)d_covs);
        else if (width == 2) hipLaunchKernelGGL(narrow32_kernel<uint16_t>, dim3(grid_for64(n_hits)), dim3(256), 0, ctx->stream, k32s, n_hits, cb, (uint16_t*)d_covs);
        else hipLaunchKernelGGL(narrow32_kernel<uint32_t>, dim3(grid_for64(n_hits)), dim3(256), 0, ctx->stream, k32s, n_hits, cb, (uint32_t*)d_covs);
        hipLaunchKernelGGL(hit_offsets32_kernel, dim3(grid_for64(G + 1)), dim3(256), 0, ctx->stream, k32s, n_hits, (uint32_t)G, cb,
                           d_cov_off, d_ccount);
    } else {
        const uint64_t* d_sorted = nullptr;
        if (n_hits) {
            db->hits_sorted.reserve((size_t)n_hits * 8);
            sort_keys_u64(ctx, db->hits.as<uint64_t>(), db->hits_sorted.as<uint64_t>(), n_hits, 0, 64);
            d_sorted = db->hits_sorted.as<uint64_t>();
            hipLaunchKernelGGL(narrow_kernel, dim3(grid_for64(n_hits)), dim3(256), 0, ctx->stream, d_sorted, n_hits,
                               (uint32_t*)d_covs);
        }
        hipLaunchKernelGGL(hit_offsets_kernel, dim3(grid_for64(G + 1)), dim3(256), 0, ctx->stream, d_sorted, n_hits, (uint32_t)G,
                           d_cov_off, d_ccount);
    }
    SY_HIP(hipGetLastError());
    // straight into pinned host memory owned by the db (no staging copy, nothing pageable registered with HIP)
    const size_t need = lay.end + 64;
    if (need > db->h_res_cap) {
        if (db->h_res) SY_HIP(hipHostFree(db->h_res));
        db->h_res = nullptr;
        db->h_res_cap = 0;
        SY_HIP(hipHostMalloc(&db->h_res, need + need / 2, hipHostMallocDefault));
        db->h_res_cap = need + need / 2;
    }
    char* h = (char*)db->h_res;
    SY_HIP(hipMemcpyAsync(h, d_res, lay.covs + (size_t)n_hits * width, hipMemcpyDeviceToHost, ctx->stream));   // one copy
    if (re && G) SY_HIP(hipMemcpyAsync(h + lay.lost, db->lost.p, G * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (cov_width) *cov_width = width;
    SY_HIP(hipStreamSynchronize(ctx->stream));
    if (!ctx->pending.empty()) profile_collect(ctx);
    return n_hits;
}

int sylph_db_contain_view(sylph_db* db, const uint64_t* sample_kmers, const uint32_t* sample_counts, uint64_t n, int mem,
                          double min_number_kmers, const uint32_t** contain_count, const uint64_t** cov_off,
                          const uint32_t** covs, uint64_t* out_n_covs) {
    return guarded([&] {
        SY_REQUIRE(db && contain_count && cov_off && covs, "null argument");
        std::lock_guard<std::mutex> lock(db->ctx->mu);
        DeviceGuard dg(db->ctx->device);
        const uint32_t n_hits = contain_impl(db, sample_kmers, sample_counts, n, mem, min_number_kmers);
        const ResultLayout& lay = db->lay;
        const char* h = (const char*)db->h_res;
        *cov_off = (const uint64_t*)h;
        *contain_count = (const uint32_t*)(h + lay.ccount);
        *covs = (const uint32_t*)(h + lay.covs);
        if (out_n_covs) *out_n_covs = n_hits;
    });
}

int sylph_db_contain_view_packed(sylph_db* db, const uint64_t* sample_kmers, const uint32_t* sample_counts, uint64_t n, int mem,
                                 double min_number_kmers, const uint32_t** contain_count, const uint64_t** cov_off,
                                 const void** covs, uint32_t* cov_width, uint64_t* out_n_covs) {
    return guarded([&] {
        SY_REQUIRE(db && contain_count && cov_off && covs && cov_width, "null argument");
        std::lock_guard<std::mutex> lock(db->ctx->mu);
        DeviceGuard dg(db->ctx->device);
        const uint32_t n_hits = contain_impl(db, sample_kmers, sample_counts, n, mem, min_number_kmers, nullptr, cov_width);
        const ResultLayout& lay = db->lay;
        const char* h = (const char*)db->h_res;
        *cov_off = (const uint64_t*)h;
        *contain_count = (const uint32_t*)(h + lay.ccount);
        *covs = h + lay.covs;
        if (out_n_covs) *out_n_covs = n_hits;
    });
}

int sylph_db_reassign_view(sylph_db* db, const uint64_t* sample_kmers, const uint32_t* sample_counts, uint64_t n, int mem,
                           const uint32_t* passing_gids, const double* passing_ani, uint32_t n_passing,
                           const uint32_t** contain_count, const uint64_t** cov_off, const uint32_t** covs, uint64_t* out_n_covs,
                           const uint32_t** kmers_lost) {
    return guarded([&] {
        SY_REQUIRE(db && contain_count && cov_off && covs && kmers_lost, "null argument");
        std::lock_guard<std::mutex> lock(db->ctx->mu);
        DeviceGuard dg(db->ctx->device);
        ReassignArgs re{passing_gids, passing_ani, n_passing};
        const uint32_t n_hits = contain_impl(db, sample_kmers, sample_counts, n, mem, 0.0, &re);
        const ResultLayout& lay = db->lay;
        const char* h = (const char*)db->h_res;
        *cov_off = (const uint64_t*)h;
        *contain_count = (const uint32_t*)(h + lay.ccount);
        *covs = (const uint32_t*)(h + lay.covs);
        *kmers_lost = (const uint32_t*)(h + lay.lost);
        if (out_n_covs) *out_n_covs = n_hits;
    });
}

int sylph_db_contain(sylph_db* db, const uint64_t* sample_kmers, const uint32_t* sample_counts, uint64_t n, int mem,
                     double min_number_kmers, uint32_t* contain_count, uint64_t* cov_off, uint32_t** out_covs) {
    return guarded([&] {
        SY_REQUIRE(db && contain_count && cov_off && out_covs, "null argument");
        std::lock_guard<std::mutex> lock(db->ctx->mu);
        DeviceGuard dg(db->ctx->device);
        const uint32_t n_hits = contain_impl(db, sample_kmers, sample_counts, n, mem, min_number_kmers);
        const uint64_t G = db->n_genomes;
        const ResultLayout& lay = db->lay;
        const char* h = (const char*)db->h_res;
        uint32_t* hcov = (uint32_t*)malloc(std::max<size_t>(1, n_hits) * 4);
        if (!hcov) throw std::bad_alloc();
        memcpy(cov_off, h, (G + 1) * 8);
        if (G) memcpy(contain_count, h + lay.ccount, G * 4);
        if (n_hits) memcpy(hcov, h + lay.covs, (size_t)n_hits * 4);
        *out_covs = hcov;
    });
}

void sylph_db_destroy(sylph_db* db) {
    if (!db) return;
    sylph_ctx* ctx = db->ctx;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        delete db;
    }
    ctx_unref(ctx);
}

}  // extern "C"
