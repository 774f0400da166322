// genomes.hip — sketch_genome (sketch.rs:550-622) / sketch_genome_individual (:481-548) for a BATCH of genomes, entirely on
// the device: seeding with positions (K1), the genome-wide duplicate removal (:594-605) and the greedy spacing filter
// (:602-614).  SURVEY.md §8f-3 ("genome DB build on device").
//
// The reference's formulation is sequential per genome (two hash sets, then a scan carrying last_contig/last_pos).  The
// data-parallel restatement used here:
//   * duplicates: one stable radix sort of (hash -> survivor index) over the whole batch; survivors are in (genome, contig,
//     position) order, so inside a run of equal hashes the entries of one genome are adjacent: an occurrence is a duplicate
//     iff its left or right neighbour in the run belongs to the same genome.  All occurrences of such a hash are dropped
//     (:605), other genomes of the batch are unaffected.
//   * spacing: among the remaining occurrences, one whose predecessor lies in another contig or more than min_spacing
//     before it is kept whatever the scan state is (last_pos never exceeds the predecessor's position) — an "anchor".
//     Between two anchors the recurrence is run by the anchor's thread (runs are 1-3 elements long at c = 200, spacing 30).
//     `last_pos == 0` (:606) only holds before the first kept k-mer of a genome, which is an anchor here (end positions
//     are >= k-1 > 0).
#include "sketch_session.h"
#include "device_common.h"

namespace sylph {

uint32_t seeds_sorted_by_pos(sylph_ctx* ctx, const uint8_t* d_bases, uint64_t n_bases, uint32_t c, uint32_t k, uint32_t* d_count);

namespace {

uint32_t grid_for(uint64_t n, uint32_t tpb = 256) { return (uint32_t)((n + tpb - 1) / tpb); }

// per survivor (position order): contig, end position inside the contig, genome, and the sort key (hash or INVALID)
__global__ __launch_bounds__(256) void annotate_genomes_kernel(const uint64_t* __restrict__ contig_off, uint64_t n_contigs,
                                                               const uint64_t* __restrict__ genome_contig_off, uint64_t n_genomes,
                                                               const uint32_t* __restrict__ pos, const uint64_t* __restrict__ hash,
                                                               uint32_t n, uint32_t bias, uint32_t k, int avx2_compat,
                                                               uint64_t* __restrict__ key,
                                                               uint32_t* __restrict__ o_contig, uint32_t* __restrict__ o_end,
                                                               uint32_t* __restrict__ o_gid, uint32_t* __restrict__ o_idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = (uint64_t)pos[i] - bias;   // K1 ran from the 16 B aligned address `bias` bytes below the batch
    uint64_t h = INVALID_HASH;
    uint32_t contig = 0, endpos = 0, gid = 0;
    if (pos[i] >= bias && p < contig_off[n_contigs]) {
        const uint64_t r = find_record(contig_off, n_contigs, p);
        const uint64_t start = contig_off[r], L = contig_off[r + 1] - start;
        if ((p - start) < n_hashed_kmers(L, k, avx2_compat, 1)) {   // contig rule: nothing below 2k (avx2_seeding.rs:160)
            h = hash[i];
            contig = (uint32_t)r;
            endpos = (uint32_t)(p - start + k - 1);                 // index of the k-mer's last base (seeding.rs:205)
            gid = (uint32_t)find_record(genome_contig_off, n_genomes, r);
        }
    }
    key[i] = h;
    o_contig[i] = contig;
    o_end[i] = endpos;
    o_gid[i] = gid;
    o_idx[i] = i;
}

// over the hash-sorted order: live[idx] = 1 iff valid and no other occurrence of the hash in the same genome
__global__ __launch_bounds__(256) void genome_dup_kernel(const uint64_t* __restrict__ key_s, const uint32_t* __restrict__ idx_s,
                                                         const uint32_t* __restrict__ gid, uint32_t n, uint32_t* __restrict__ live) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t h = key_s[j];
    const uint32_t i = idx_s[j];
    bool ok = h != INVALID_HASH;
    if (ok) {
        const uint32_t g = gid[i];
        if (j > 0 && key_s[j - 1] == h && gid[idx_s[j - 1]] == g) ok = false;
        if (j + 1 < n && key_s[j + 1] == h && gid[idx_s[j + 1]] == g) ok = false;
    }
    live[i] = ok ? 1u : 0u;
}

// gather the live occurrences (position order kept): contig, end position, genome, hash
__global__ __launch_bounds__(256) void compact_live_kernel(const uint32_t* __restrict__ live, const uint32_t* __restrict__ lpos,
                                                           const uint32_t* __restrict__ contig, const uint32_t* __restrict__ endp,
                                                           const uint32_t* __restrict__ gid, const uint64_t* __restrict__ hash,
                                                           uint32_t n, uint32_t* __restrict__ l_contig, uint32_t* __restrict__ l_end,
                                                           uint32_t* __restrict__ l_gid, uint64_t* __restrict__ l_hash) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !live[i]) return;
    const uint32_t j = lpos[i];
    l_contig[j] = contig[i];
    l_end[j] = endp[i];
    l_gid[j] = gid[i];
    l_hash[j] = hash[i];
}

// greedy spacing filter (sketch.rs:602-614); n_live is read from the scan's sentinel
__global__ __launch_bounds__(256) void spacing_kernel(const uint32_t* __restrict__ l_contig, const uint32_t* __restrict__ l_end,
                                                      const uint32_t* __restrict__ n_live_p, uint64_t min_spacing,
                                                      uint32_t* __restrict__ kept) {
    const uint32_t n_live = *n_live_p;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_live) return;
    auto anchor = [&](uint32_t t) {
        return t == 0 || l_contig[t] != l_contig[t - 1] || (uint64_t)(l_end[t] - l_end[t - 1]) > min_spacing;
    };
    if (!anchor(j)) return;
    kept[j] = 1u;
    uint32_t last = l_end[j];
    for (uint32_t t = j + 1; t < n_live && !anchor(t); t++) {
        if ((uint64_t)(l_end[t] - last) > min_spacing) { kept[t] = 1u; last = l_end[t]; }
        else kept[t] = 0u;
    }
}

__global__ __launch_bounds__(256) void emit_genomes_kernel(const uint32_t* __restrict__ kept, const uint32_t* __restrict__ kpos,
                                                           const uint64_t* __restrict__ l_hash, const uint32_t* __restrict__ n_live_p,
                                                           int pseudotax, uint64_t* __restrict__ out_k, uint64_t* __restrict__ out_t) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= *n_live_p) return;
    if (kept[j]) out_k[kpos[j]] = l_hash[j];
    else if (pseudotax) out_t[j - kpos[j]] = l_hash[j];   // rejected k-mers in position order (:610-611)
}

// koff[g] / toff[g] = kept / tracked k-mers of genomes < g (l_gid is non-decreasing)
__global__ __launch_bounds__(256) void genome_offsets_kernel(const uint32_t* __restrict__ l_gid, const uint32_t* __restrict__ kpos,
                                                             const uint32_t* __restrict__ n_live_p, uint64_t n_genomes,
                                                             uint64_t* __restrict__ koff, uint64_t* __restrict__ toff) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_genomes) return;
    const uint32_t n_live = *n_live_p;
    uint32_t lo = 0, hi = n_live;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)l_gid[mid] < g) lo = mid + 1; else hi = mid;
    }
    const uint32_t k = kpos[lo];   // kpos has n_live + 1 valid entries (sentinel)
    koff[g] = k;
    toff[g] = lo - k;
}

}  // namespace

struct GenomeBatchOut { std::vector<uint64_t> kmers, tracked, koff, toff; };

void sketch_genomes_impl(sylph_ctx* ctx, const uint8_t* bases, const uint64_t* contig_off, uint64_t n_contigs,
                         const uint64_t* genome_contig_off, uint64_t n_genomes, uint32_t c, uint32_t k, int seed_mode,
                         uint64_t min_spacing, int pseudotax, int mem, uint64_t** out_kmers, uint64_t* kmer_off, uint64_t** out_tracked,
                         uint64_t* tracked_off) {
    SY_REQUIRE(c >= 1, "c must be >= 1");
    SY_REQUIRE(seed_mode == SYLPH_SEED_SCALAR || seed_mode == SYLPH_SEED_AVX2_COMPAT, "bad seed_mode %d", seed_mode);
    SY_REQUIRE(k == 21 || k == 31, "k must be 21 or 31 (avx2_seeding.rs:46-52)");
    SY_REQUIRE(mem == SYLPH_MEM_HOST || mem == SYLPH_MEM_DEVICE, "bad mem kind %d", mem);
    SY_REQUIRE(contig_off && contig_off[0] == 0, "bad contig offsets");
    SY_REQUIRE(genome_contig_off && genome_contig_off[0] == 0 && genome_contig_off[n_genomes] == n_contigs,
               "genome_contig_off must start at 0 and end at n_contigs");
    SY_REQUIRE(n_contigs < (1ull << 32) && n_genomes < (1ull << 32), "too many contigs or genomes in one batch");
    for (uint64_t i = 0; i < n_contigs; i++) SY_REQUIRE(contig_off[i] <= contig_off[i + 1], "contig offsets must not decrease");
    for (uint64_t g = 0; g < n_genomes; g++)
        SY_REQUIRE(genome_contig_off[g] <= genome_contig_off[g + 1], "genome_contig_off must not decrease");
    const uint64_t n_bases = contig_off[n_contigs];
    SY_REQUIRE(bases || n_bases == 0, "null bases");
    SY_REQUIRE(n_bases < (1ull << 32), "batch larger than 2^32-1 bases: split it");
    for (uint64_t g = 0; g <= n_genomes; g++) { kmer_off[g] = 0; if (tracked_off) tracked_off[g] = 0; }
    *out_kmers = nullptr;
    if (out_tracked) *out_tracked = nullptr;
    auto empty_out = [&] {
        *out_kmers = (uint64_t*)malloc(8);
        if (out_tracked) *out_tracked = (uint64_t*)malloc(8);
        if (!*out_kmers || (out_tracked && !*out_tracked)) throw std::bad_alloc();
    };
    if (n_bases == 0 || n_genomes == 0) { empty_out(); return; }

    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard dg(ctx->device);
    DevBuf d_bases(ctx), d_off(ctx), d_goff(ctx);
    const uint8_t* db = bases;
    if (mem == SYLPH_MEM_HOST) {
        d_bases.reserve(n_bases + 64);
        ctx->h2d(d_bases.p, bases, n_bases);
        db = d_bases.as<uint8_t>();
    }
    d_off.reserve((n_contigs + 1) * 8);
    d_goff.reserve((n_genomes + 1) * 8);
    ctx->h2d(d_off.p, contig_off, (n_contigs + 1) * 8);
    ctx->h2d(d_goff.p, genome_contig_off, (n_genomes + 1) * 8);
    ctx->counters.reserve(64);
    const uint32_t bias = (uint32_t)((uintptr_t)db & 15);   // a caller's device pointer need not be 16 B aligned
    SY_REQUIRE(n_bases + bias < (1ull << 32), "batch larger than 2^32-1 bases: split it");
    const uint32_t n = seeds_sorted_by_pos(ctx, db - bias, n_bases + bias, c, k, ctx->counters.as<uint32_t>());
    if (!n) { empty_out(); return; }
    const uint32_t* s_pos = ctx->scratch[2].as<uint32_t>();
    const uint64_t* s_hash = ctx->scratch[3].as<uint64_t>();

    DevBuf b_key(ctx), b_key_s(ctx), b_idx(ctx), b_idx_s(ctx), b_contig(ctx), b_end(ctx), b_gid(ctx), b_live(ctx), b_lpos(ctx);
    const size_t n1 = (size_t)n + 1;
    b_key.reserve((size_t)n * 8); b_key_s.reserve((size_t)n * 8);
    b_idx.reserve((size_t)n * 4); b_idx_s.reserve((size_t)n * 4);
    b_contig.reserve((size_t)n * 4); b_end.reserve((size_t)n * 4); b_gid.reserve((size_t)n * 4);
    b_live.reserve(n1 * 4); b_lpos.reserve(n1 * 4);
    {
        ScopedKernelTimer t(ctx, "annotate");
        hipLaunchKernelGGL(annotate_genomes_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_off.as<uint64_t>(), n_contigs,
                           d_goff.as<uint64_t>(), n_genomes, s_pos, s_hash, n, bias, k, seed_mode == SYLPH_SEED_AVX2_COMPAT,
                           b_key.as<uint64_t>(), b_contig.as<uint32_t>(), b_end.as<uint32_t>(), b_gid.as<uint32_t>(),
                           b_idx.as<uint32_t>());
        SY_HIP(hipGetLastError());
    }
    sort_pairs_u64_u32(ctx, b_key.as<uint64_t>(), b_key_s.as<uint64_t>(), b_idx.as<uint32_t>(), b_idx_s.as<uint32_t>(), n, 0, 64);
    DevBuf l_contig(ctx), l_end(ctx), l_gid(ctx), l_hash(ctx), b_kept(ctx), b_kpos(ctx), d_koff(ctx), d_toff(ctx);
    l_contig.reserve((size_t)n * 4); l_end.reserve((size_t)n * 4); l_gid.reserve((size_t)n * 4); l_hash.reserve((size_t)n * 8);
    b_kept.reserve(n1 * 4); b_kpos.reserve(n1 * 4);
    d_koff.reserve((n_genomes + 1) * 8); d_toff.reserve((n_genomes + 1) * 8);
    DevBuf d_out_k(ctx), d_out_t(ctx);
    d_out_k.reserve((size_t)n * 8);
    d_out_t.reserve((size_t)n * 8);
    {
        ScopedKernelTimer t(ctx, "genome_filter");
        SY_HIP(hipMemsetAsync(b_live.as<uint32_t>() + n, 0, 4, ctx->stream));   // scan sentinels
        SY_HIP(hipMemsetAsync(b_kept.p, 0, n1 * 4, ctx->stream));
        hipLaunchKernelGGL(genome_dup_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, b_key_s.as<uint64_t>(), b_idx_s.as<uint32_t>(),
                           b_gid.as<uint32_t>(), n, b_live.as<uint32_t>());
    }
    exclusive_sum_u32(ctx, b_live.as<uint32_t>(), b_lpos.as<uint32_t>(), n1);
    const uint32_t* d_n_live = b_lpos.as<uint32_t>() + n;
    {
        ScopedKernelTimer t(ctx, "genome_filter");
        hipLaunchKernelGGL(compact_live_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, b_live.as<uint32_t>(), b_lpos.as<uint32_t>(),
                           b_contig.as<uint32_t>(), b_end.as<uint32_t>(), b_gid.as<uint32_t>(), s_hash, n, l_contig.as<uint32_t>(),
                           l_end.as<uint32_t>(), l_gid.as<uint32_t>(), l_hash.as<uint64_t>());
        hipLaunchKernelGGL(spacing_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, l_contig.as<uint32_t>(), l_end.as<uint32_t>(),
                           d_n_live, min_spacing, b_kept.as<uint32_t>());
        SY_HIP(hipGetLastError());
    }
    exclusive_sum_u32(ctx, b_kept.as<uint32_t>(), b_kpos.as<uint32_t>(), n1);
    {
        ScopedKernelTimer t(ctx, "genome_filter");
        hipLaunchKernelGGL(emit_genomes_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, b_kept.as<uint32_t>(), b_kpos.as<uint32_t>(),
                           l_hash.as<uint64_t>(), d_n_live, pseudotax ? 1 : 0, d_out_k.as<uint64_t>(), d_out_t.as<uint64_t>());
        hipLaunchKernelGGL(genome_offsets_kernel, dim3(grid_for(n_genomes + 1)), dim3(256), 0, ctx->stream, l_gid.as<uint32_t>(),
                           b_kpos.as<uint32_t>(), d_n_live, n_genomes, d_koff.as<uint64_t>(), d_toff.as<uint64_t>());
        SY_HIP(hipGetLastError());
    }
    std::vector<uint64_t> h_toff(n_genomes + 1);
    ctx->d2h(kmer_off, d_koff.p, (n_genomes + 1) * 8);
    ctx->d2h(h_toff.data(), d_toff.p, (n_genomes + 1) * 8);
    const uint64_t nk = kmer_off[n_genomes], nt = pseudotax ? h_toff[n_genomes] : 0;
    SY_REQUIRE(nk <= n && nt <= n, "internal: inconsistent genome sketch sizes");
    uint64_t* hk = (uint64_t*)malloc(std::max<size_t>(1, nk) * 8);
    uint64_t* ht = out_tracked ? (uint64_t*)malloc(std::max<size_t>(1, nt) * 8) : nullptr;
    if (!hk || (out_tracked && !ht)) { free(hk); free(ht); throw std::bad_alloc(); }
    try {
        if (nk) ctx->d2h(hk, d_out_k.p, nk * 8);
        if (nt && ht) ctx->d2h(ht, d_out_t.p, nt * 8);
    } catch (...) { free(hk); free(ht); throw; }
    *out_kmers = hk;
    if (out_tracked) *out_tracked = ht;
    if (tracked_off && pseudotax) memcpy(tracked_off, h_toff.data(), (n_genomes + 1) * 8);
}

}  // namespace sylph

using namespace sylph;

extern "C" {

int sylph_sketch_genomes(sylph_ctx* ctx, const uint8_t* bases, const uint64_t* contig_off, uint64_t n_contigs,
                         const uint64_t* genome_contig_off, uint64_t n_genomes, uint32_t c, uint32_t k, int seed_mode,
                         uint64_t min_spacing, int pseudotax, int mem, uint64_t** out_kmers, uint64_t* kmer_off, uint64_t** out_tracked,
                         uint64_t* tracked_off) {
    return guarded([&] {
        SY_REQUIRE(ctx && out_kmers && kmer_off, "null argument");
        SY_REQUIRE((out_tracked == nullptr) == (tracked_off == nullptr), "out_tracked and tracked_off go together");
        sketch_genomes_impl(ctx, bases, contig_off, n_contigs, genome_contig_off, n_genomes, c, k, seed_mode, min_spacing, pseudotax,
                            mem, out_kmers, kmer_off, out_tracked, tracked_off);
    });
}

// sketch_genome, sketch.rs:550-622: the batch entry point with one genome.
int sylph_sketch_genome(sylph_ctx* ctx, const uint8_t* bases, const uint64_t* contig_off, uint64_t n_contigs, uint32_t c,
                        uint32_t k, int seed_mode, uint64_t min_spacing, int pseudotax, uint64_t** out_genome_kmers,
                        uint64_t* out_n, uint64_t** out_tracked, uint64_t* out_n_tracked) {
    return guarded([&] {
        SY_REQUIRE(ctx && out_genome_kmers && out_n, "null argument");
        const uint64_t goff[2] = {0, n_contigs};
        const uint64_t zero_off[1] = {0};
        uint64_t koff[2] = {0, 0}, toff[2] = {0, 0};
        uint64_t* tr = nullptr;
        sketch_genomes_impl(ctx, bases, n_contigs ? contig_off : zero_off, n_contigs, goff, 1, c, k, seed_mode, min_spacing, pseudotax,
                            SYLPH_MEM_HOST, out_genome_kmers, koff, &tr, toff);
        *out_n = koff[1];
        if (out_tracked) *out_tracked = tr; else free(tr);
        if (out_n_tracked) *out_n_tracked = toff[1];
    });
}

}  // extern "C"
