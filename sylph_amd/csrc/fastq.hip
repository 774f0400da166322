// fastq.hip — the records of plain four-line FASTQ text, found on the device.
//
// The reference reads a sample's files record by record with needletail (sketch.rs:775-815, :897-921: parse_fastx_file + next())
// and hands each record's sequence to the seeding; the host feed of this repository (host/feed.cpp) does the same job with all parse
// threads: index the newlines, gather the sequence lines, pack them, send 0.25 B per base.  On a box whose container may use 16
// CPUs that is 60-110 ms of host work per 1 Gbp sample against ~1 ms of GPU work.  Here the TEXT travels (2.1 B per base over
// PCIe, ~37 ms per Gbp) and the device does the rest:
//
//   fq_trim_kernel      trailing '\n' / '\r' bytes are not part of the text (needletail ignores blank space behind the last record)
//   fq_count_kernel     newlines per 4 KiB tile                      (one 16-byte load per lane, exact SWAR byte test)
//   exclusive scan      first line number of every tile
//   fq_lines_kernel     start of every line: line_start[L], L = 0 .. n_lines (the last entry is one past a virtual final newline)
//   fq_records_kernel   record r = lines 4r .. 4r+3: '@' and '+' in place, sequence and quality equally long (a '\r' in front of the
//                       newline is not part of a line); start + length of the sequence line, the bases of the text summed
//   fq_gather_kernel    a batch's sequences copied side by side (pairs interleaved: mate 1, mate 2) behind one offsets array — the
//                       (bases, rec_off) batch every seeding kernel of this library takes (SYLPH_MEM_DEVICE, SYLPH_ENC_ASCII)
//
// Anything that is not exactly that — a line count that is no multiple of four, a record without its '@' or '+', unequal lengths
// (multi-line FASTQ, FASTA, damage) — makes sylph_fastq_index return SYLPH_ERR_FORMAT: the caller then reads the file with its host
// reader, whose record and error semantics are needletail's (host/formats.cpp ChunkStream).  Nothing here guesses.
#include "common.h"
#include "device_common.h"
#include "sketch_session.h"
#include "partition.h"

struct sylph_fastq {
    sylph_ctx* ctx = nullptr;
    sylph::DevBuf text_own;               // the text when it came from the host (else borrowed: `text`)
    const uint8_t* text = nullptr;        // device pointer to byte 0
    uint64_t n = 0;                       // bytes without the trailing blank space
    sylph::DevBuf seq_start, seq_len;     // u64 / u32 per record
    uint64_t n_rec = 0, n_bases = 0;
    explicit sylph_fastq(sylph_ctx* c) : ctx(c), text_own(c), seq_start(c), seq_len(c) {}
};

namespace sylph {
namespace {

constexpr int FQ_TPB = 256, FQ_TILE = FQ_TPB * 16;       // one 16-byte load per lane
constexpr uint32_t FQ_MAX_TRAILING = 1u << 16;

// what the kernels leave for the host: [0] bytes of text without its trailing blank space, [1] first bad record (~0: none),
// [2] bases of the text, [3] flags (1: more than FQ_MAX_TRAILING blank bytes behind the text), [4] newlines, counted in 64 bits (the
// tiles' line numbers are a 32-bit scan: a text with 2^32 lines or more is refused, not mis-numbered)
struct FqWords { unsigned long long n_eff, bad_rec, n_bases, flags, n_nl; };

__global__ void fq_trim_kernel(const uint8_t* __restrict__ t, uint64_t n, FqWords* __restrict__ w) {
    uint64_t e = n;
    uint32_t steps = 0;
    while (e > 0 && (t[e - 1] == '\n' || t[e - 1] == '\r') && steps < FQ_MAX_TRAILING) { e--; steps++; }
    w->n_eff = e;
    w->bad_rec = ~0ull;
    w->n_bases = 0;
    w->n_nl = 0;
    w->flags = (e > 0 && (t[e - 1] == '\n' || t[e - 1] == '\r')) ? 1ull : 0ull;
}

// bit 7 of every byte of x that equals '\n' (exact: no borrow runs from one byte into the next)
__device__ __forceinline__ uint32_t newline_flags(uint32_t x) {
    x ^= 0x0A0A0A0Au;
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}

// the lane's 16 bytes of the aligned stream `al` (text = al + bias): newline flags of its four dwords, bytes outside [0, n_eff) cleared
__device__ __forceinline__ void lane_flags(const uint8_t* __restrict__ al, uint32_t bias, uint64_t n_eff, uint64_t tile, uint32_t f[4],
                                           int64_t& i0) {
    const uint64_t p = tile * FQ_TILE + (uint64_t)threadIdx.x * 16;       // position in the aligned stream
    i0 = (int64_t)p - (int64_t)bias;                                        // index of the lane's first byte in the text
    f[0] = f[1] = f[2] = f[3] = 0;
    if (i0 >= (int64_t)n_eff || i0 + 16 <= 0) return;
    const uint4 v = *reinterpret_cast<const uint4*>(al + p);
    f[0] = newline_flags(v.x); f[1] = newline_flags(v.y); f[2] = newline_flags(v.z); f[3] = newline_flags(v.w);
    if (i0 < 0 || i0 + 16 > (int64_t)n_eff) {                               // the text's first / last lane: byte by byte
#pragma unroll
        for (int d = 0; d < 4; d++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int64_t i = i0 + d * 4 + b;
                if (i < 0 || i >= (int64_t)n_eff) f[d] &= ~(0x80u << (8 * b));
            }
    }
}

__global__ __launch_bounds__(FQ_TPB) void fq_count_kernel(const uint8_t* __restrict__ al, uint32_t bias, FqWords* __restrict__ w,
                                                          uint32_t* __restrict__ tile_cnt) {
    __shared__ uint32_t s_wave[FQ_TPB / 64];
    uint32_t f[4];
    int64_t i0;
    lane_flags(al, bias, w->n_eff, blockIdx.x, f, i0);
    const uint32_t c = __popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]);
    uint32_t tot = 0;
    (void)block_excl_sum<FQ_TPB>(c, s_wave, &tot);
    if (threadIdx.x == 0) {
        tile_cnt[blockIdx.x] = tot;
        if (tot) atomicAdd(&w->n_nl, (unsigned long long)tot);
    }
}

__global__ __launch_bounds__(FQ_TPB) void fq_lines_kernel(const uint8_t* __restrict__ al, uint32_t bias, const FqWords* __restrict__ w,
                                                          const uint32_t* __restrict__ tile_base, uint64_t n_lines,
                                                          uint64_t* __restrict__ line_start) {
    __shared__ uint32_t s_wave[FQ_TPB / 64];
    uint32_t f[4];
    int64_t i0;
    const uint64_t n_eff = w->n_eff;
    lane_flags(al, bias, n_eff, blockIdx.x, f, i0);
    const uint32_t c = __popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]);
    uint64_t ord = (uint64_t)tile_base[blockIdx.x] + block_excl_sum<FQ_TPB>(c, s_wave, nullptr);   // newlines in front of this lane's bytes
    if (blockIdx.x == 0 && threadIdx.x == 0) { line_start[0] = 0; line_start[n_lines] = n_eff + 1; }
    if (!c) return;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t m = f[d];
        while (m) {
            const int b = (__ffs((int)m) - 1) >> 3;                         // lowest address first
            m &= ~(0x80u << (8 * b));
            ord++;
            if (ord < n_lines) line_start[ord] = (uint64_t)(i0 + d * 4 + b) + 1;
        }
    }
}

__global__ __launch_bounds__(256) void fq_records_kernel(const uint8_t* __restrict__ t, const uint64_t* __restrict__ line_start, uint64_t n_rec,
                                                         uint64_t* __restrict__ seq_start, uint32_t* __restrict__ seq_len, FqWords* __restrict__ w) {
    __shared__ unsigned long long s_sum[4];
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long len = 0;
    if (r < n_rec) {
        const uint64_t s0 = line_start[4 * r], s1 = line_start[4 * r + 1], s2 = line_start[4 * r + 2], s3 = line_start[4 * r + 3],
                       s4 = line_start[4 * r + 4];
        uint64_t sl = s2 - 1 - s1, ql = s4 - 1 - s3;
        if (sl && t[s1 + sl - 1] == '\r') sl--;
        if (ql && t[s3 + ql - 1] == '\r') ql--;
        const bool ok = t[s0] == '@' && t[s2] == '+' && sl == ql && sl <= 0xFFFFFFFEull;
        if (!ok) atomicMin(&w->bad_rec, (unsigned long long)r);
        seq_start[r] = s1;
        seq_len[r] = ok ? (uint32_t)sl : 0u;
        len = ok ? sl : 0ull;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) len += __shfl_xor(len, d);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&w->n_bases, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

// lengths of a batch's records in batch order (pairs interleaved), one sentinel 0 behind them for the scan; their 64-bit sum in *total
// (the scan is a 32-bit one: the caller checks the sum before it trusts the offsets)
__global__ void fq_batch_lens_kernel(const uint32_t* __restrict__ la, const uint32_t* __restrict__ lb, uint64_t first, uint64_t n_items,
                                     uint32_t* __restrict__ out, unsigned long long* __restrict__ total) {
    const uint64_t n = lb ? 2 * n_items : n_items;
    unsigned long long sum = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = j == n ? 0u : (lb ? ((j & 1) ? lb[first + (j >> 1)] : la[first + (j >> 1)]) : la[first + j]);
        out[j] = l;
        sum += l;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(total, sum);
}
__global__ void fq_widen_kernel(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) out[j] = in[j];
}
// one wavefront per record: its bytes from the text to their place in the batch
__global__ __launch_bounds__(256) void fq_gather_kernel(const uint8_t* __restrict__ ta, const uint64_t* __restrict__ sa, const uint8_t* __restrict__ tb,
                                                        const uint64_t* __restrict__ sb, uint64_t first, uint64_t n_rec_batch,
                                                        const uint32_t* __restrict__ off, uint8_t* __restrict__ bases) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t j = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); j < n_rec_batch; j += waves) {
        const uint32_t o = off[j], len = off[j + 1] - o;
        const uint8_t* src = tb ? ((j & 1) ? tb + sb[first + (j >> 1)] : ta + sa[first + (j >> 1)]) : ta + sa[first + j];
        for (uint32_t x = lane; x < len; x += 64) bases[o + x] = src[x];
    }
}

struct FormatError { std::string msg; };

uint32_t grid1(uint64_t n, uint32_t tpb = 256, uint32_t cap = 1u << 20) { return (uint32_t)std::min<uint64_t>(cap, std::max<uint64_t>(1, (n + tpb - 1) / tpb)); }

void fastq_index_impl(sylph_fastq* f, const void* text, uint64_t n_bytes, int mem) {
    sylph_ctx* ctx = f->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard dg(ctx->device);
    if (mem == SYLPH_MEM_DEVICE) {
        f->text = (const uint8_t*)text;
    } else {
        f->text_own.reserve(n_bytes + 64);
        if (mem == SYLPH_MEM_HOST_PINNED) SY_HIP(hipMemcpyAsync(f->text_own.p, text, n_bytes, hipMemcpyHostToDevice, ctx->stream));
        else ctx->h2d(f->text_own.p, text, n_bytes);
        f->text = f->text_own.as<uint8_t>();
    }
    const uint32_t bias = (uint32_t)((uintptr_t)f->text & 15);
    const uint8_t* al = f->text - bias;
    const uint64_t n_tiles = (n_bytes + bias + FQ_TILE - 1) / FQ_TILE;
    SY_REQUIRE(n_tiles < (1ull << 31), "sylph_fastq_index: text too large for one index (%llu bytes)", (unsigned long long)n_bytes);
    DevBuf& b_words = ctx->scratch[0];
    DevBuf& b_cnt = ctx->scratch[1];
    DevBuf& b_base = ctx->scratch[2];
    DevBuf& b_lines = ctx->scratch[3];
    b_words.reserve(sizeof(FqWords));
    b_cnt.reserve((n_tiles + 1) * 4);
    b_base.reserve((n_tiles + 1) * 4);
    FqWords* d_w = b_words.as<FqWords>();
    hipLaunchKernelGGL(fq_trim_kernel, dim3(1), dim3(1), 0, ctx->stream, f->text, n_bytes, d_w);
    SY_HIP(hipMemsetAsync(b_cnt.as<uint32_t>() + n_tiles, 0, 4, ctx->stream));
    hipLaunchKernelGGL(fq_count_kernel, dim3((uint32_t)n_tiles), dim3(FQ_TPB), 0, ctx->stream, al, bias, d_w, b_cnt.as<uint32_t>());
    SY_HIP(hipGetLastError());
    exclusive_sum_u32(ctx, b_cnt.as<uint32_t>(), b_base.as<uint32_t>(), n_tiles + 1);
    FqWords hw;
    uint32_t n_nl = 0;
    ctx->read_back(&n_nl, b_base.as<uint32_t>() + n_tiles, 4);     // (synchronises the stream)
    ctx->read_back(&hw, d_w, sizeof(hw));
    if (hw.flags & 1ull) throw FormatError{"more than 64 KiB of blank space behind the last line"};
    f->n = hw.n_eff;
    if (f->n == 0) throw FormatError{"no text"};
    if (hw.n_nl != (unsigned long long)n_nl) throw FormatError{std::to_string(hw.n_nl) + " lines: at most 2^32 - 1 per index"};
    const uint64_t n_lines = (uint64_t)n_nl + 1;                   // the last line is unterminated (its newline was trimmed)
    if (n_lines % 4) throw FormatError{"the number of lines (" + std::to_string(n_lines) + ") is not a multiple of four"};
    f->n_rec = n_lines / 4;
    b_lines.reserve((n_lines + 1) * 8);
    f->seq_start.reserve(f->n_rec * 8);
    f->seq_len.reserve(f->n_rec * 4 + 4);
    hipLaunchKernelGGL(fq_lines_kernel, dim3((uint32_t)n_tiles), dim3(FQ_TPB), 0, ctx->stream, al, bias, d_w, b_base.as<uint32_t>(), n_lines,
                       b_lines.as<uint64_t>());
    hipLaunchKernelGGL(fq_records_kernel, dim3(grid1(f->n_rec, 256, 1u << 30)), dim3(256), 0, ctx->stream, f->text, b_lines.as<uint64_t>(), f->n_rec,
                       f->seq_start.as<uint64_t>(), f->seq_len.as<uint32_t>(), d_w);
    SY_HIP(hipGetLastError());
    ctx->read_back(&hw, d_w, sizeof(hw));
    if (hw.bad_rec != ~0ull) throw FormatError{"record " + std::to_string(hw.bad_rec) + " is not a four-line record ('@' line, sequence, '+' line, quality of the sequence's length)"};
    f->n_bases = hw.n_bases;
}

}  // namespace
}  // namespace sylph

using namespace sylph;

extern "C" {

int sylph_fastq_index(sylph_ctx* ctx, const void* text, uint64_t n_bytes, int mem, sylph_fastq** out) {
    if (!ctx || !out || (!text && n_bytes)) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    if (mem != SYLPH_MEM_HOST && mem != SYLPH_MEM_DEVICE && mem != SYLPH_MEM_HOST_PINNED) { set_error("bad mem kind %d", mem); return SYLPH_ERR_INVALID; }
    *out = nullptr;
    if (n_bytes == 0) { set_error("sylph_fastq_index: no text"); return SYLPH_ERR_FORMAT; }
    ctx->refs.fetch_add(1);
    sylph_fastq* f = nullptr;
    int format = 0;
    const int rc = guarded([&] {
        f = new sylph_fastq(ctx);
        try { fastq_index_impl(f, text, n_bytes, mem); }
        catch (const FormatError& e) { set_error("sylph_fastq_index: not plain four-line FASTQ: %s", e.msg.c_str()); format = 1; }
    });
    if (rc != SYLPH_OK || format) {
        if (f) { std::lock_guard<std::mutex> lock(ctx->mu); delete f; }
        ctx_unref(ctx);
        return rc != SYLPH_OK ? rc : SYLPH_ERR_FORMAT;
    }
    *out = f;
    return SYLPH_OK;
}

int sylph_fastq_counts(const sylph_fastq* f, uint64_t* n_records, uint64_t* n_bases) {
    if (!f) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    if (n_records) *n_records = f->n_rec;
    if (n_bases) *n_bases = f->n_bases;
    return SYLPH_OK;
}

int sylph_fastq_lengths(sylph_fastq* f, uint64_t first, uint64_t n, uint32_t* out) {
    return guarded([&] {
        SY_REQUIRE(f && (out || n == 0), "null argument");
        SY_REQUIRE(first <= f->n_rec && n <= f->n_rec - first, "sylph_fastq_lengths: records [%llu, +%llu) of %llu", (unsigned long long)first,
                   (unsigned long long)n, (unsigned long long)f->n_rec);
        if (!n) return;
        std::lock_guard<std::mutex> lock(f->ctx->mu);
        DeviceGuard dg(f->ctx->device);
        f->ctx->d2h(out, f->seq_len.as<uint32_t>() + first, n * 4);
    });
}

void sylph_fastq_destroy(sylph_fastq* f) {
    if (!f) return;
    sylph_ctx* ctx = f->ctx;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard dg(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);       // kernels that read the text / the index may still be queued
        delete f;
    }
    ctx_unref(ctx);
}

int sylph_sketch_push_fastq(sylph_sketch* sk, sylph_fastq* a, sylph_fastq* b, uint64_t first, uint64_t n_items) {
    if (!sk || !a) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    uint64_t n_rec_batch = 0, n_bases = 0;
    const int rc = guarded([&] {
        sylph_ctx* ctx = sk->ctx;
        SY_REQUIRE(a->ctx == ctx && (!b || b->ctx == ctx), "sylph_sketch_push_fastq: the index lives on another context than the session");
        SY_REQUIRE((b != nullptr) == (sk->paired != 0), "sylph_sketch_push_fastq: a paired session takes two texts, a single-end session one");
        SY_REQUIRE(first <= a->n_rec && n_items <= a->n_rec - first && (!b || (first <= b->n_rec && n_items <= b->n_rec - first)),
                   "sylph_sketch_push_fastq: records [%llu, +%llu) are not all there", (unsigned long long)first, (unsigned long long)n_items);
        if (!n_items) return;
        n_rec_batch = b ? 2 * n_items : n_items;
        SY_REQUIRE(n_rec_batch < (1ull << 31), "sylph_sketch_push_fastq: at most 2^31 - 1 records per push");
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard dg(ctx->device);
        // a deferred batch of this session reads sk->fq_bases until its verdict is in: settle it before the buffer is written again
        resolve_deferred_slots(sk);
        DevBuf& b_len = ctx->scratch[0];
        DevBuf& b_off = ctx->scratch[1];
        DevBuf& b_tot = ctx->scratch[2];
        b_len.reserve((n_rec_batch + 1) * 4);
        b_off.reserve((n_rec_batch + 1) * 4);
        b_tot.reserve(8);
        SY_HIP(hipMemsetAsync(b_tot.p, 0, 8, ctx->stream));
        hipLaunchKernelGGL(fq_batch_lens_kernel, dim3(grid1(n_rec_batch + 1, 256, 4096)), dim3(256), 0, ctx->stream, a->seq_len.as<uint32_t>(),
                           b ? b->seq_len.as<uint32_t>() : nullptr, first, n_items, b_len.as<uint32_t>(), b_tot.as<unsigned long long>());
        SY_HIP(hipGetLastError());
        exclusive_sum_u32(ctx, b_len.as<uint32_t>(), b_off.as<uint32_t>(), n_rec_batch + 1);
        unsigned long long total = 0;
        ctx->read_back(&total, b_tot.p, 8);
        SY_REQUIRE(total < (1ull << 32) - 64, "sylph_sketch_push_fastq: %llu bases in one push (at most 2^32 - 65: push fewer records at a time)", total);
        n_bases = total;
        // no room for the gathered batch and its seeding?  Say so BEFORE anything of the session is touched (round 6): the caller
        // pushes fewer records at a time, or takes its host feed, which works through a sample in 256 Mbp batches
        if (n_bases + 64 > sk->fq_bases.cap) {
            size_t free_b = 0, total_b = 0;
            size_t pooled = 0;                                  // blocks the context's pool holds and hands out again
            for (const auto& blk : ctx->pool_free) pooled += blk.first;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + pooled < n_bases + n_bases / 2 + (256u << 20))
                throw HipError{hipErrorOutOfMemory, "sylph_sketch_push_fastq: no room for the batch", __FILE__, __LINE__};
        }
        sk->fq_bases.reserve(n_bases + 64);
        sk->fq_off.reserve((n_rec_batch + 1) * 8);
        hipLaunchKernelGGL(fq_widen_kernel, dim3(grid1(n_rec_batch + 1, 256, 4096)), dim3(256), 0, ctx->stream, b_off.as<uint32_t>(), n_rec_batch + 1,
                           sk->fq_off.as<uint64_t>());
        hipLaunchKernelGGL(fq_gather_kernel, dim3(grid1(n_rec_batch, 4, 1u << 16)), dim3(256), 0, ctx->stream, a->text, a->seq_start.as<uint64_t>(),
                           b ? b->text : nullptr, b ? b->seq_start.as<uint64_t>() : nullptr, first, n_rec_batch, b_off.as<uint32_t>(),
                           sk->fq_bases.as<uint8_t>());
        SY_HIP(hipMemsetAsync(sk->fq_bases.as<uint8_t>() + n_bases, 0, 64, ctx->stream));
        SY_HIP(hipGetLastError());
    });
    if (rc != SYLPH_OK || !n_rec_batch) return rc;
    return sylph_sketch_push_enc(sk, sk->fq_bases.as<uint8_t>(), sk->fq_off.as<uint64_t>(), n_rec_batch, n_bases, SYLPH_MEM_DEVICE, SYLPH_ENC_ASCII);
}

}  // extern "C"
