// inflate.hip — gzip (.fastq.gz / .fasta.gz, single-member, multi-member and BGZF) inflated ON THE DEVICE.
//
// The reference reads its normal input — gzip files (README.md:51; sketch.rs:780-781, :906 through needletail -> flate2) — with one
// thread per file: ~0.4 Gbp/s.  Rounds 4-5 inflated on the host with all parse threads (host/pgunzip.cpp): 1.6-2.0 Gbp/s per
// command on the GPU box, the slowest road of the feed although compressed bytes (0.4 B per base) are the cheapest thing to move
// over PCIe.  Here the COMPRESSED bytes travel and the device does the rest (bookkeeping and the proof of it: inflate_plan.h):
//
//   scan1_kernel      every bit position of the file: "BTYPE = dynamic, HLIT/HDIST in range, the code-length code is complete"
//                     (one 17-bit test + a Kraft sum over the 3-bit lengths from a 512-entry LDS table) -> ~0.1 % survive
//   scan2_kernel      one lane per survivor: the header's 258-316 code lengths decoded through a per-lane 7-bit table; both codes
//                     complete (or zlib's single-code exception), an end-of-block code -> the CANDIDATES (true block starts + ~0)
//   decode_kernel     one WAVEFRONT per candidate: the block's Huffman tables built in LDS by the 64 lanes (canonical arithmetic,
//                     10-bit / 8-bit roots, codes beyond the root decoded by first-code comparison — no second-level tables:
//                     3.4 KB of LDS per wave), symbols decoded wave-uniformly (bit buffer and state live in SGPRs; the compressed
//                     words in two VGPRs read with v_readlane; one VGPR as a 64-entry table of 1-3 literal runs), output through
//                     an LDS ring of the last 4 Ki cells: matches are copied by all lanes, from the ring where they reach.  Output:
//                     16-bit cells — a byte, or 256 + w = "byte w of the 32 KiB in front of this block", which nobody knows yet.
//                     The wave runs on through stored / fixed blocks and stops in front of the next dynamic header or behind
//                     a final block, and reports where.
//   (host)            chain_walk: which candidates are the stream, where their bytes go, the members' trailers
//   winfn_kernel      per group of ~sqrt(K) chain blocks, sequentially inside the group: block k's window as a FUNCTION of the
//                     group's first window (32 Ki cells: a byte or a reference into that window), both in LDS (128 KB of the 160)
//   winchain_kernel   the groups' first windows, one after the other (one workgroup, 32 KiB look-ups per group)
//   translate_kernel  all blocks side by side: the block's real window composed in LDS, cells -> bytes in their place in the text
//   crc_kernel        CRC-32 of the text: 64 bytes per lane through slicing-by-4 tables in LDS, the lanes' registers shifted to their
//                     place by polynomial arithmetic (crc32 is linear over GF(2)) and XORed into the member's word; the host
//                     compares with the trailers
//
// Any doubt — a chain that breaks, an output that does not fit its region, a CRC or ISIZE that differs, no room on the device —
// returns SYLPH_ERR_FORMAT / SYLPH_ERR_NOMEM and nothing else happens: the caller inflates with zlib as before.
#include <zlib.h>

#include <algorithm>

#include "common.h"
#include "inflate_plan.h"

struct sylph_inflated {
    sylph_ctx* ctx = nullptr;
    void* buf = nullptr;                  // hipMalloc'ed: 256 bytes of zero padding, the text, 256 bytes of zero padding
    uint64_t n = 0;
    uint64_t n_members = 0, n_blocks = 0, n_candidates = 0, n_host_members = 0, n_redone = 0;
    std::vector<std::pair<uint64_t, uint64_t>> files;   // [begin, end) of every file's text in the whole text
    const uint8_t* text() const { return (const uint8_t*)buf + 256; }
};

namespace sylph {
namespace {

using namespace inflate_plan;

struct FormatDecline { std::string msg; };

constexpr int LIT_ROOT = 10, DIST_ROOT = 8, PRE_ROOT = 7;
constexpr uint32_t T_LONG = 0xFFFFu;
constexpr uint32_t REGION_RATIO = 16;       // cells a candidate may write per compressed byte up to the next candidate ...
constexpr uint32_t REGION_SLACK = 1024;     // ... plus this
constexpr uint64_t GZ_PAD_WORDS = 1024;     // zero words behind the compressed bytes (a header parsed at the very end reads < 600 bytes on)

// =================================================================================================================================
// scan, stage 1: the cheap test at every bit position
// =================================================================================================================================
constexpr int SCAN_TPB = 256;
constexpr uint32_t SCAN_STAGE = 4096;       // staged hits per workgroup before a flush (>= SCAN_TPB * 8 + what a flush leaves)

__global__ __launch_bounds__(SCAN_TPB) void scan1_kernel(const uint32_t* __restrict__ gzw, uint64_t byte_lo, uint64_t byte_hi, uint64_t* __restrict__ out,
                                                         unsigned long long* __restrict__ n_out, uint64_t cap) {
    __shared__ uint8_t kraft[512];
    __shared__ uint64_t stage[SCAN_STAGE];
    __shared__ uint32_t stage_n;
    __shared__ unsigned long long gbase;
    for (uint32_t i = threadIdx.x; i < 512; i += SCAN_TPB) {
        uint32_t s = 0;
        for (int f = 0; f < 3; f++) { const uint32_t v = (i >> (3 * f)) & 7; if (v) s += 128u >> v; }
        kraft[i] = (uint8_t)s;                                                // <= 192
    }
    if (threadIdx.x == 0) stage_n = 0;
    __syncthreads();
    const uint64_t span = byte_hi - byte_lo;
    for (uint64_t base = (uint64_t)blockIdx.x * SCAN_TPB; base < span; base += (uint64_t)gridDim.x * SCAN_TPB) {
        const uint64_t b = byte_lo + base + threadIdx.x;
        if (b < byte_hi) {
            const uint64_t wi = b >> 2;
            const uint32_t d0 = gzw[wi], d1 = gzw[wi + 1], d2 = gzw[wi + 2], d3 = gzw[wi + 3];
            const uint32_t s0 = (uint32_t)(b & 3) * 8;
#pragma unroll
            for (uint32_t sh = 0; sh < 8; sh++) {
                const uint32_t s = s0 + sh;                                   // 0 .. 31
                const uint32_t h0 = __builtin_amdgcn_alignbit(d1, d0, s);
                // BTYPE == 2 (bits 1-2 = 0,1), HLIT <= 29, HDIST <= 29
                if (((h0 >> 1) & 3) != 2 || ((h0 >> 3) & 31) > 29 || ((h0 >> 8) & 31) > 29) continue;
                const uint32_t h1 = __builtin_amdgcn_alignbit(d2, d1, s), h2 = __builtin_amdgcn_alignbit(d3, d2, s);
                const uint32_t hclen = ((h0 >> 13) & 15) + 4;
                uint64_t c = ((((uint64_t)h1 << 32) | h0) >> 17) | ((uint64_t)h2 << 47);
                c &= (1ull << (3 * hclen)) - 1;
                uint32_t sum = 0;
#pragma unroll
                for (int j = 0; j < 7; j++) sum += kraft[(uint32_t)(c >> (9 * j)) & 511];
                if (sum != 128) continue;
                const uint32_t at = atomicAdd(&stage_n, 1u);
                stage[at] = b * 8 + sh;
            }
        }
        __syncthreads();
        const uint32_t m = stage_n;                                           // (uniform: everybody reads the same word between two barriers)
        __syncthreads();
        if (m > SCAN_STAGE - SCAN_TPB * 8) {
            if (threadIdx.x == 0) { gbase = atomicAdd(n_out, (unsigned long long)m); stage_n = 0; }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < m; i += SCAN_TPB)
                if (gbase + i < cap) out[gbase + i] = stage[i];
            __syncthreads();
        }
    }
    __syncthreads();
    const uint32_t m = stage_n;
    if (threadIdx.x == 0 && m) gbase = atomicAdd(n_out, (unsigned long long)m);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += SCAN_TPB)
        if (gbase + i < cap) out[gbase + i] = stage[i];
}

// =================================================================================================================================
// scan, stage 2: one lane per survivor, the whole header
// =================================================================================================================================
struct LaneBits {                                  // per-lane bit reader over the dword stream
    const uint32_t* w;
    uint64_t acc, wpos;
    uint32_t cnt;
    __device__ __forceinline__ void init(const uint32_t* words, uint64_t bitpos) {
        w = words;
        wpos = bitpos >> 5;
        const uint32_t s = (uint32_t)bitpos & 31;
        acc = (uint64_t)(w[wpos++] >> s);
        cnt = 32 - s;
        refill();
    }
    __device__ __forceinline__ void refill() { if (cnt <= 32) { acc |= (uint64_t)w[wpos++] << cnt; cnt += 32; } }
    __device__ __forceinline__ uint32_t take(uint32_t k) { const uint32_t v = (uint32_t)acc & ((1u << k) - 1); acc >>= k; cnt -= k; return v; }
    __device__ __forceinline__ uint64_t bitpos() const { return wpos * 32 - cnt; }
};

__device__ __constant__ uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

constexpr int SCAN2_TPB = 128;
__global__ __launch_bounds__(SCAN2_TPB) void scan2_kernel(const uint32_t* __restrict__ gzw, uint64_t n_bits, const uint64_t* __restrict__ in,
                                                          const unsigned long long* __restrict__ n_in_p, uint64_t in_cap, uint64_t* __restrict__ out,
                                                          unsigned long long* __restrict__ n_out, uint64_t cap) {
    __shared__ uint8_t pre[SCAN2_TPB][128 + 4];    // (+4: the lanes' tables start in different banks)
    const uint64_t n_in = (uint64_t)min((unsigned long long)*n_in_p, (unsigned long long)in_cap);
    const uint64_t i = (uint64_t)blockIdx.x * SCAN2_TPB + threadIdx.x;
    if (i >= n_in) return;
    const uint64_t p = in[i];
    LaneBits b;
    b.init(gzw, p);
    b.take(3);
    const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
    uint64_t clp = 0;                               // the 19 code lengths of the code-length code, 3 bits each, by symbol
    for (uint32_t j = 0; j < hclen; j++) { b.refill(); clp |= (uint64_t)b.take(3) << (3 * CL_ORDER[j]); }
    // canonical codes of the code-length code (complete: stage 1 checked the Kraft sum) into the lane's 128-entry table
    uint8_t* tab = pre[threadIdx.x];
    {
        uint32_t code = 0;
        for (uint32_t l = 1; l <= 7; l++) {
            for (uint32_t s = 0; s < 19; s++) {
                if (((clp >> (3 * s)) & 7) != l) continue;
                const uint32_t rev = __brev(code) >> (32 - l);
                for (uint32_t k = rev; k < 128; k += 1u << l) tab[k] = (uint8_t)(s << 3 | l);
                code++;
            }
            code <<= 1;
        }
    }
    const uint32_t total = hlit + hdist;
    uint32_t idx = 0, prev = 0, kl = 0, kd = 0, nzl = 0, nzd = 0, l1 = 0, d1 = 0, eob = 0;
    bool ok = true;
    while (idx < total) {
        b.refill();
        const uint32_t e = tab[(uint32_t)b.acc & 127];
        b.take(e & 7);
        const uint32_t sym = e >> 3;
        uint32_t val, rep;
        if (sym < 16) { val = sym; rep = 1; }
        else if (sym == 16) { if (idx == 0) { ok = false; break; } val = prev; rep = 3 + b.take(2); }
        else if (sym == 17) { val = 0; rep = 3 + b.take(3); }
        else { val = 0; rep = 11 + b.take(7); }
        if (idx + rep > total) { ok = false; break; }
        if (val) {
            const uint32_t nl = idx >= hlit ? 0 : min(rep, hlit - idx), nd = rep - nl;
            kl += nl << (15 - val); nzl += nl; if (val == 1) l1 += nl;
            kd += nd << (15 - val); nzd += nd; if (val == 1) d1 += nd;
            if (idx <= 256 && 256 < idx + rep) eob = val;
        }
        prev = val;
        idx += rep;
    }
    if (!ok || b.bitpos() > n_bits || eob == 0) return;
    if (kl > 32768 || (kl < 32768 && !(nzl == 1 && l1 == 1))) return;          // zlib inflate_table: incomplete only for a single 1-bit code
    if (nzd && (kd > 32768 || (kd < 32768 && !(nzd == 1 && d1 == 1)))) return;
    const unsigned long long at = atomicAdd(n_out, 1ull);
    if (at < cap) out[at] = p;
}

// =================================================================================================================================
// decode: one wavefront per candidate
// =================================================================================================================================
constexpr uint32_t FLUSH = 1024;     // cells of a block leaving the LDS ring for global memory at a time (DecodeOut)
// RING: cells of a block kept in LDS — 4096, or 2048 when the file has more blocks than wavefronts fit beside each other with the larger ring
template <uint32_t RING>
struct WaveTables {
    uint16_t lit[1 << LIT_ROOT];          // sym << 4 | len;  0 = no code;  T_LONG = longer than the root
    uint16_t dist[1 << DIST_ROOT];        // (the code-length code is built here first, 7-bit root)
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t lit_first[16], lit_count[16], lit_off[16];
    uint16_t dist_first[16], dist_count[16], dist_off[16];
    uint8_t lens[320 + 64];
    uint8_t plens[20 + 64];
    uint32_t lit_maxlen, dist_maxlen;
    alignas(16) uint16_t ring[RING + 64];   // (+64: where the lanes that have nothing to write write — see DUMMY below)
    uint32_t stream[256];
};

// Everything the decoding wavefront keeps — bit buffer, positions, counts, the symbol just decoded — is the same in all 64 lanes, and it
// has to live in SGPRs for the loop to run on the scalar unit: a value that came out of LDS or global memory is "divergent" to the
// compiler until it has been through v_readfirstlane, and one divergent value in a loop condition drags the whole state into VGPRs
// behind exec masks (the first build of this file: 7,900 lines of s_and_saveexec).
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

struct WaveBits {                                  // the same reader, but everything in it is wave-uniform (SGPRs)
    const uint32_t* w;
    uint64_t acc, wpos, wlimit, cbase;
    uint32_t cnt;
    bool over;
    // The stream's words [cbase - 64, cbase + 128) lie in an LDS buffer of 256 words (word i at [i & 255]); the 64 words behind them wait
    // in a VGPR, loaded one chunk — 256 compressed bytes, one coalesced load — ahead of their use.  Both readers take their words there:
    // this one a word per refill (a scalar load per refill put a trip to memory, ~1-2 thousand cycles with nothing else to do, into the
    // chain of every ten literals), the window step three words per lane.
    uint32_t nxt;
    uint32_t* stream;
    uint64_t at;                                   // !synced: the position (acc / cnt / wpos are stale: the window step moves `at` alone)
    bool synced;
    __device__ __forceinline__ uint64_t position() const { return synced ? wpos * 32 - cnt : at; }
    __device__ __forceinline__ void move_to(uint64_t bitpos) {            // (the window step: no word is fetched for the scalar reader)
        at = bitpos;
        synced = false;
        if ((long long)((bitpos >> 5) - cbase) >= 64) next_chunk();
    }
    __device__ __forceinline__ void sync() { if (!synced) seek(at); }
    __device__ __forceinline__ void init(const uint32_t* words, uint64_t bitpos, uint64_t limit_words) {
        const uint32_t lane = threadIdx.x & 63;
        w = words;
        wlimit = limit_words;
        over = false;
        wpos = bitpos >> 5;
        cbase = wpos & ~(uint64_t)63;
        stream[(cbase + lane) & 255] = w[cbase + lane];
        stream[(cbase + 64 + lane) & 255] = w[cbase + 64 + lane];
        nxt = w[cbase + 128 + lane];
        const uint32_t s = (uint32_t)bitpos & 31;
        acc = (uint64_t)(word() >> s);
        cnt = 32 - s;
        synced = true;
        at = 0;
        refill();
    }
    __device__ __forceinline__ void next_chunk() {
        const uint32_t lane = threadIdx.x & 63;
        stream[(cbase + 128 + lane) & 255] = nxt;
        cbase += 64;
        nxt = w[cbase + 128 + lane];
    }
    __device__ __forceinline__ uint32_t word() {                          // the stream's word wpos; wpos moves on
        const uint32_t v = uni(stream[wpos & 255]);
        wpos++;
        if ((long long)(wpos - cbase) >= 64) next_chunk();
        return v;
    }
    // continue at another bit position: at most two words behind the words already taken, less than a chunk ahead of them
    __device__ __forceinline__ void seek(uint64_t bitpos) {
        wpos = bitpos >> 5;
        if ((long long)(wpos - cbase) >= 64) next_chunk();
        const uint32_t s = (uint32_t)bitpos & 31;
        acc = (uint64_t)(word() >> s);
        cnt = 32 - s;
        synced = true;
        refill();
    }
    __device__ __forceinline__ void refill() {
        if (cnt <= 32) {
            if (wpos >= wlimit) { over = true; wpos++; }
            else acc |= (uint64_t)word() << cnt;
            cnt += 32;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t k) const { return (uint32_t)acc & ((1u << k) - 1); }
    __device__ __forceinline__ void drop(uint32_t k) { acc >>= k; cnt -= k; }
    __device__ __forceinline__ uint32_t take(uint32_t k) { const uint32_t v = peek(k); drop(k); return v; }
    __device__ __forceinline__ uint64_t bitpos() const { return wpos * 32 - cnt; }
};

__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1; }

// Canonical code from lens[0, nsym): sorted symbols, first code / count / offset per length, the root table.  kind: 0 = the
// code-length code (must be complete), 1 = literal/length, 2 = distance (zlib inflate_table: an incomplete code is fine only when
// it is a single 1-bit code; a distance code with no code at all is fine).  Wave-uniform result.
__device__ bool build_code_raw(const uint8_t* lens, uint32_t nsym, uint32_t root, uint16_t* tab, uint16_t* sorted, uint16_t* first, uint16_t* count,
                           uint16_t* off, uint32_t* maxlen_out, int kind) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t used = 0;
    // (every loop of this kernel counts in wave-uniform steps and tests the lane inside: a loop whose induction variable depends on the
    //  lane has, for the compiler, an exit that lanes may take at different times — and everything behind it stops being uniform)
    for (uint32_t base = 0; base < nsym; base += 64)
        if (base + lane < nsym) used |= 1u << lens[base + lane];
    used = uni(wave_or(used)) & ~1u;
    uint32_t run = 0, code = 0, prev_count = 0, maxlen = 0;
    int left = 1;
    bool over = false;
    for (uint32_t L = 1; L <= 15; L++) {
        const uint32_t off_l = run;
        if (used >> L & 1) {
            for (uint32_t base = 0; base < nsym; base += 64) {
                const uint32_t s = base + lane;
                const bool hit = s < nsym && lens[s] == L;
                const uint64_t mask = __ballot(hit);
                if (hit) sorted[run + __popcll(mask & lanemask_lt())] = (uint16_t)s;
                run += __popcll(mask);
            }
            maxlen = L;
        }
        const uint32_t c = run - off_l;
        code = (code + prev_count) << 1;
        prev_count = c;
        left = (left << 1) - (int)c;
        if (left < 0) over = true;
        if (lane == 0) { first[L] = (uint16_t)code; count[L] = (uint16_t)c; off[L] = (uint16_t)off_l; }
    }
    *maxlen_out = maxlen;
    __syncthreads();
    if (over) return false;
    if (run == 0) { if (kind != 2) return false; }
    else if (left > 0 && (kind == 0 || maxlen != 1)) return false;
    const uint32_t top = min(root, maxlen);
    for (uint32_t base = 0; base < (1u << root); base += 64) {
        const uint32_t idx = base + lane;
        const uint32_t r = __brev(idx) >> (32 - root);
        uint32_t e = maxlen > root ? T_LONG : 0;
        for (uint32_t L = 1; L <= top; L++) {
            const uint32_t d = (r >> (root - L)) - first[L];
            if (d < count[L]) { e = (uint32_t)sorted[off[L] + d] << 4 | L; break; }
        }
        tab[idx] = (uint16_t)e;
    }
    __syncthreads();
    return true;
}

// (the value a call returns is divergent to the compiler, whatever the callee computes: one v_readfirstlane says what it is)
__device__ __forceinline__ bool build_code(const uint8_t* lens, uint32_t nsym, uint32_t root, uint16_t* tab, uint16_t* sorted, uint16_t* first, uint16_t* count,
                                           uint16_t* off, uint32_t* maxlen_out, int kind) {
    return uni(build_code_raw(lens, nsym, root, tab, sorted, first, count, off, maxlen_out, kind) ? 1u : 0u) != 0;
}

// a code longer than the root: first-code comparison, lengths root+1 .. maxlen.  -> sym << 4 | len, or 0
__device__ __forceinline__ uint32_t decode_long(uint32_t bits, uint32_t root, uint32_t maxlen, const uint16_t* sorted, const uint16_t* first,
                                                const uint16_t* count, const uint16_t* off) {
    const uint32_t r = __brev(bits);
    for (uint32_t L = root + 1; L <= maxlen; L++) {
        const uint32_t d = (r >> (32 - L)) - uni(first[L]);
        if (d < uni(count[L])) return uni(sorted[uni(off[L]) + d]) << 4 | L;
    }
    return 0;
}

// DUMMY: a store that only some lanes make is written as a store every lane makes, the idle ones into a pad behind the array — an
// `if (lane < c)` is a divergent branch, and LLVM's uniformity analysis marks every phi of the block where its two sides meet as
// divergent; when that block is also where a loop's `continue` / `break` paths meet, the loop's exit stops being uniform and all of its
// state moves into VGPRs behind exec masks (found with opt -passes='print<uniformity>').
// Where a block's cells go: the last RING of them live in an LDS ring, and whole chunks of FLUSH cells leave for the candidate's
// region in global memory behind the wave (two 16-byte stores per lane).  A copy whose source lies inside the ring — distance +
// length <= RING — is LDS reads and writes, ~100 cycles; only a copy from further back loads from global memory (its source left
// the ring at least a chunk ago, so the stores are issued; the same wave's loads see them).  The first version kept nothing on the
// chip and every match waited for the acknowledgement of all earlier stores and for its own load — vmcnt counts both —, ~5,000
// cycles per match: 35 ms for a block of 16 K matches, whatever the literals cost.
// one candidate for a wavefront: where its block starts, where its cells go (offset into the cells buffer) and how many fit
struct DecodeJob { unsigned long long start_bit, region; uint32_t cap, have_window; };
// the spill arena behind the candidates' regions: SPILL_SLOTS regions of SPILL_CELLS cells, handed out by an atomic counter (next)
struct Spill { unsigned long long base; uint32_t slots; };
constexpr uint32_t SPILL_CELLS = 1u << 20, SPILL_SLOTS = 96;

template <uint32_t RING>
struct DecodeOut {                       // n, flushed, cap: wave-uniform
    static constexpr uint32_t RING_MASK = RING - 1;
    uint16_t* out;
    uint16_t* ring;
    uint32_t n, flushed, cap;
    bool dry;                            // the region is full: the rest of the block is decoded for its length and its end only (room())
    __device__ __forceinline__ void flush_chunk() {
        if (!dry) {
            const uint32_t lane = threadIdx.x & 63;
            const uint4* src = reinterpret_cast<const uint4*>(ring + ((flushed + lane * 16) & RING_MASK));
            uint4* dst = reinterpret_cast<uint4*>(out + region + flushed + lane * 16);
            const uint4 a = src[0], b = src[1];
            dst[0] = a;
            dst[1] = b;
        }
        flushed += FLUSH;
    }
    // k more cells: false = not even countable (a block of 4 Gi cells).  A block that outgrows its region — a header-like bit pattern
    // inside it cut the region short (one in ~4,000 blocks of a FASTQ file), or it deflates better than REGION_RATIO — MOVES, once, to a
    // region of SPILL_CELLS cells of the spill arena (what it has written so far goes with it: whole chunks, 16 bytes per lane and
    // step); a block that outgrows that too, or finds the arena empty, goes on DRY: it still reports where it ends and how long it is,
    // and the host has it decoded once more into a region of exactly that size.
    uint32_t* spill_next;
    unsigned long long spill_base;
    uint32_t spill_slots;
    bool moved;
    unsigned long long region;
    __device__ __forceinline__ bool room(uint32_t k) {
        if (n + k > cap && !dry) {
            uint32_t slot = ~0u;
            if (!moved && n + k <= SPILL_CELLS) {
                if ((threadIdx.x & 63) == 0) slot = atomicAdd(spill_next, 1u);
                slot = uni(slot);
            }
            if (slot < spill_slots) {
                const unsigned long long to = spill_base + (unsigned long long)slot * SPILL_CELLS;
                for (uint32_t at = 0; at < flushed; at += 64 * 8) {
                    const uint32_t i = at + (threadIdx.x & 63) * 8;
                    if (i < flushed) *reinterpret_cast<uint4*>(out + to + i) = *reinterpret_cast<const uint4*>(out + region + i);
                }
                region = to;
                cap = SPILL_CELLS;
                moved = true;
            } else dry = true;
        }
        return n + k <= 0xFFFFFF00u;
    }
    __device__ __forceinline__ void advance(uint32_t k) {
        n += k;
        while (n - flushed >= FLUSH) flush_chunk();
    }
    __device__ __forceinline__ void finish() {
        for (uint32_t base = flushed; base < n && !dry; base += 64) {
            const uint32_t i = base + (threadIdx.x & 63);
            if (i < n) out[region + i] = ring[i & RING_MASK];
        }
        flushed = n;
    }
};

template <uint32_t RING>
__global__ __launch_bounds__(64) void decode_kernel(const uint32_t* __restrict__ gzw, uint64_t n_words, const DecodeJob* __restrict__ jobs, uint16_t* __restrict__ cells,
                                                    Spill spill, uint32_t* __restrict__ spill_next, BlockResult* __restrict__ res) {
    constexpr uint32_t RING_MASK = RING - 1;
    __shared__ WaveTables<RING> T;
    const uint32_t k = blockIdx.x, lane = threadIdx.x;
    const DecodeJob job = jobs[k];
    const uint64_t start = job.start_bit;
    const uint64_t region = job.region;
    DecodeOut<RING> o;
    o.out = cells;                                  // (the whole buffer: a block's cells begin at o.region — which the block may change, room())
    o.ring = T.ring;
    o.n = 0; o.flushed = 0;
    o.cap = job.cap;
    o.dry = false;
    o.spill_base = spill.base;
    o.spill_slots = spill.slots;
    o.spill_next = spill_next;
    o.moved = false;
    o.region = region;
    const bool have_window = job.have_window != 0;
    WaveBits b;
    b.stream = T.stream;
    b.init(gzw, start, n_words);
    uint32_t status = ST_NONE, flags = 0;
    uint64_t end_bit = 0;
    bool first_block = true;
    uint32_t n_sym = 0, n_far = 0;
    uint64_t t_build = 0;

    const uint64_t t_start = __builtin_readcyclecounter();
    for (;;) {
        if (status != ST_NONE) break;
        b.refill();
        const uint64_t header_at = b.bitpos();
        const uint32_t bfinal = b.peek(1), btype = (b.peek(3) >> 1);
        if (!first_block && btype == 2) { status = ST_NEXT_DYNAMIC; end_bit = header_at; break; }
        first_block = false;
        b.drop(3);
        if (btype == 3) { status = ST_ERR_CODE; break; }
        if (btype == 0) {
            b.drop(b.cnt & 7);                                                 // to the byte boundary
            b.refill();
            const uint32_t len = b.take(16);
            b.refill();
            const uint32_t nlen = b.take(16);
            if (b.over || (len ^ 0xFFFFu) != nlen) { status = ST_ERR_STORED; break; }
            const uint64_t from = b.bitpos() >> 3;
            if (from + len > n_words * 4) { status = ST_ERR_OVERRUN; break; }
            if (!o.room(len)) { status = ST_OVERFLOW; break; }
            const uint8_t* gzb = reinterpret_cast<const uint8_t*>(gzw);
            for (uint32_t base = 0; base < len; base += 64) {
                o.ring[base + lane < len ? (o.n + lane) & RING_MASK : RING + lane] = gzb[from + min(base + lane, len - 1)];
                o.advance(min(64u, len - base));
            }
            b.init(gzw, (from + len) * 8, n_words);
        } else {
            const uint64_t t_b0 = __builtin_readcyclecounter();
            if (btype == 2) {
                b.refill();
                const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
                if (hlit > 286 || hdist > 30) { status = ST_ERR_CODE; break; }
                T.plens[lane < 19 ? lane : 20 + lane] = 0;   // (no branch on the lane: DUMMY)
                __syncthreads();
                for (uint32_t i = 0; i < hclen; i++) {
                    b.refill();
                    const uint32_t v = b.take(3);
                    T.plens[lane == 0 ? (uint32_t)CL_ORDER[i] : 20 + lane] = (uint8_t)v;
                }
                __syncthreads();
                uint32_t pre_max;
                if (b.over || !build_code(T.plens, 19, PRE_ROOT, T.dist, T.dist_sorted, T.dist_first, T.dist_count, T.dist_off, &pre_max, 0)) {
                    status = b.over ? ST_ERR_OVERRUN : ST_ERR_CODE;
                    break;
                }
                const uint32_t total = hlit + hdist;
                uint32_t i = 0, prev = 0;
                bool bad = false;
                while (i < total) {
                    b.refill();
                    const uint32_t e = uni(T.dist[b.peek(PRE_ROOT)]);
                    if (e == 0) { bad = true; break; }
                    b.drop(e & 15);
                    const uint32_t sym = e >> 4;
                    uint32_t val, rep;
                    if (sym < 16) { val = sym; rep = 1; }
                    else if (sym == 16) { if (i == 0) { bad = true; break; } val = prev; rep = 3 + b.take(2); }
                    else if (sym == 17) { val = 0; rep = 3 + b.take(3); }
                    else { val = 0; rep = 11 + b.take(7); }
                    if (i + rep > total) { bad = true; break; }
                    for (uint32_t jb = 0; jb < rep; jb += 64) T.lens[jb + lane < rep ? i + jb + lane : 320 + lane] = (uint8_t)val;
                    prev = val;
                    i += rep;
                }
                __syncthreads();
                if (bad || b.over || uni(T.lens[256]) == 0) { status = b.over ? ST_ERR_OVERRUN : ST_ERR_CODE; break; }
                if (!build_code(T.lens, hlit, LIT_ROOT, T.lit, T.lit_sorted, T.lit_first, T.lit_count, T.lit_off, &T.lit_maxlen, 1) ||
                    !build_code(T.lens + hlit, hdist, DIST_ROOT, T.dist, T.dist_sorted, T.dist_first, T.dist_count, T.dist_off, &T.dist_maxlen, 2)) {
                    status = ST_ERR_CODE;
                    break;
                }
            } else {
                for (uint32_t sb = 0; sb < 288; sb += 64) {
                    const uint32_t s = sb + lane;
                    if (s < 288) T.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                }
                if (lane < 32) T.lens[288 + lane] = 5;
                __syncthreads();
                build_code(T.lens, 288, LIT_ROOT, T.lit, T.lit_sorted, T.lit_first, T.lit_count, T.lit_off, &T.lit_maxlen, 1);
                build_code(T.lens + 288, 32, DIST_ROOT, T.dist, T.dist_sorted, T.dist_first, T.dist_count, T.dist_off, &T.dist_maxlen, 2);
            }
            const uint32_t lit_max = uni(T.lit_maxlen), dist_max = uni(T.dist_maxlen);
            t_build += __builtin_readcyclecounter() - t_b0;
            // ---- the block's symbols
            // THE WINDOW STEP.  A wavefront that decodes one symbol at a time spends 300 cycles on a literal and 1,500 on a match, whatever
            // the instruction count: every turn is a chain of dependent scalar instructions, taken branches and trips to the LDS, each
            // waited for with nothing else to do (profiles/r06_inflate_steps.txt).  So the 64 lanes decode the 64 BIT POSITIONS behind the
            // stream's position at once — lane j: "if a symbol started j bits from here, which one, how long, how many bits" (its own three
            // words of the stream, one look-up in each table) — and the wave then walks the chain of true starts with v_readlane (bits used
            // by the symbol at 0 lead to the next start, ...), four scalar instructions per symbol.  The cells the walked symbols produce
            // (at most 64 per step) are dealt to the lanes, literals written, copies resolved in rounds — a round gives every cell whose
            // source lies in front of the first unresolved one its value: one round unless a copy reads what this step wrote.  What the
            // step cannot take — a code longer than the table's root, a copy of more than 64 cells, the last bits of the stream — is left
            // to the one-symbol path behind it, which also reports the errors.
            bool eob = false;
            const uint64_t limit_bits = n_words * 32;
            while (!eob && status == ST_NONE) {
                n_sym++;
                uint32_t taken = 0;
                const uint64_t p = b.position();
                if (p + 256 <= limit_bits) {
                    // lane j: the symbols that would start at bit p + j and at bit p + 64 + j (a window of 128 bit positions per step: the
                    // LDS round trips of a step are what it costs, and both positions' look-ups travel together)
                    const uint64_t pj = p + lane;
                    const uint32_t wj = (uint32_t)(pj >> 5), sj = (uint32_t)pj & 31;
                    const uint32_t d0 = T.stream[wj & 255], d1 = T.stream[(wj + 1) & 255], d2 = T.stream[(wj + 2) & 255], d3 = T.stream[(wj + 3) & 255],
                                   d4 = T.stream[(wj + 4) & 255];
                    const uint32_t xa0 = __builtin_amdgcn_alignbit(d1, d0, sj), xa1 = __builtin_amdgcn_alignbit(d2, d1, sj);
                    const uint32_t xb0 = __builtin_amdgcn_alignbit(d3, d2, sj), xb1 = __builtin_amdgcn_alignbit(d4, d3, sj);
                    const uint32_t ea1 = T.lit[xa0 & ((1u << LIT_ROOT) - 1)], eb1 = T.lit[xb0 & ((1u << LIT_ROOT) - 1)];
                    // -> walk word (bits | cells << 6 | end-of-block << 13 | not-for-this-step << 14) and what a cell needs of its symbol
                    //    (cells << 7 | literal << 14 | (the literal's byte, or the distance) << 15; the first cell's 7 bits come from the walk)
                    auto decode_at = [&](uint32_t x0, uint32_t x1, uint32_t e1, uint32_t& walk, uint32_t& about) {
                        const uint32_t l1 = e1 & 15, sy = e1 >> 4;
                        const bool is_lit = sy < 256, is_eob = sy == 256, is_len = sy > 256;
                        // (a length code's extra bits and distance: computed by every lane, kept by the lanes that saw one — no branch on the lane)
                        const uint32_t ls = is_len ? sy - 257 : 0;
                        const uint32_t leb = ls < 8 || ls >= 28 ? 0u : (ls - 4) >> 2;
                        const uint32_t lbase = ls < 8 ? 3 + ls : ls >= 28 ? 258u : 3 + ((4 + (ls & 3)) << leb);
                        const uint64_t x = ((uint64_t)x1 << 32 | x0) >> l1;
                        const uint32_t len = lbase + ((uint32_t)x & ((1u << leb) - 1));
                        const uint32_t y = (uint32_t)(x >> leb);
                        const uint32_t e2 = T.dist[y & ((1u << DIST_ROOT) - 1)];
                        const uint32_t l2 = e2 & 15, dsy = e2 >> 4;
                        const uint32_t deb = dsy < 4 ? 0u : min((dsy - 2) >> 1, 13u);
                        const uint32_t dbase = dsy < 4 ? 1 + dsy : 1 + ((2 + (dsy & 1)) << deb);
                        const bool bad = e1 == T_LONG || e1 == 0 || sy > 285 || (is_len && (e2 == T_LONG || e2 == 0 || dsy >= 30 || len > 64));
                        const uint32_t bits = is_len ? l1 + leb + l2 + deb : l1;
                        const uint32_t cells = is_len ? len : is_lit ? 1u : 0u;
                        about = cells << 7 | (is_lit ? 1u << 14 : 0u) | (is_lit ? sy : dbase + ((y >> l2) & ((1u << deb) - 1))) << 15;
                        walk = bits | cells << 6 | (is_eob ? 1u << 13 : 0u) | (bad ? 1u << 14 : 0u);
                    };
                    uint32_t walk_a, about_a, walk_b, about_b;
                    decode_at(xa0, xa1, ea1, walk_a, about_a);
                    decode_at(xb0, xb1, eb1, walk_b, about_b);
                    // the chain of symbol starts, from position 0; owner of cell c = the last walked symbol whose first cell is <= c
                    uint32_t pos = 0, total = 0, owner = 0, first_a = 0, first_b = 0;
                    while (pos < 128) {
                        const uint32_t w1 = pos < 64 ? __builtin_amdgcn_readlane(walk_a, pos) : __builtin_amdgcn_readlane(walk_b, pos - 64);
                        const uint32_t c = (w1 >> 6) & 127;
                        if ((w1 >> 14) || total + c > 64) break;
                        owner = lane >= total ? pos : owner;
                        first_a = lane == pos ? total : first_a;
                        first_b = lane + 64 == pos ? total : first_b;
                        total += c;
                        taken++;
                        pos += w1 & 63;
                        if (w1 >> 13 & 1) { eob = true; break; }
                    }
                    if (taken) {
                        if (!o.room(total)) { status = ST_OVERFLOW; break; }
                        const uint32_t from_a = __shfl(about_a | first_a, owner & 63), from_b = __shfl(about_b | first_b, owner & 63);
                        const uint32_t mine_about = owner < 64 ? from_a : from_b;
                        const uint32_t my_first = mine_about & 127, my_len = (mine_about >> 7) & 127, my_val = mine_about >> 15;
                        const bool mine = lane < total, my_lit = (mine_about >> 14 & 1) != 0;
                        const uint32_t i = lane - my_first;                  // my cell's index in its symbol; my_val: its distance
                        const int src = (int)(o.n + my_first) - (int)my_val + (int)(my_val >= my_len ? i : my_val == 1 ? 0u : i % max(my_val, 1u));
                        const bool is_copy = mine && !my_lit;
                        if (__ballot(is_copy && src < 0)) {
                            if (!have_window || __ballot(is_copy && src < -(int)WINDOW)) { status = ST_ERR_DISTANCE; break; }
                            flags |= 1;
                        }
                        const bool far = is_copy && src >= 0 && (int)(o.n + lane) - src > (int)(RING - 128);
                        n_far += __ballot(far) ? 1u : 0u;
                        // literals now; copies in rounds
                        o.ring[mine && my_lit ? (o.n + lane) & RING_MASK : RING + lane] = (uint16_t)my_val;
                        uint64_t pending = __ballot(is_copy);
                        while (pending) {
                            const uint32_t frontier = o.n + (uint32_t)__builtin_ctzll(pending);   // every cell in front of it has its value
                            const bool can = is_copy && (pending >> lane & 1) && src < (int)frontier;
                            const uint32_t at = (uint32_t)max(src, 0);
                            uint16_t v = o.ring[at & RING_MASK];
                            if (__ballot(can && far) && !o.dry) { const uint16_t g = o.out[o.region + (far ? at : 0u)]; v = far ? g : v; }
                            o.ring[can ? (o.n + lane) & RING_MASK : RING + lane] = src < 0 ? (uint16_t)(256 + WINDOW + src) : v;
                            pending &= ~__ballot(can);
                        }
                        o.advance(total);
                        b.move_to(p + pos);
                    }
                }
                if (taken || eob || status != ST_NONE) continue;
                // ---- one symbol, the slow way
                b.sync();
                b.refill();
                uint32_t e = uni(T.lit[b.peek(LIT_ROOT)]);
                if (e == T_LONG) e = uni(decode_long((uint32_t)b.acc, LIT_ROOT, lit_max, T.lit_sorted, T.lit_first, T.lit_count, T.lit_off));
                if (e == 0) { status = ST_ERR_CODE; break; }
                b.drop(e & 15);
                const uint32_t sym = e >> 4;
                if (sym < 256) {
                    if (!o.room(1)) { status = ST_OVERFLOW; break; }
                    o.ring[lane == 0 ? o.n & RING_MASK : RING + lane] = (uint16_t)sym;
                    o.advance(1);
                    continue;
                }
                if (sym == 256) { eob = true; continue; }
                const uint32_t s = sym - 257;
                if (s >= 29) { status = ST_ERR_CODE; break; }
                uint32_t len;
                if (s < 8) len = 3 + s;
                else if (s == 28) len = 258;
                else { const uint32_t eb = (s - 4) >> 2; len = 3 + ((4 + (s & 3)) << eb) + b.take(eb); }
                b.refill();
                uint32_t de = uni(T.dist[b.peek(DIST_ROOT)]);
                if (de == T_LONG) de = uni(decode_long((uint32_t)b.acc, DIST_ROOT, dist_max, T.dist_sorted, T.dist_first, T.dist_count, T.dist_off));
                if (de == 0) { status = ST_ERR_CODE; break; }
                b.drop(de & 15);
                const uint32_t ds = de >> 4;
                if (ds >= 30) { status = ST_ERR_CODE; break; }
                uint32_t dist;
                if (ds < 4) dist = 1 + ds;
                else { const uint32_t eb = (ds - 2) >> 1; dist = 1 + ((2 + (ds & 1)) << eb) + b.take(eb); }
                if (b.over) { status = ST_ERR_OVERRUN; break; }
                if (dist > o.n) {
                    if (!have_window || dist - o.n > WINDOW) { status = ST_ERR_DISTANCE; break; }
                    flags |= 1;
                }
                if (!o.room(len)) { status = ST_OVERFLOW; break; }
                // sources: cell n - dist + (i mod dist) for i < len — all of them in front of n, whatever the overlap
                const int src0 = (int)o.n - (int)dist;
                const bool near = dist + len <= RING;
                n_far += near ? 0u : 1u;
                for (uint32_t base = 0; base < len; base += 64) {
                    const uint32_t i = min(base + lane, len - 1);
                    const int src = src0 + (int)(dist >= len ? i : dist == 1 ? 0u : i % dist);
                    const uint32_t at = (uint32_t)max(src, 0);
                    // (two loads and a select of the VALUES: a select between an LDS and a global POINTER becomes a flat load whose
                    //  aperture test hipcc 7.2 cannot select — "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base")
                    const bool from_ring = near || o.dry;
                    const uint16_t in_ring = o.ring[at & RING_MASK];
                    uint16_t got = in_ring;
                    if (!from_ring) got = *const_cast<const volatile uint16_t*>(&o.out[o.region + at]);   // (volatile: not to be merged with the load above)
                    o.ring[base + lane < len ? (o.n + i) & RING_MASK : RING + lane] = src < 0 ? (uint16_t)(256 + WINDOW + src) : got;
                }
                o.advance(len);
            }
            b.sync();
            if (status != ST_NONE) break;
            if (b.over) { status = ST_ERR_OVERRUN; break; }
        }
        if (bfinal) { status = ST_FINAL; end_bit = b.bitpos(); }
    }
    o.finish();
    if (lane == 0) {
        BlockResult r;
        r.end_bit = end_bit;
        r.n_out = o.n;
        r.status = status;
        r.flags = flags | (o.dry ? 2u : 0u);
        r.pad = 0;
        r.region = o.region;
        r.stats[0] = n_sym;
        r.stats[1] = n_far;
        r.stats[2] = (uint32_t)(t_build >> 10);
        r.stats[3] = (uint32_t)((__builtin_readcyclecounter() - t_start) >> 10);
        res[k] = r;
    }
}

// =================================================================================================================================
// windows
// =================================================================================================================================
struct PlanBlock {                      // one chain block, as the device needs it
    const uint16_t* cells;              // its region of cells (in the first pass's buffer, or in the buffer of the blocks decoded again)
    unsigned long long out_off;
    uint32_t n_out;
    uint32_t flags;
};

constexpr int WIN_TPB = 1024;
// group g = chain blocks [g * G, min(K, (g + 1) * G)).  S (in LDS, 2 x 32 Ki cells): the window in front of the current block as a
// function of the window in front of the group's first block.  sfn[k] receives the S of block k, ffn[g] the one behind the group.
__global__ __launch_bounds__(WIN_TPB) void winfn_kernel(const PlanBlock* __restrict__ plan, uint32_t K, uint32_t G, uint16_t* __restrict__ sfn,
                                                        uint16_t* __restrict__ ffn) {
    extern __shared__ uint16_t S[];     // [2][WINDOW]
    const uint32_t g = blockIdx.x, k0 = g * G, k1 = min(K, k0 + G);
    uint16_t* cur = S;
    uint16_t* nxt = S + WINDOW;
    for (uint32_t j = threadIdx.x; j < WINDOW; j += WIN_TPB) cur[j] = (uint16_t)(256 + j);
    __syncthreads();
    for (uint32_t k = k0; k < k1; k++) {
        uint16_t* dst = sfn + (uint64_t)k * WINDOW;
        for (uint32_t j = threadIdx.x; j < WINDOW; j += WIN_TPB) dst[j] = cur[j];
        const uint32_t t = plan[k].n_out;
        const uint16_t* c = plan[k].cells;
        if (t >= WINDOW) {
            const uint16_t* tail = c + (t - WINDOW);
            for (uint32_t j = threadIdx.x; j < WINDOW; j += WIN_TPB) {
                const uint16_t v = tail[j];
                nxt[j] = v < 256 ? v : cur[v - 256];
            }
        } else {
            const uint32_t keep = WINDOW - t;                                   // the older window, moved down
            for (uint32_t j = threadIdx.x; j < WINDOW; j += WIN_TPB) {
                if (j < keep) nxt[j] = cur[j + t];
                else { const uint16_t v = c[j - keep]; nxt[j] = v < 256 ? v : cur[v - 256]; }
            }
        }
        __syncthreads();
        uint16_t* sw = cur; cur = nxt; nxt = sw;
    }
    uint16_t* dst = ffn + (uint64_t)g * WINDOW;
    for (uint32_t j = threadIdx.x; j < WINDOW; j += WIN_TPB) dst[j] = cur[j];
}

// the real window in front of every group, one group after the other
__global__ __launch_bounds__(WIN_TPB) void winchain_kernel(const uint16_t* __restrict__ ffn, uint32_t n_groups, uint8_t* __restrict__ rwin) {
    __shared__ uint8_t R[2][WINDOW];
    uint8_t* cur = R[0];
    uint8_t* nxt = R[1];
    for (uint32_t j = threadIdx.x; j < WINDOW; j += WIN_TPB) cur[j] = 0;
    __syncthreads();
    for (uint32_t g = 0; g < n_groups; g++) {
        uint8_t* dst = rwin + (uint64_t)g * WINDOW;
        const uint16_t* f = ffn + (uint64_t)g * WINDOW;
        for (uint32_t j = threadIdx.x; j < WINDOW; j += WIN_TPB) {
            dst[j] = cur[j];
            const uint16_t v = f[j];
            nxt[j] = v < 256 ? (uint8_t)v : cur[v - 256];
        }
        __syncthreads();
        uint8_t* sw = cur; cur = nxt; nxt = sw;
    }
}

// =================================================================================================================================
// cells -> bytes
// =================================================================================================================================
constexpr int TR_TPB = 256;
__global__ __launch_bounds__(TR_TPB) void translate_kernel(const PlanBlock* __restrict__ plan, uint32_t G,
                                                           const uint16_t* __restrict__ sfn, const uint8_t* __restrict__ rwin, uint8_t* __restrict__ text) {
    __shared__ uint8_t W[WINDOW];
    const uint32_t k = blockIdx.x;
    const PlanBlock pb = plan[k];
    if (pb.flags & 1) {
        const uint16_t* s = sfn + (uint64_t)k * WINDOW;
        const uint8_t* r = rwin + (uint64_t)(k / G) * WINDOW;
        for (uint32_t j = threadIdx.x; j < WINDOW; j += TR_TPB) { const uint16_t v = s[j]; W[j] = v < 256 ? (uint8_t)v : r[v - 256]; }
        __syncthreads();
    }
    const uint16_t* c = pb.cells;
    uint8_t* o = text + pb.out_off;
    const uint32_t n = pb.n_out;
    // byte-wise up to the first 4-byte boundary of the text, then four cells -> one dword per lane
    const uint32_t head = min(n, (uint32_t)((4 - (pb.out_off & 3)) & 3));
    if (threadIdx.x < head) { const uint16_t v = c[threadIdx.x]; o[threadIdx.x] = v < 256 ? (uint8_t)v : W[v - 256]; }
    const uint32_t quads = (n - head) / 4;
    for (uint32_t q = threadIdx.x; q < quads; q += TR_TPB) {
        const uint16_t* cq = c + head + 4 * q;
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint16_t v = cq[j]; w |= (uint32_t)(v < 256 ? (uint8_t)v : W[v - 256]) << (8 * j); }
        *reinterpret_cast<uint32_t*>(o + head + 4 * q) = w;
    }
    const uint32_t done = head + 4 * quads;
    if (threadIdx.x < n - done) { const uint16_t v = c[done + threadIdx.x]; o[done + threadIdx.x] = v < 256 ? (uint8_t)v : W[v - 256]; }
}

// =================================================================================================================================
// CRC-32
// =================================================================================================================================
constexpr int CRC_TPB = 256;
constexpr uint32_t CRC_LANE = 64;                       // bytes per lane and step
constexpr uint32_t CRC_STEP = 64 * CRC_LANE;            // bytes per wavefront and step: 4 KiB
constexpr uint32_t CRC_SPAN = 16 * CRC_STEP;            // bytes per wavefront: 64 KiB
// member_end[m]: ascending end offsets of the members in the text.  raw[m] ^= (register of a piece, started at 0) shifted by the
// bytes between the piece's end and the member's end.  A wavefront walks 64 KiB in steps of 4 KiB: every lane takes the register
// of its 64 bytes through the slicing tables and multiplies it by x^(8 * bytes behind it in the step) — a constant per lane —, the
// 64 products are XORed across the wave, and the step joins the wave's running register (one more multiplication by a constant);
// the expensive shift by an arbitrary distance (crc_shift: ~30 multiplications) happens once per span, or where a member ends.
// Steps that hold a member boundary (one in sixteen for BGZF) or the text's ragged end go lane by lane.
__global__ __launch_bounds__(CRC_TPB) void crc_kernel(const uint8_t* __restrict__ text, uint64_t total, const unsigned long long* __restrict__ member_end,
                                                      uint32_t n_members, const uint32_t* __restrict__ x2n_g, uint32_t* __restrict__ raw) {
    __shared__ uint32_t T[4][256];
    __shared__ uint32_t x2n[64];
    __shared__ uint32_t lane_k[64];                     // x^(8 * 64 * (63 - lane))
    for (uint32_t i = threadIdx.x; i < 256; i += CRC_TPB) {
        uint32_t c = i;
        for (int b = 0; b < 8; b++) c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1;
        T[0][i] = c;
    }
    if (threadIdx.x < 64) x2n[threadIdx.x] = x2n_g[threadIdx.x];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256; i += CRC_TPB) {
        const uint32_t c1 = (T[0][i] >> 8) ^ T[0][T[0][i] & 0xFF];
        const uint32_t c2 = (c1 >> 8) ^ T[0][c1 & 0xFF];
        const uint32_t c3 = (c2 >> 8) ^ T[0][c2 & 0xFF];
        T[1][i] = c1; T[2][i] = c2; T[3][i] = c3;
    }
    if (threadIdx.x < 64) lane_k[threadIdx.x] = crc_x8n(x2n, (uint64_t)CRC_LANE * (63 - threadIdx.x));
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (CRC_TPB / 64) + (threadIdx.x >> 6);
    const uint64_t span0 = wave * CRC_SPAN;
    if (span0 >= total) return;
    const uint64_t span1 = min(total, span0 + CRC_SPAN);
    const uint32_t k_step = crc_x8n(x2n, CRC_STEP), k_mine = lane_k[lane];
    // the member that holds span0: the first whose end lies behind it
    uint32_t lo = 0, hi = n_members - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (member_end[mid] > span0) hi = mid; else lo = mid + 1; }
    uint32_t m = lo;
    uint32_t acc = 0;                                   // the wave's running register over [acc_from, pos), all of it inside member m
    bool have = false;
    for (uint64_t pos = span0; pos < span1; pos += CRC_STEP) {
        while (member_end[m] <= pos) m++;
        const uint64_t m_end = member_end[m];
        if (pos + CRC_STEP <= span1 && pos + CRC_STEP <= m_end) {
            const uint4* p = reinterpret_cast<const uint4*>(text + pos + (uint64_t)lane * CRC_LANE);
            uint32_t reg = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = p[q];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    reg ^= w[j];
                    reg = T[3][reg & 0xFF] ^ T[2][(reg >> 8) & 0xFF] ^ T[1][(reg >> 16) & 0xFF] ^ T[0][reg >> 24];
                }
            }
            uint32_t x = crc_multmodp(k_mine, reg);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) x ^= __shfl_xor(x, o);
            acc = (have ? crc_multmodp(k_step, acc) : 0) ^ x;
            have = true;
            if (pos + CRC_STEP == m_end) {              // the member ends with this step: its word takes what the wave holds, unshifted
                if (lane == 0) atomicXor(&raw[m], acc);
                have = false;
                acc = 0;
            }
            continue;
        }
        // a member ends inside this step, or the text does: first what the wave holds, then lane by lane
        if (have) {
            if (lane == 0) atomicXor(&raw[m], crc_shift(x2n, acc, m_end - pos));
            have = false;
            acc = 0;
        }
        uint64_t q = pos + (uint64_t)lane * CRC_LANE;
        const uint64_t q_end = min(span1, q + CRC_LANE);
        uint32_t mm = m;
        while (q < q_end) {
            while (member_end[mm] <= q) mm++;
            const uint64_t e = member_end[mm];
            const uint64_t stop = min(q_end, e);
            uint32_t reg = 0;
            for (; q < stop; q++) reg = T[0][(reg ^ text[q]) & 0xFF] ^ (reg >> 8);
            atomicXor(&raw[mm], crc_shift(x2n, reg, e - stop));
        }
    }
    if (have && lane == 0) atomicXor(&raw[m], crc_shift(x2n, acc, member_end[m] - span1));
}

// =================================================================================================================================
// host
// =================================================================================================================================
struct Raw {                            // plain hipMalloc / hipFree: GBs of scratch per call must not stay in the context's pool
    void* p = nullptr;
    size_t cap = 0;
    ~Raw() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    void alloc(size_t bytes) {
        release();
        bytes = std::max<size_t>(bytes, 256);
        const hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); throw HipError{e, "hipMalloc (inflate scratch)", __FILE__, __LINE__}; }
        cap = bytes;
    }
    void reserve(size_t bytes) { if (bytes > cap) alloc(bytes + bytes / 8); }      // contents are not kept
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};
// What a call needs beside the text it returns: 16 x 2 B of cells per compressed byte and 64 KB per block — 13 GB for a 1 Gbp pair.
// ONE set per device and PROCESS, whichever context calls (the calls of a device take turns: a decode launch fills the chip anyway), kept
// between calls: allocating and freeing it per sample cost nothing on an idle box and, sporadically, a SECOND per sample beside another
// process that holds HBM (bench.py's own situation: profiles/r06_gz_e2e_trace.txt, "hold") — and a set per context would multiply it by
// the 16 engines of a `profile`.  A set that grew beyond SCRATCH_KEEP (an unusually large file) is freed behind its call.
constexpr size_t SCRATCH_KEEP = 20ull << 30;
struct InflateScratch {
    Raw gz, c1, c2, cnt, cells, res, jobs, plan, sfn, ffn, rwin, mend, x2n, raw;
    size_t bytes() const { return gz.cap + c1.cap + c2.cap + cnt.cap + cells.cap + res.cap + jobs.cap + plan.cap + sfn.cap + ffn.cap + rwin.cap + mend.cap + x2n.cap + raw.cap; }
    void release() { for (Raw* r : {&gz, &c1, &c2, &cnt, &cells, &res, &jobs, &plan, &sfn, &ffn, &rwin, &mend, &x2n, &raw}) r->release(); }
};
constexpr int MAX_DEVICES = 64;
struct DeviceScratch { std::mutex mu; InflateScratch s; };
DeviceScratch& device_scratch(int device) {
    static DeviceScratch* all = new DeviceScratch[MAX_DEVICES];      // (never destroyed: the process's HIP runtime may be gone by then)
    return all[std::min(std::max(device, 0), MAX_DEVICES - 1)];
}

// a small member through zlib (gzip wrapper: zlib checks CRC and ISIZE itself)
bool zlib_member(const uint8_t* gz, size_t n, size_t p, size_t* end, std::vector<uint8_t>& out) {
    constexpr size_t MAX_OUT = 1u << 20, MAX_IN = 1u << 18;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return false;
    out.resize(MAX_OUT);
    zs.next_in = const_cast<Bytef*>(gz + p);
    zs.avail_in = (uInt)std::min<size_t>(n - p, MAX_IN);
    zs.next_out = out.data();
    zs.avail_out = (uInt)MAX_OUT;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END;
    if (ok) { *end = p + zs.total_in; out.resize(zs.total_out); }
    inflateEnd(&zs);
    return ok;
}

// Several files in one call are ONE stream of gzip members to everything below (`cat a.gz b.gz` is a gzip file): their bytes lie back
// to back on the device, one scan, one decode launch, one chain — the two mates of a pair share the latency of a block's wavefront
// (~15 ms whatever the file's size) instead of paying it one after the other.
void inflate_impl(sylph_inflated* t, const void* const* gzs, const uint64_t* n_bytes, uint32_t n_files) {
    sylph_ctx* ctx = t->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard dg(ctx->device);
    hipStream_t s = ctx->stream;
    Segments segs;
    segs.base.push_back(0);
    for (uint32_t i = 0; i < n_files; i++) {
        if (!member_body((const uint8_t*)gzs[i], (size_t)n_bytes[i], 0)) throw FormatDecline{"not a gzip file"};
        segs.ptr.push_back((const uint8_t*)gzs[i]);
        segs.base.push_back(segs.base.back() + (size_t)n_bytes[i]);
    }
    const uint64_t n = segs.total();
    const size_t body0 = member_body(segs.ptr[0], (size_t)n_bytes[0], 0);
    if (n >= (3ull << 30)) throw FormatDecline{"3 GiB of gzip bytes or more: the host reader takes them"};
    const uint64_t byte_end = n;                    // (trailers and later headers are scanned too: no harm, and members end anywhere)
    // ---- the compressed bytes, padded with zero words
    const uint64_t n_words = (n + 3) / 4;
    DeviceScratch& DS = device_scratch(ctx->device);
    std::lock_guard<std::mutex> turn(DS.mu);                       // one inflate per device at a time: they share the scratch
    InflateScratch& S = DS.s;
    // (on every way out — a decline included — the stream is idle before the next call may touch the scratch; destroyed before `turn`)
    struct Done { InflateScratch& s; hipStream_t st; ~Done() { (void)hipStreamSynchronize(st); if (s.bytes() > SCRATCH_KEEP) s.release(); } } done{S, s};
    Raw& d_gz = S.gz;
    d_gz.reserve((n_words + GZ_PAD_WORDS) * 4);
    {
        HostPhase hp(ctx, "inflate: upload");
        SY_HIP(hipMemsetAsync(d_gz.as<uint8_t>() + (n & ~(uint64_t)3), 0, (n_words + GZ_PAD_WORDS) * 4 - (n & ~(uint64_t)3), s));
        for (uint32_t i = 0; i < n_files; i++) ctx->h2d(d_gz.as<uint8_t>() + segs.base[i], segs.ptr[i], (size_t)n_bytes[i]);
    }
    // ---- candidates
    const uint64_t cap1 = n / 8 + 4096, cap2 = n / 64 + 4096;
    Raw &d_c1 = S.c1, &d_c2 = S.c2, &d_cnt = S.cnt;
    d_c1.reserve(cap1 * 8);
    d_c2.reserve(cap2 * 8);
    d_cnt.reserve(64);
    SY_HIP(hipMemsetAsync(d_cnt.p, 0, 64, s));
    unsigned long long* cnt = d_cnt.as<unsigned long long>();
    unsigned long long n1 = 0, n2 = 0;
    {
        ScopedKernelTimer kt(ctx, "inflate_scan");
        HostPhase hp(ctx, "inflate: scan");
        const uint64_t span = byte_end - body0;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((span + SCAN_TPB - 1) / SCAN_TPB, 256 * 16);
        hipLaunchKernelGGL(scan1_kernel, dim3(std::max(grid, 1u)), dim3(SCAN_TPB), 0, s, d_gz.as<uint32_t>(), (uint64_t)body0, byte_end, d_c1.as<uint64_t>(), cnt, cap1);
        SY_HIP(hipGetLastError());
        ctx->read_back(&n1, cnt, 8);
        if (n1 > cap1) throw FormatDecline{"more header-like bit positions than the scan has room for"};
        if (n1) {
            hipLaunchKernelGGL(scan2_kernel, dim3((uint32_t)((n1 + SCAN2_TPB - 1) / SCAN2_TPB)), dim3(SCAN2_TPB), 0, s, d_gz.as<uint32_t>(), n * 8, d_c1.as<uint64_t>(),
                               cnt, cap1, d_c2.as<uint64_t>(), cnt + 1, cap2);
            SY_HIP(hipGetLastError());
        }
        ctx->read_back(&n2, cnt + 1, 8);
        if (n2 > cap2) throw FormatDecline{"more block-start candidates than the scan has room for"};
    }
    std::vector<uint64_t> cand(n2 + 1);
    if (n2) ctx->d2h(cand.data(), d_c2.p, n2 * 8);
    cand[n2] = (uint64_t)body0 * 8;                 // the stream's first block, whatever its type
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    const uint32_t K = (uint32_t)cand.size();
    if (cand.size() >= (1ull << 31)) throw FormatDecline{"too many candidates"};
    t->n_candidates = K;
    // ---- decode every candidate
    // (SYLPH_HIP_INFLATE_REGION_RATIO: the tests' way to make every block outgrow its region)
    const char* rr_env = getenv("SYLPH_HIP_INFLATE_REGION_RATIO");
    // A single-member file says how well it deflates (ISIZE in its trailer): its blocks get 1.5 x that + 1 cells per compressed byte instead
    // of the flat REGION_RATIO (bench.py's mate file: 5.3 -> 9: 7 GB of cells for a 1 Gbp pair instead of 12.7) — less to allocate, and less
    // for the process to hand back when it ends.  Several members (BGZF ends with an empty one), a wrapped ISIZE: the flat ratio.
    uint64_t auto_ratio = REGION_RATIO;
    {
        double worst = 0;
        bool known = true;
        for (uint32_t i = 0; i < n_files && known; i++) {
            const uint8_t* tr = segs.ptr[i] + n_bytes[i] - 4;
            const double isize = (double)(tr[0] | (uint32_t)tr[1] << 8 | (uint32_t)tr[2] << 16 | (uint32_t)tr[3] << 24), r = isize / (double)n_bytes[i];
            if (r < 1.5 || r > 40) known = false;
            worst = std::max(worst, r);
        }
        if (known) auto_ratio = std::min<uint64_t>(REGION_RATIO, (uint64_t)(worst * 1.5) + 2);
    }
    const uint64_t region_ratio = rr_env ? std::max(1, std::min(64, atoi(rr_env))) : auto_ratio;
    const uint64_t n_cells = (byte_end - body0) * region_ratio + (uint64_t)K * REGION_SLACK + 64;
    Raw &d_cells = S.cells, &d_res = S.res, &d_jobs = S.jobs;
    const uint64_t spill_base = (n_cells + 63) & ~(uint64_t)63;
    d_cells.reserve((spill_base + (uint64_t)SPILL_SLOTS * SPILL_CELLS + 64) * 2);
    d_res.reserve((size_t)K * sizeof(BlockResult));
    d_jobs.reserve((size_t)K * sizeof(DecodeJob));
    SY_HIP(hipMemsetAsync(cnt + 2, 0, 8, s));                                 // the spill arena's counter
    const Spill spill{spill_base, SPILL_SLOTS}, no_spill{0, 0};
    uint32_t* const spill_next = reinterpret_cast<uint32_t*>(cnt + 2);
    std::vector<BlockResult> res(K);
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    auto launch_decode = [&](const DecodeJob* jobs, uint32_t n_jobs, uint16_t* cells, const Spill& sp, BlockResult* out) {
        // the larger ring while all blocks' wavefronts fit on the chip beside each other with it (LDS: ~13 KB per wavefront -> 12 per CU)
        if (n_jobs <= (uint32_t)n_cu * 12 && !getenv("SYLPH_HIP_INFLATE_SMALL_RING"))
            hipLaunchKernelGGL(decode_kernel<4096>, dim3(n_jobs), dim3(64), 0, s, d_gz.as<uint32_t>(), n_words, jobs, cells, sp, spill_next, out);
        else
            hipLaunchKernelGGL(decode_kernel<2048>, dim3(n_jobs), dim3(64), 0, s, d_gz.as<uint32_t>(), n_words, jobs, cells, sp, spill_next, out);
        SY_HIP(hipGetLastError());
    };
    {
        ScopedKernelTimer kt(ctx, "inflate_decode");
        HostPhase hp(ctx, "inflate: decode");
        // a candidate's region: REGION_RATIO cells per compressed byte up to the next candidate, + REGION_SLACK
        std::vector<DecodeJob> jobs(K);
        for (uint32_t i = 0; i < K; i++) {
            const uint64_t b0 = cand[i] >> 3, b1 = i + 1 < K ? cand[i + 1] >> 3 : byte_end;
            jobs[i].start_bit = cand[i];
            jobs[i].region = (b0 - body0) * region_ratio + (uint64_t)i * REGION_SLACK;
            jobs[i].cap = (uint32_t)std::min<uint64_t>(0xFFFFFF00u, (b1 - b0) * region_ratio + REGION_SLACK);
            jobs[i].have_window = i != 0;
        }
        ctx->h2d(d_jobs.p, jobs.data(), (size_t)K * sizeof(DecodeJob));
        launch_decode(d_jobs.as<DecodeJob>(), K, d_cells.as<uint16_t>(), spill, d_res.as<BlockResult>());
    }
    ctx->d2h(res.data(), d_res.p, (size_t)K * sizeof(BlockResult));
    if (getenv("SYLPH_HIP_INFLATE_STATS")) {
        uint64_t sym = 0, far = 0, tb = 0, tt = 0, tmax = 0, cells_out = 0;
        for (const BlockResult& r : res) { sym += r.stats[0]; far += r.stats[1]; tb += r.stats[2]; tt += r.stats[3]; tmax = std::max<uint64_t>(tmax, r.stats[3]); cells_out += r.n_out; }
        uint64_t moved_n = 0, dry_n = 0;
        for (const BlockResult& r : res) { moved_n += r.region >= spill_base; dry_n += (r.flags & 2) != 0; }
        fprintf(stderr, "[sylph_hip inflate] %llu candidates moved to the spill arena, %llu ran dry\n", (unsigned long long)moved_n, (unsigned long long)dry_n);
        fprintf(stderr, "[sylph_hip inflate] %u candidates: %llu symbol-loop turns, %llu copies from global memory, %llu cells; per wave: %.0f k-cycles mean (%.0f on headers + tables), %llu max\n",
                K, (unsigned long long)sym, (unsigned long long)far, (unsigned long long)cells_out, (double)tt / K, (double)tb / K, (unsigned long long)tmax);
    }
    // ---- the chain
    std::vector<std::vector<uint8_t>> host_bytes;
    Chain chain = chain_walk(segs, cand, res.data(), [&](size_t p, size_t* end, uint64_t* n_out) {
        std::vector<uint8_t> o;
        const size_t si = segs.find(p), gb = segs.base[si];
        if (!zlib_member(segs.ptr[si], segs.base[si + 1] - gb, p - gb, end, o)) return false;
        *end += gb;
        *n_out = o.size();
        if (!o.empty()) host_bytes.push_back(std::move(o));
        return true;
    });
    if (!chain.why.empty()) throw FormatDecline{chain.why};
    const uint32_t KB = (uint32_t)chain.blocks.size();
    const uint32_t NM = (uint32_t)chain.members.size();
    t->n_blocks = KB;
    t->n_members = NM;
    t->n_host_members = 0;
    for (const Member& m : chain.members) t->n_host_members += m.on_host;
    t->n = chain.total;
    t->files.assign(n_files, {~0ull, 0});
    for (const Member& m : chain.members) {                   // a file's text: from its first member's first byte to its last member's end
        auto& f = t->files[m.file];
        f.first = std::min<uint64_t>(f.first, m.out_begin);
        f.second = std::max<uint64_t>(f.second, m.out_end);
    }
    for (auto& f : t->files) if (f.first == ~0ull) throw FormatDecline{"a file without a gzip member"};
    // ---- blocks of the chain that outgrew their region (BlockResult::flags bit 1): once more, each into a region of exactly its size
    Raw d_cells2;
    std::vector<uint64_t> redo_region(KB, ~0ull);
    {
        std::vector<DecodeJob> jobs;
        std::vector<uint32_t> which;
        uint64_t at = 0;
        for (uint32_t i = 0; i < KB; i++) {
            const ChainBlock& cb = chain.blocks[i];
            if (!(cb.flags & 2)) continue;
            DecodeJob j;
            j.start_bit = cand[cb.cand];
            j.region = at;
            j.cap = cb.n_out;
            j.have_window = cb.cand != 0;
            redo_region[i] = at;
            at += ((uint64_t)cb.n_out + 63) & ~(uint64_t)63;       // (regions start 128-byte aligned: the ring leaves in 16-byte stores)
            jobs.push_back(j);
            which.push_back(i);
        }
        t->n_redone = jobs.size();
        if (!jobs.empty()) {
            ScopedKernelTimer kt(ctx, "inflate_decode");
            HostPhase hp(ctx, "inflate: decode again");
            Raw d_jobs2, d_res2;
            d_cells2.alloc((at + 64) * 2);
            d_jobs2.alloc(jobs.size() * sizeof(DecodeJob));
            d_res2.alloc(jobs.size() * sizeof(BlockResult));
            ctx->h2d(d_jobs2.p, jobs.data(), jobs.size() * sizeof(DecodeJob));
            launch_decode(d_jobs2.as<DecodeJob>(), (uint32_t)jobs.size(), d_cells2.as<uint16_t>(), no_spill, d_res2.as<BlockResult>());
            std::vector<BlockResult> res2(jobs.size());
            ctx->d2h(res2.data(), d_res2.p, jobs.size() * sizeof(BlockResult));
            for (size_t q = 0; q < jobs.size(); q++) {
                const ChainBlock& cb = chain.blocks[which[q]];
                const BlockResult& a = res[cb.cand];
                const BlockResult& b2 = res2[q];
                if (b2.status != a.status || b2.n_out != a.n_out || b2.end_bit != a.end_bit || (b2.flags & 2))
                    throw FormatDecline{"a block decoded a second time came out differently"};
            }
        }
    }
    // ---- the text
    {
        const hipError_t e = hipMalloc(&t->buf, chain.total + 512);
        if (e != hipSuccess) { t->buf = nullptr; (void)hipGetLastError(); throw HipError{e, "hipMalloc (inflated text)", __FILE__, __LINE__}; }
    }
    uint8_t* text = (uint8_t*)t->buf + 256;
    SY_HIP(hipMemsetAsync(t->buf, 0, 256, s));
    SY_HIP(hipMemsetAsync(text + chain.total, 0, 256, s));
    for (size_t i = 0; i < chain.host.size(); i++) ctx->h2d(text + chain.host[i].out_off, host_bytes[i].data(), host_bytes[i].size());
    if (KB) {
        std::vector<PlanBlock> plan(KB);
        bool any_window = false;
        for (uint32_t i = 0; i < KB; i++) {
            const ChainBlock& cb = chain.blocks[i];
            plan[i].cells = redo_region[i] != ~0ull ? d_cells2.as<uint16_t>() + redo_region[i] : d_cells.as<uint16_t>() + res[cb.cand].region;
            plan[i].out_off = cb.out_off;
            plan[i].n_out = cb.n_out;
            plan[i].flags = cb.flags;
            any_window |= (cb.flags & 1) != 0;
        }
        Raw &d_plan = S.plan, &d_sfn = S.sfn, &d_ffn = S.ffn, &d_rwin = S.rwin;
        d_plan.reserve((size_t)KB * sizeof(PlanBlock));
        ctx->h2d(d_plan.p, plan.data(), (size_t)KB * sizeof(PlanBlock));
        uint32_t G = 1;
        while ((uint64_t)G * G < KB) G++;
        G = std::max(G, 8u);
        const uint32_t NG = (KB + G - 1) / G;
        if (any_window) {
            ScopedKernelTimer kt(ctx, "inflate_windows");
            HostPhase hp(ctx, "inflate: windows");
            d_sfn.reserve((size_t)KB * WINDOW * 2);
            d_ffn.reserve((size_t)NG * WINDOW * 2);
            d_rwin.reserve((size_t)NG * WINDOW);
            static std::once_flag once;
            std::call_once(once, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(winfn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WINDOW * 4); });
            hipLaunchKernelGGL(winfn_kernel, dim3(NG), dim3(WIN_TPB), WINDOW * 4, s, d_plan.as<PlanBlock>(), KB, G, d_sfn.as<uint16_t>(), d_ffn.as<uint16_t>());
            SY_HIP(hipGetLastError());
            hipLaunchKernelGGL(winchain_kernel, dim3(1), dim3(WIN_TPB), 0, s, d_ffn.as<uint16_t>(), NG, d_rwin.as<uint8_t>());
            SY_HIP(hipGetLastError());
        }
        {
            ScopedKernelTimer kt(ctx, "inflate_translate");
            HostPhase hp(ctx, "inflate: translate");
            hipLaunchKernelGGL(translate_kernel, dim3(KB), dim3(TR_TPB), 0, s, d_plan.as<PlanBlock>(), G, d_sfn.as<uint16_t>(), d_rwin.as<uint8_t>(), text);
            SY_HIP(hipGetLastError());
        }
        // ---- CRC-32 of every member
        std::vector<unsigned long long> m_end(NM);
        for (uint32_t i = 0; i < NM; i++) m_end[i] = chain.members[i].out_end;
        uint32_t x2n[64];
        crc_x2n_table(x2n);
        Raw &d_mend = S.mend, &d_x2n = S.x2n, &d_raw = S.raw;
        d_mend.reserve((size_t)NM * 8);
        d_x2n.reserve(256);
        d_raw.reserve((size_t)NM * 4);
        std::vector<uint32_t> raw(NM);
        {
            ScopedKernelTimer kt(ctx, "inflate_crc");
            HostPhase hp(ctx, "inflate: crc");
            ctx->h2d(d_mend.p, m_end.data(), (size_t)NM * 8);
            ctx->h2d(d_x2n.p, x2n, 256);
            SY_HIP(hipMemsetAsync(d_raw.p, 0, (size_t)NM * 4, s));
            if (chain.total) {
                const uint64_t waves = (chain.total + CRC_SPAN - 1) / CRC_SPAN;
                hipLaunchKernelGGL(crc_kernel, dim3((uint32_t)((waves + CRC_TPB / 64 - 1) / (CRC_TPB / 64))), dim3(CRC_TPB), 0, s, text, chain.total, d_mend.as<unsigned long long>(), NM,
                                   d_x2n.as<uint32_t>(), d_raw.as<uint32_t>());
                SY_HIP(hipGetLastError());
            }
        }
        ctx->d2h(raw.data(), d_raw.p, (size_t)NM * 4);
        for (uint32_t i = 0; i < NM; i++) {
            const Member& m = chain.members[i];
            if (m.on_host) continue;
            const uint32_t got = crc_finish(x2n, raw[i], m.out_end - m.out_begin);
            if (got != m.crc) {
                char msg[128];
                snprintf(msg, sizeof(msg), "member %u: CRC-32 %08x, the trailer says %08x", i, got, m.crc);
                throw FormatDecline{msg};
            }
        }
    }
    SY_HIP(hipStreamSynchronize(s));      // the scratch is freed on the way out
}

}  // namespace
}  // namespace sylph

using namespace sylph;

extern "C" {

int sylph_inflate_files(sylph_ctx* ctx, const void* const* gz, const uint64_t* n_bytes, uint32_t n_files, int mem, sylph_inflated** out) {
    if (!ctx || !out || !gz || !n_bytes || !n_files) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    for (uint32_t i = 0; i < n_files; i++) if (!gz[i]) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    if (mem != SYLPH_MEM_HOST && mem != SYLPH_MEM_HOST_PINNED) { set_error("sylph_inflate: the compressed bytes must lie in host memory (mem kind %d)", mem); return SYLPH_ERR_INVALID; }
    *out = nullptr;
    ctx->refs.fetch_add(1);
    sylph_inflated* t = nullptr;
    int format = 0;
    const int rc = guarded([&] {
        t = new sylph_inflated();
        t->ctx = ctx;
        try { inflate_impl(t, gz, n_bytes, n_files); }
        catch (const FormatDecline& e) { set_error("sylph_inflate: declined: %s", e.msg.c_str()); format = 1; }
    });
    if (rc != SYLPH_OK || format) {
        {
            std::lock_guard<std::mutex> lock(ctx->mu);
            DeviceGuard dg(ctx->device);
            (void)hipStreamSynchronize(ctx->stream);
            if (t && t->buf) (void)hipFree(t->buf);
            delete t;
        }
        ctx_unref(ctx);
        return rc != SYLPH_OK ? rc : SYLPH_ERR_FORMAT;
    }
    *out = t;
    return SYLPH_OK;
}

int sylph_inflate(sylph_ctx* ctx, const void* gz, uint64_t n_bytes, int mem, sylph_inflated** out) {
    return sylph_inflate_files(ctx, &gz, &n_bytes, 1, mem, out);
}

int sylph_inflated_file(const sylph_inflated* t, uint32_t i, const void** dev_text, uint64_t* n_bytes) {
    if (!t || i >= t->files.size()) { set_error("sylph_inflated_file: no such file"); return SYLPH_ERR_INVALID; }
    if (dev_text) *dev_text = t->text() + t->files[i].first;
    if (n_bytes) *n_bytes = t->files[i].second - t->files[i].first;
    return SYLPH_OK;
}

int sylph_inflated_text(const sylph_inflated* t, const void** dev_text, uint64_t* n_bytes) {
    if (!t) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    if (dev_text) *dev_text = t->text();
    if (n_bytes) *n_bytes = t->n;
    return SYLPH_OK;
}

int sylph_inflated_info(const sylph_inflated* t, uint64_t* n_members, uint64_t* n_blocks, uint64_t* n_candidates, uint64_t* n_host_members,
                        uint64_t* n_decoded_again) {
    if (!t) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    if (n_members) *n_members = t->n_members;
    if (n_blocks) *n_blocks = t->n_blocks;
    if (n_candidates) *n_candidates = t->n_candidates;
    if (n_host_members) *n_host_members = t->n_host_members;
    if (n_decoded_again) *n_decoded_again = t->n_redone;
    return SYLPH_OK;
}

int sylph_inflated_read(sylph_inflated* t, uint64_t first, uint64_t n, void* host_out) {
    return guarded([&] {
        SY_REQUIRE(t && (host_out || n == 0), "null argument");
        SY_REQUIRE(first <= t->n && n <= t->n - first, "sylph_inflated_read: bytes [%llu, +%llu) of %llu", (unsigned long long)first, (unsigned long long)n,
                   (unsigned long long)t->n);
        if (!n) return;
        std::lock_guard<std::mutex> lock(t->ctx->mu);
        DeviceGuard dg(t->ctx->device);
        t->ctx->d2h(host_out, t->text() + first, n);
    });
}

void sylph_inflated_destroy(sylph_inflated* t) {
    if (!t) return;
    sylph_ctx* ctx = t->ctx;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard dg(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);       // kernels that read the text may still be queued
        if (t->buf) (void)hipFree(t->buf);
        delete t;
    }
    ctx_unref(ctx);
}

}  // extern "C"
