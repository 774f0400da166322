// device_common.h — device-side building blocks shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sylph {

// seeding.rs:4-15 mm_hash64 / avx2_seeding.rs:6-30 mm_hash256.  NB first step is ~(key + (key << 21)).
__device__ __forceinline__ uint64_t mm_hash64(uint64_t key) {
    key = ~(key + (key << 21));
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

// Same function with the instruction selection pinned for gfx950.  Left to itself the compiler folds every
// "x + (x << n)" into a 64-bit multiply and lowers that to 2-3 v_mad_u64_u32 plus operand shuffling (~15 mads and
// ~19 v_mov per k-mer).  Measured issue costs on MI355X (tools/valu_rates.hip; VOP2 32-bit op = 1): v_lshl_add_u64
// 1.9, v_lshlrev/lshrrev_b64 1.6, v_mad_u64_u32 1.9, any VOP3 32-bit op 1.6.  v_lshl_add_u64 (shift <= 4) does
// x*9, x*5 and x*21 in one/two instructions; the NOT of step 1 is folded into step 2 as one xor with a constant:
//   ~t ^ (~t >> 24)  ==  (t ^ (t >> 24)) ^ 0xFFFFFF0000000000.
template <int N>
__device__ __forceinline__ uint64_t lshl_add_u64(uint64_t a, uint64_t b) {   // (a << N) + b, N in 0..4
    uint64_t d;
    asm("v_lshl_add_u64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(N), "v"(b));
    return d;
}
__device__ __forceinline__ uint64_t mm_hash64_gfx950(uint64_t key) {
    uint64_t t = lshl_add_u64<0>(key << 21, key);                 // key + (key << 21)
    {                                                             // ~t, then ^= >> 24: the high word takes ONE three-input xor
        uint64_t sh;                                              // (gfx9 VOP3 has no literals: the constant rides in an SGPR;
        asm("v_lshrrev_b64 %0, 24, %1" : "=v"(sh) : "v"(t));      //  the shift pinned to ONE 64-bit instruction, not lshr + alignbit)
        uint32_t hi;
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(hi) : "v"((uint32_t)(t >> 32)), "v"((uint32_t)(sh >> 32)), "s"(0xFFFFFF00u));
        t = ((uint64_t)hi << 32) | ((uint32_t)t ^ (uint32_t)sh);
    }
    t = lshl_add_u64<0>(t << 8, lshl_add_u64<3>(t, t));           // * 265 = (t << 8) + 9t
    t = t ^ (t >> 14);
    t = lshl_add_u64<4>(t, lshl_add_u64<2>(t, t));                // * 21
    t = t ^ (t >> 28);
    t = lshl_add_u64<0>(t << 31, t);                              // + (t << 31)
    return t;
}
// The same hash up to its last step, then only what `h < T` looks at first: u = hi(t) + bits 1..32 of t + 1 (mod 2^32).  With the
// carry out of the low word's sum, hi(h) = u - 1 + carry: h < T implies u <= hi(T) + 1 (u = 0, the wrapped case, included), so
// { u <= hi(T) + 1 } is a superset of { h < T } that is wrong for about 3 in 2^32 keys (hi(h) in {hi(T), hi(T) + 1, 2^32 - 1}).
// v_alignbit + v_add3 + a 32-bit compare where the exact test takes v_lshlrev_b64 + v_lshl_add_u64 + a 64-bit compare.
__device__ __forceinline__ uint32_t mm_hash64_gfx950_hi1(uint64_t key) {
    uint64_t t = lshl_add_u64<0>(key << 21, key);
    {
        uint64_t sh;
        asm("v_lshrrev_b64 %0, 24, %1" : "=v"(sh) : "v"(t));
        uint32_t hi;
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(hi) : "v"((uint32_t)(t >> 32)), "v"((uint32_t)(sh >> 32)), "s"(0xFFFFFF00u));
        t = ((uint64_t)hi << 32) | ((uint32_t)t ^ (uint32_t)sh);
    }
    t = lshl_add_u64<0>(t << 8, lshl_add_u64<3>(t, t));
    t = t ^ (t >> 14);
    t = lshl_add_u64<4>(t, lshl_add_u64<2>(t, t));
    t = t ^ (t >> 28);
    return (uint32_t)(t >> 32) + __builtin_amdgcn_alignbit((uint32_t)(t >> 32), (uint32_t)t, 1) + 1u;
}

// types.rs:50-59 BYTE_TO_SEQ for one byte, computed instead of looked up:
// A/a=0 C/c=1 G/g=2 T/t/U/u=3; raw bytes 1,2,3 -> 1,2,3; everything else (N, IUPAC, gaps, ...) -> 0.
__device__ __forceinline__ uint32_t byte_to_seq(uint32_t b) {
    const uint32_t u = b & 0xDFu;
    const uint32_t c = ((b >> 1) ^ (b >> 2)) & 3u;
    const bool acgtu = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T') | (u == 'U');
    return acgtu ? c : (b <= 3u ? b : 0u);
}

// 4 ASCII bytes -> 4 two-bit codes (one per byte lane).  Fast path valid when every byte is one of ACGTacgt:
// code = ((b>>1)^(b>>2))&3; v_perm_b32 rebuilds the upper-case letter from the code and the comparison proves
// the byte really was that letter.  `bad` accumulates non-zero if any byte needs the exact slow path.
__device__ __forceinline__ uint32_t codes4_fast(uint32_t w, uint32_t& bad) {
    const uint32_t c = ((w >> 1) ^ (w >> 2)) & 0x03030303u;
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* 'T','G','C','A' */, c);
    bad |= (w & 0xDFDFDFDFu) ^ expect;
    return c;
}
__device__ __forceinline__ uint32_t codes4_exact(uint32_t w) {
    return byte_to_seq(w & 0xFFu) | (byte_to_seq((w >> 8) & 0xFFu) << 8) | (byte_to_seq((w >> 16) & 0xFFu) << 16) |
           (byte_to_seq(w >> 24) << 24);
}

// Bytes that are not ACGTacgt, without leaving the fast path: d = the difference codes4_fast / codes4_bitop found for the dword
// (nonzero byte <=> the byte failed the letter test).  Every such byte that BYTE_TO_SEQ maps to 0 — N, n, IUPAC letters, gaps,
// anything else — just loses its two code bits in c.  The bytes it maps to something else (U / u -> 3, the raw values 1, 2, 3;
// 0 is lumped in) are reported through `other`: only they need codes4_exact.  With 0.1 % N two of three wavefront iterations
// of a load loop see an odd byte somewhere in their 64 x 16 bytes; the exact path costs ~200 instructions, this one 11 per dword.
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) {                 // bit 7 of every byte of x that is not 0
    return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
__device__ __forceinline__ void clear_invalid_codes(uint32_t w, uint32_t d, uint32_t& c, uint32_t& other) {
    const uint32_t inv = nonzero_bytes(d);
    const uint32_t not_u = nonzero_bytes((w & 0xDFDFDFDFu) ^ 0x55555555u), not_small = nonzero_bytes(w & 0xFCFCFCFCu);
    other |= inv & ~(not_u & not_small);
    c &= ~((inv >> 6) | (inv >> 7));
}

// 16 ASCII bases -> forward stream word (base j at bits 30-2j).  Per dword: the 2-bit code of every byte ((b>>1 ^ b>>2) & 3,
// one v_bitop3 after the two shifts), the proof that every byte was one of ACGTacgt (v_perm_b32 rebuilds the letter from the
// code), then ONE multiply gathers the four codes into the top byte, first base most significant: the code of byte k sits at
// bit 8k and must land at bit 30 - 2k, i.e. move left by 30 - 10k; 2^30 + 2^20 + 2^10 + 1 does the four moves at once and no
// partial product reaches bits 24..31 from anywhere else (v_mul_lo_u32 issues at the rate of v_perm_b32, tools/valu_rates.hip:
// 4 of them replace the 8 + 4 permutes and shift-ors of a 4x4 byte transpose).  Two byte-selects and an OR assemble the word.
__device__ __forceinline__ uint32_t codes4_bitop(uint32_t w, uint32_t& diff) {
    uint32_t c;
    const uint32_t a = w >> 1, b = w >> 2, m = 0x03030303u, up = 0xDFDFDFDFu;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x28" : "=v"(c) : "v"(a), "v"(b), "v"(m));      // (a ^ b) & m
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* 'T','G','C','A' */, c);
    // (w & 0xDF..) ^ expect in one instruction; kept opaque so that the four results are OR-ed and tested ONCE (left to itself
    // the compiler turns `bad |= ...; if (bad)` into four compares and a chain of 16-bit boolean ops: 17 instructions for 7)
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6a" : "=v"(diff) : "v"(w), "v"(up), "v"(expect));   // (a & b) ^ c
    return c;
}
__device__ __forceinline__ uint32_t pack16_fwd(uint4 v) {
    uint32_t d0, d1, d2, d3, bad;
    uint32_t c0 = codes4_bitop(v.x, d0), c1 = codes4_bitop(v.y, d1), c2 = codes4_bitop(v.z, d2), c3 = codes4_bitop(v.w, d3);
    asm("v_or3_b32 %0, %1, %2, %3" : "=v"(bad) : "v"(d0), "v"(d1), "v"(d2));
    asm("v_or_b32 %0, %1, %2" : "=v"(bad) : "v"(bad), "v"(d3));
    if (bad) {
        // an odd byte somewhere in the wavefront's 64 x 16: N and the like lose their code bits in place, only U, u and raw 0-3
        // take the exact path (device_common.h)
        uint32_t other = 0;
        clear_invalid_codes(v.x, d0, c0, other); clear_invalid_codes(v.y, d1, c1, other);
        clear_invalid_codes(v.z, d2, c2, other); clear_invalid_codes(v.w, d3, c3, other);
        if (other) { c0 = codes4_exact(v.x); c1 = codes4_exact(v.y); c2 = codes4_exact(v.z); c3 = codes4_exact(v.w); }
    }
    constexpr uint32_t GATHER = (1u << 30) | (1u << 20) | (1u << 10) | 1u;
    const uint32_t p0 = c0 * GATHER, p1 = c1 * GATHER, p2 = c2 * GATHER, p3 = c3 * GATHER;
    // __builtin_amdgcn_perm(hi, lo, sel): selector 0-3 = bytes of lo, 4-7 = bytes of hi, 0x0C = constant 0
    return __builtin_amdgcn_perm(p0, p1, 0x07030C0Cu) | __builtin_amdgcn_perm(p2, p3, 0x0C0C0703u);
}

// reverse-complement image of a stream word: base i of `a` (bits 31-2i, 30-2i) -> its complement at bits 2i+1, 2i
__device__ __forceinline__ uint32_t rcword(uint32_t a) {
    const uint32_t r = __brev(a);
    return ~(((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u));
}

// 16 ASCII bases (memory order x,y,z,w) -> F: base j at bits 30-2j (big-endian), R: (3 - base j) at bits 2j: the forward word and
// its reverse-complement image (rounds 1-2 built R with a second 4x4 byte transpose: 8 v_perm_b32 + 6 shift-ors where the bit
// reversal of F takes 5 instructions).
__device__ __forceinline__ void pack16(uint4 v, uint32_t& F, uint32_t& R) {
    F = pack16_fwd(v);
    R = rcword(F);
}

// Number of k-mer start positions of a length-L sequence that the reference hashes.
//   scalar  (seeding.rs:93,120):              L >= k ? L-k+1 : 0
//   avx2    (avx2_seeding.rs:37-44,95):       4*((L-k+1)/4), nothing if L < min_len (k+1 reads, 2k contigs :160)
__device__ __host__ __forceinline__ uint64_t n_hashed_kmers(uint64_t L, uint32_t k, int avx2_compat, int positions) {
    if (L < k) return 0;
    if (!avx2_compat) return L - k + 1;
    const uint64_t min_len = positions ? 2ull * k : (uint64_t)k + 1;
    if (L < min_len) return 0;
    return ((L - k + 1) / 4) * 4;
}

// Place of the k-mer that starts at base i of a record with nh hashed k-mers in the order extract_markers emits the record's seeds
// (sketch.rs:53-69): position order for the scalar routine (seeding.rs:86-146); the 4-lane AVX2 routine (avx2_seeding.rs:33-148) walks
// four quarters of the record in lock step and pushes lane 0..3 of every step: (i mod len4, i div len4) with len4 = nh / 4.
__device__ __host__ __forceinline__ uint64_t emission_rank(uint32_t i, uint32_t nh, int avx2_compat) {
    if (!avx2_compat || nh < 4) return i;
    const uint32_t len4 = nh >> 2, lane = i / len4;
    return (uint64_t)(i - lane * len4) * 4 + lane;
}

// Adds the number of lanes with `pred` to *counter with ONE atomic per wavefront (64-bit ballot + s_bcnt1).
__device__ __forceinline__ void wave_count_add(unsigned int* counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m && (__lane_id() == (unsigned)__ffsll((long long)m) - 1)) atomicAdd(counter, (unsigned int)__popcll(m));
}

// index of the record containing flat position p: largest r with off[r] <= p (off has n+1 entries, p < off[n]).
__device__ __forceinline__ uint64_t find_record(const uint64_t* __restrict__ off, uint64_t n, uint64_t p) {
    uint64_t lo = 0, hi = n;   // invariant: off[lo] <= p < off[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// Inclusive prefix sum over the 64 lanes of a wavefront on the DPP path: four row shifts inside the rows of 16 lanes, then lane 15 of
// every row to the row behind it and lane 31 to the upper half (row_bcast:15 / :31, the gfx9 wave-scan idiom) — six v_add_u32 with a
// dpp modifier.  The __shfl_up loop it replaces (round 6, VERDICT r05 #4b) is a ds_bpermute, a compare, a select and an add per step.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
#ifndef SYLPH_NO_DPP_SCAN
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
    return x;
#else
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    return x;
#endif
}

}  // namespace sylph
