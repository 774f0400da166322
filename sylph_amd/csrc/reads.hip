// reads.hip — seeding + per-read bookkeeping fused for SHORT-READ batches (every record <= READ_HALO = 400 bases): one lane
// per RECORD instead of one lane per 64 flat positions (seeds.hip), so that
//   * only the k-mers the reference hashes are hashed — the position kernel also hashes the k-1 positions per read whose
//     k-mer crosses a record boundary (20 % of a 150 bp read) and throws them away later;
//   * the record, the validity rule (avx2_seeding.rs:37-44) and the dedup markers (pair_kmer / pair_kmer_single,
//     sketch.rs:625-688) are known right here, from the packed bases already in LDS: the separate annotate kernel with its
//     scattered marker loads disappears, and the occurrences leave this kernel as finished 32 B records in file order.
// Replaces extract_markers + the marker half of the record loops (sketch.rs:897-959, :771-895) for Illumina-like input;
// long reads and genomes keep the position kernel (boundary waste there is k/L < 1 %).
//
// A workgroup owns the records whose start falls into a block of `rt` (16 B-aligned) base coordinates — rt is chosen by the
// host so that a block holds about 256 records — and loads that block plus a halo of READ_HALO bases on both sides (the end of its last record; mate 1 of a mate 2 that starts the block)
// as a 2-bit big-endian stream F (input may already BE 2-bit: SYLPH_ENC_2BIT, a quarter of the PCIe bytes).  Each lane then
// walks ITS record with no rolling state at all: it keeps three lane-aligned stream words A(g..g+2) (16 bases each, refilled
// from LDS once per 16 k-mers) and their reverse-complement images B = ~pairswap(bitreverse(A)); the forward k-mer t of the
// group and its reverse complement are funnel-shift extracts (v_alignbit_b32) of those registers at COMPILE-TIME shifts,
// both left-aligned in 64 bits with a few garbage bits below (the next base / the complement of the previous one), which
// cannot change which of the two is smaller; one shift drops the garbage from the winner.  4 instructions per k-mer
// instead of the 8 of a rolling update (r01), then the same hash / threshold sequences as the position kernel.
// Hits are accumulated as bit masks (one bit per k-mer, 32 k-mers per LDS word), counted, scanned across the workgroup —
// lanes are in record order, so the scan gives file order — and only then re-hashed and written.
#include <cstddef>

#include "common.h"
#include "device_common.h"
#include "sketch_session.h"

namespace sylph {

namespace {

constexpr int RTPB = 256;
// A workgroup's block of aligned base coordinates is sized by the host so that it holds about RTPB records (rt = 256 x mean
// record length, a multiple of 16): with a fixed 16 KiB block only 109 of the 256 lanes had a 150 bp read to work on.
constexpr int RT_MIN = 4096, RT_MAX = 65536;
constexpr int RH = 400;                            // halo = longest record taken (pair_kmer_single's upper limit, sketch.rs:923)
constexpr int RPAD = 32;                           // lanes that idle behind the longest read of their wave read past the data
constexpr int MASKW = 12;                          // 12 * 32 = 384 >= RH - 20 k-mers per record
constexpr int OFFS_256 = 600;                      // record offsets staged in LDS (256-record blocks)
constexpr int RTPB_RAGGED = 512;                   // lanes (= records per pass) of the kernel's variant for ragged input

template <int K>
struct KC {
    static constexpr uint64_t MASK = (1ull << (2 * K)) - 1;
};

// 64-bit window of 32 bases starting at stream base `b`: base b in bits 63:62
__device__ __forceinline__ uint64_t win64(const uint32_t* sF, uint32_t b) {
    const uint32_t j = b >> 4, ph = (b & 15u) * 2u;
    const uint64_t x = ((uint64_t)sF[j] << 32) | sF[j + 1];
    return (x << ph) | ((uint64_t)sF[j + 2] >> (32u - ph));   // ph = 0: second term shifts a 32-bit value by 32 -> 0
}

// reverse complement of the k-mer in the top 2K bits of x (anything below is ignored): field order reversed by a 64-bit
// bit reversal, the two bits of every field swapped back, complement = bitwise NOT of a 2-bit code
template <int K>
__device__ __forceinline__ uint64_t revcomp_top(uint64_t x) {
    const uint64_t rev = __brevll(x);
    const uint64_t y = ((rev & 0x5555555555555555ull) << 1) | ((rev >> 1) & 0x5555555555555555ull);
    return (~y) & KC<K>::MASK;
}

// every other base of a 32-base window, first base most significant (the order pair_kmer builds its 16-mers in): ev = bases
// 0, 2, .., 30, od = bases 1, 3, .., 31.  Done on the two 32-bit halves: two shift-or steps bring the wanted fields of every
// byte pair side by side (bytes 3 and 1 then hold four bases each), and ONE v_perm_b32 picks those bytes out of both halves —
// 9-11 instructions per 16-mer where five 64-bit shift/or/and stages took about 25.
__device__ __forceinline__ uint32_t even_fields_to_bytes31(uint32_t w) {
    uint32_t t = w & 0xCCCCCCCCu;
    t = (t | (t << 2)) & 0xF0F0F0F0u;
    return t | (t << 4);                       // byte 3 = bases 0,2,4,6 of the half, byte 1 = bases 8,10,12,14
}
__device__ __forceinline__ void evenodd16(uint64_t x, uint32_t& ev, uint32_t& od) {
    const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    // __builtin_amdgcn_perm(a, b, sel): selector 0-3 = bytes of b, 4-7 = bytes of a
    ev = __builtin_amdgcn_perm(even_fields_to_bytes31(hi), even_fields_to_bytes31(lo), 0x07050301u);
    od = __builtin_amdgcn_perm(even_fields_to_bytes31(hi << 2), even_fields_to_bytes31(lo << 2), 0x07050301u);
}

// One k-mer of a group of 16: T = index inside the group.  A0..A2: the lane-aligned forward words of the group and the two
// behind it; Bm, B0..B2: reverse-complement images of the word before the group and of A0..A2.  Forward window = bases
// [T, T + 32) of A0:A1:A2; reverse-complement window = bits [2T - D, 2T - D + 64) of the little-endian multiword Bm:B0:B1:B2
// counted from B0's bit 0, D = 64 - 2K: both hold their k-mer in the top 2K bits.
template <int K, int T, int HV>
__device__ __forceinline__ void kmer_step(uint32_t A0, uint32_t A1, uint32_t A2, uint32_t Bm, uint32_t B0, uint32_t B1, uint32_t B2,
                                          uint64_t thr, uint32_t& mask) {
    constexpr int D = 64 - 2 * K;
    uint32_t fhi, flo, rhi, rlo;
    if constexpr (T == 0) { fhi = A0; flo = A1; }
    else { fhi = __builtin_amdgcn_alignbit(A0, A1, 32 - 2 * T); flo = __builtin_amdgcn_alignbit(A1, A2, 32 - 2 * T); }
    constexpr int OFF = 2 * T - D;
    if constexpr (OFF < 0) { rlo = __builtin_amdgcn_alignbit(B0, Bm, OFF + 32); rhi = __builtin_amdgcn_alignbit(B1, B0, OFF + 32); }
    else if constexpr (OFF == 0) { rlo = B0; rhi = B1; }
    else { rlo = __builtin_amdgcn_alignbit(B1, B0, OFF); rhi = __builtin_amdgcn_alignbit(B2, B1, OFF); }
    const uint64_t f = ((uint64_t)fhi << 32) | flo, rc = ((uint64_t)rhi << 32) | rlo;
    const uint64_t canon = (f < rc ? f : rc) >> D;                                      // seeding.rs:134-139
    if constexpr (HV == 2) {
        // candidate test on the HIGH word: u = hi(h) + 1 - (the low word's carry), `thr` holds hi(T) + 1 here (reads_kernel)
        const uint32_t u = mm_hash64_gfx950_hi1(canon);
        asm("v_cmp_ge_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(u), "s"((uint32_t)thr) : "vcc");
    } else {
        const uint64_t h = HV ? mm_hash64_gfx950(canon) : mm_hash64(canon);
        asm("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(h), "s"(thr) : "vcc");
    }
}
template <int K, int T0, int HV>
__device__ __forceinline__ void kmer_steps8(uint32_t A0, uint32_t A1, uint32_t A2, uint32_t Bm, uint32_t B0, uint32_t B1, uint32_t B2,
                                            uint64_t thr, uint32_t& mask) {
    kmer_step<K, T0 + 0, HV>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 1, HV>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 2, HV>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 3, HV>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 4, HV>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 5, HV>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 6, HV>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 7, HV>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
}

// blk_rec[b] = first record whose aligned start coordinate (off + bias) is >= b * rt, for b in [0, n_blk]
// (also clears the words the short-read kernel accumulates into: the scan sentinel behind the block counts and the
//  long-record / overflow flags — two memset dispatches less per batch)
__global__ __launch_bounds__(256) void block_records_kernel(const uint64_t* __restrict__ off, uint64_t n_rec, uint32_t bias,
                                                            uint32_t rt, uint32_t n_entries, uint32_t* __restrict__ blk_rec,
                                                            uint32_t* __restrict__ count_sentinel, uint32_t* __restrict__ state_words) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) { *count_sentinel = 0; state_words[0] = 0; state_words[1] = 0; }
    if (b >= n_entries) return;
    const uint64_t target = (uint64_t)b * rt;
    uint64_t lo = 0, hi = n_rec;   // first r in [0, n_rec] with off[r] + bias >= target (off is non-decreasing)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] + bias < target) lo = mid + 1; else hi = mid;
    }
    blk_rec[b] = (uint32_t)lo;
}

#ifndef SYLPH_READS_HASH_DEFAULT
#define SYLPH_READS_HASH_DEFAULT 1
#endif
#ifndef SYLPH_READS_WAVES
#define SYLPH_READS_WAVES 5
#endif
template <int K, int HV, int ENC, int TPB>
// (512 lanes are two wavefronts per SIMD and workgroup: three workgroups = six per SIMD — five would leave room for two)
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(TPB == 256 ? SYLPH_READS_WAVES : 6, TPB == 256 ? SYLPH_READS_WAVES : 6))) void reads_kernel(const uint8_t* __restrict__ bases_al, uint32_t bias, uint64_t n_al,
                                                     const uint64_t* __restrict__ off, uint64_t n_rec,
                                                     const uint32_t* __restrict__ blk_rec, uint32_t n_blk, uint32_t it_begin, uint32_t it_end,
                                                     uint32_t rt, uint64_t thr,
                                                     uint32_t cand_slack, int avx2_compat, int paired, int want_markers, uint64_t rec_base,
                                                     uint32_t slot_cap, OccRec* __restrict__ slot_rec,
                                                     uint32_t* __restrict__ slot_key, int key_sh,
                                                     uint32_t* __restrict__ blk_count,
                                                     ReadsState* __restrict__ state, const uint32_t* __restrict__ blk_list,
                                                     uint32_t* __restrict__ spill_slot_of_blk) {
    // TPB = RTPB (256: one record per lane, a block of ~256 records); RTPB_RAGGED (512) is an A/B variant for ragged input (round 6): the
    // pass's records are dealt to the lanes by length, so a wavefront pays for the longest record of ITS share — a quarter of 256 records,
    // an eighth of 512: trimmed reads of 35..151 bp would idle 11 % of their hash loop's lane-steps instead of 23 %.  Measured slower (see
    // push_short_reads): not the default.
    constexpr int RTPB = TPB, OFFS = TPB == 256 ? OFFS_256 : 2 * TPB + 80;
    extern __shared__ uint32_t sF[];                                 // (rt + 2 RH) / 16 + 3 stream words + RPAD
    __shared__ uint64_t s_off[OFFS + 4];
    const uint32_t n_words = (rt + 2 * RH) / 16 + 3;
    __shared__ uint32_t s_mask[MASKW][RTPB];
    __shared__ uint32_t s_wave[RTPB / 64];
    // per lane: the record's markers (they take over the offsets' LDS once the lanes hold their offsets in registers: a
    // block with several passes keeps its offsets in global memory instead), its first hit of the pass (bit 31: has markers),
    // its stream base
    static_assert((OFFS + 4) >= 2 * RTPB, "the marker arrays alias the staged offsets");
    uint64_t* const s_m0 = s_off;
    uint64_t* const s_m1 = s_off + RTPB;
    __shared__ uint32_t s_first[RTPB + 1], s_rel[RTPB];
    __shared__ uint16_t s_perm[RTPB], s_nh[RTPB];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // HV == 2: the loop tests hi(h) only (kmer_step) and yields a SUPERSET of the k-mers below the threshold — about 3 in 2^32 k-mers
    // too many; the survivors' pass, which hashes every candidate exactly anyway, strikes those from the hit masks and the pass's
    // bookkeeping is redone once (s_redo).  cand_slack widens the superset on purpose: the tests' way of making that road common.
    __shared__ uint32_t s_redo, s_redo_deal;
    const uint64_t thr_loop = HV == 2 ? (uint64_t)((uint32_t)(thr >> 32) + 1u + cand_slack) : thr;
    if (tid == 0) s_redo = 0;                                        // (ordered before its first reader by the barriers below)
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give every XCD one contiguous eighth of the
    // blocks, so that the halo a block shares with its neighbour is found in the same L2.  (Outputs are indexed by block,
    // so the order of the results does not depend on this mapping.)
    // (a launch covers the positions [it_begin, it_end) of that dealing: the pipeline launches a sample's last positions separately, see
    //  push_short_reads)
    const uint32_t per_xcd = (n_blk + 7) / 8;
    for (uint32_t it0 = it_begin + blockIdx.x; it0 < it_end; it0 += gridDim.x) {
        const uint32_t it = blk_list ? it0 : (it0 & 7u) * per_xcd + (it0 >> 3);
        if (it >= n_blk) continue;                                   // padding of the last XCD's range (uniform per workgroup)
        const uint32_t blk = blk_list ? blk_list[it] : it;
        const int64_t a0 = (int64_t)blk * rt - RH;                 // aligned coordinate of stream base 0 (multiple of 16)
        // (only the first and the last block of a batch reach outside [0, n_al): every other block loads unconditionally)
        const bool interior = a0 >= 0 && (uint64_t)(a0 + (int64_t)n_words * 16) <= n_al;
        if (interior) {
            if constexpr (ENC == 0) {
                const uint4* __restrict__ src = reinterpret_cast<const uint4*>(bases_al + a0);
                for (uint32_t ci = tid; ci < n_words; ci += RTPB) sF[ci] = pack16_fwd(src[ci]);
            } else {
                const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(bases_al + (a0 >> 2));
                for (uint32_t ci = tid; ci < n_words; ci += RTPB) sF[ci] = __builtin_amdgcn_perm(0u, src[ci], 0x00010203u);
            }
        } else {
            for (uint32_t ci = tid; ci < n_words; ci += RTPB) {
                const int64_t a = a0 + (int64_t)ci * 16;              // aligned base coordinate of this word's first base
                if constexpr (ENC == 0) {                              // ASCII: 16 bytes -> one word
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (a >= 0 && (uint64_t)a < n_al) v = *reinterpret_cast<const uint4*>(bases_al + a);
                    sF[ci] = pack16_fwd(v);
                } else {                                               // packed input: 4 bytes, first base in bits 7:6 of byte 0
                    uint32_t w = 0;
                    if (a >= 0 && (uint64_t)a < n_al) w = *reinterpret_cast<const uint32_t*>(bases_al + (a >> 2));
                    sF[ci] = __builtin_amdgcn_perm(0u, w, 0x00010203u);
                }
            }
        }
        const uint64_t R0 = blk_rec[blk], R1 = blk_rec[blk + 1];
        // offsets of the block's records (+ the mate-1 offset in front of a block that starts with a mate 2, + two behind)
        const uint64_t w_lo = paired ? (R0 & ~1ull) : R0;
        const uint64_t w_hi = min(R1 + 2, n_rec);                   // last offset index needed
        const bool in_lds = (w_hi - w_lo + 1) <= (uint64_t)(OFFS + 4) && (R1 - R0) <= (uint64_t)RTPB;
        if (in_lds)
            for (uint64_t t = tid; t <= w_hi - w_lo; t += RTPB) s_off[t] = off[w_lo + t];
        __syncthreads();
        uint32_t base_prev = 0;                                      // survivors of the earlier passes of this block
        const uint64_t out0 = (uint64_t)it * slot_cap;
        for (uint64_t pass = R0; pass < R1; pass += RTPB) {         // more than 256 records in a block: only tiny reads
            const uint64_t r = pass + tid;
            const bool active = r < R1;
            uint64_t start = 0, L = 0, s1 = 0, s2 = 0, e2 = 0;
            if (active) {
                if (in_lds) {
                    start = s_off[r - w_lo];
                    L = s_off[r + 1 - w_lo] - start;
                    if (paired) { const uint64_t r1 = (r & ~1ull) - w_lo; s1 = s_off[r1]; s2 = s_off[r1 + 1]; e2 = s_off[r1 + 2]; }
                } else {
                    start = off[r];
                    L = off[r + 1] - start;
                    if (paired) { const uint64_t r1 = r & ~1ull; s1 = off[r1]; s2 = off[r1 + 1]; e2 = off[r1 + 2]; }
                }
            }
            const bool too_long = active && (L > (uint64_t)RH || (paired && (s2 - s1 > (uint64_t)RH || e2 - s2 > (uint64_t)RH)));
            if (too_long) state->long_record = 1u;                  // the host reruns the batch through the position kernel
            const uint32_t nh = (active && !too_long) ? (uint32_t)n_hashed_kmers(L, K, avx2_compat, 0) : 0u;
            const uint32_t rel = active ? (uint32_t)((int64_t)(start + bias) - a0) : 0u;   // stream base of the record
            // A wavefront walks as many k-mer groups as its LONGEST record has: deal the pass's records to the lanes in order of
            // length (counting sort by the number of half-groups, longest first; ties in any order), so that trimmed reads of
            // 35..151 bp cost a wavefront what its own quartile needs instead of what the longest read of 64 needs.  Only the
            // hash loop runs in this order — hit masks are stored under the record's own slot and everything after the loop is
            // indexed by record again, so the output does not depend on the dealing.  A pass of equally long records (every
            // lane in one bin) is hashed as it lies.
            uint32_t* const s_hist = s_first;                         // (s_first is not in use before the survivor pass)
            if (tid < 64) s_hist[tid] = 0;
            s_rel[tid] = rel;
            s_nh[tid] = (uint16_t)nh;
#ifndef SYLPH_READS_ALWAYS_DEAL
            // Round 6 (VERDICT r05 #4a): whether the pass needs dealing at all is asked FIRST — one compare with the first record's count and
            // one LDS word — and a pass of equally long records (every pass of an untrimmed 2 x 150 bp sample) skips the histogram: its 256
            // atomic adds all went to ONE counter (the LDS serialises them), then a scan and a barrier, to find that nothing moves.
            if (tid == 0) s_redo_deal = 0;
            __syncthreads();
            if (__ballot(nh != (uint32_t)s_nh[0]) && lane == 0) s_redo_deal = 1;
            __syncthreads();
            const bool dealt = s_redo_deal != 0;                       // uniform over the workgroup
            const uint32_t bin = 63u - ((nh + 7) >> 3);              // nh <= RH - 20: at most 48 half-groups
            uint32_t hg_max = (nh + 7) >> 3, arrival = 0, hc = 0;
            if (dealt) {
                arrival = atomicAdd(&s_hist[bin], 1u);
                __syncthreads();
                hc = s_hist[lane];
                // the longest record of the pass (uniform over the workgroup: every wavefront reads the same histogram): its lowest
                // non-empty bin; the mask rows above its words are free in this pass (see the survivors' list below)
                const uint64_t bins_used = __ballot(hc != 0u);
                hg_max = bins_used ? 63u - (uint32_t)(__ffsll((unsigned long long)bins_used) - 1) : 0u;   // half-groups of 8 k-mers
            }
#else
            __syncthreads();
            const uint32_t bin = 63u - ((nh + 7) >> 3);              // nh <= RH - 20: at most 48 half-groups
            const uint32_t arrival = atomicAdd(&s_hist[bin], 1u);
            __syncthreads();
            const uint32_t hc = s_hist[lane];
            const bool dealt = s_hist[bin] != (uint32_t)RTPB;         // uniform over the workgroup
            const uint64_t bins_used = __ballot(hc != 0u);
            const uint32_t hg_max = bins_used ? 63u - (uint32_t)(__ffsll((unsigned long long)bins_used) - 1) : 0u;   // half-groups of 8 k-mers
#endif
            const uint32_t rows_used = min((uint32_t)MASKW, (hg_max * 8u + 31u) >> 5);
            if (dealt) {
                const uint32_t incl = wave_inclusive_sum(hc);
                s_perm[(uint32_t)__shfl((int)(incl - hc), (int)bin) + arrival] = (uint16_t)tid;
                __syncthreads();
            }
            const uint32_t slot = dealt ? s_perm[tid] : tid;         // the record slot this lane hashes
            const uint32_t rel_h = dealt ? s_rel[slot] : rel, nh_h = dealt ? s_nh[slot] : nh;
            uint32_t nh_max = nh_h;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) nh_max = max(nh_max, (uint32_t)__shfl_xor((int)nh_max, d));
            if (nh_max) {
                // lane-aligned stream words A(j) = bases [rel + 16 j, rel + 16 j + 16): one LDS read and one 64-bit shift each
                const uint32_t w0 = rel_h >> 4, sh = 32u - (rel_h & 15u) * 2u;      // sh in [2, 32]
                uint32_t raw = sF[w0], nxt = sF[w0 + 1];
                auto next_word = [&](uint32_t j) {                                   // A(j), advancing the raw pair to j + 1
                    const uint32_t a = (uint32_t)((((uint64_t)raw << 32) | nxt) >> sh);
                    raw = nxt;
                    nxt = sF[w0 + j + 2];
                    return a;
                };
                uint32_t A0 = next_word(0), A1 = next_word(1), A2 = next_word(2);
                uint32_t Bm = 0, B0 = rcword(A0), B1 = rcword(A1), B2 = rcword(A2);  // (Bm only feeds garbage bits of group 0)
                const uint32_t n_half = (nh_max + 7) >> 3;                           // half-groups of 8 k-mers (uniform per wave)
                // k-mer i <-> bit 31 - (i & 31) of word i >> 5: a group of 16 is one 16-bit half of its word, even groups the
                // upper half.  Stored as halves (ds_write_b16): no "is the word complete" bookkeeping in the loop; whatever a
                // half that was never written holds lies beyond nh and is cleared with the tail below.
                uint16_t* const mask_half = reinterpret_cast<uint16_t*>(&s_mask[0][slot]);
                const uint32_t n_grp = n_half >> 1;
                // whole groups of 16: one straight-line block (the odd half-group of a record's tail is done after the loop —
                // with the test inside the loop the second half of every group lived in its own basic block and took two
                // extra register moves per k-mer)
                for (uint32_t g = 0; g < n_grp; g++) {
                    uint32_t mask = 0;
                    kmer_steps8<K, 0, HV>(A0, A1, A2, Bm, B0, B1, B2, thr_loop, mask);
                    kmer_steps8<K, 8, HV>(A0, A1, A2, Bm, B0, B1, B2, thr_loop, mask);
                    A0 = A1; A1 = A2; A2 = next_word(g + 3);
                    Bm = B0; B0 = B1; B1 = B2; B2 = rcword(A2);
                    mask_half[((g >> 1) * RTPB * 2) + ((g & 1u) ^ 1u)] = (uint16_t)mask;
                }
                if (n_half & 1u) {                                                   // a last half-group: its 8 k-mers are the top byte
                    uint32_t mask = 0;
                    kmer_steps8<K, 0, HV>(A0, A1, A2, Bm, B0, B1, B2, thr_loop, mask);
                    mask_half[((n_grp >> 1) * RTPB * 2) + ((n_grp & 1u) ^ 1u)] = (uint16_t)(mask << 8);
                }
            }
            if (dealt) __syncthreads();                                 // masks were written by other lanes
            uint32_t total = 0;
            for (;;) {                                                  // (once; HV == 2: again after a candidate failed the exact test)
            // count the real hits of this lane's own record (bit 31 - (i & 31) of word i >> 5 <-> k-mer i < nh)
            uint32_t cnt = 0;
            const uint32_t nw = (nh + 31) >> 5;
            for (uint32_t w = 0; w + 1 < nw; w++) cnt += __popc(s_mask[w][tid]);
            if (nw) {                                                   // the record's last word: bits beyond k-mer nh - 1 are not its own
                const uint32_t left = nh - (nw - 1) * 32;               // 1 .. 32 k-mers in it
                const uint32_t m = s_mask[nw - 1][tid] & (uint32_t)(0xFFFFFFFF00000000ull >> left);
                s_mask[nw - 1][tid] = m;
                cnt += __popc(m);
            }
            const uint32_t x = wave_inclusive_sum(cnt);
            __syncthreads();
            if (lane == 63) s_wave[wave] = x;
            __syncthreads();
            uint32_t before = 0;
            total = 0;
#pragma unroll
            for (int w = 0; w < RTPB / 64; w++) {
                const uint32_t t = s_wave[w];
                if ((uint32_t)w < wave) before += t;
                total += t;
            }
            // Survivors (1 in c k-mers) are finished COOPERATIVELY: lanes publish where their hits start in the pass and their
            // record's markers; then the pass's hits are dealt one per lane — owner found by a search over the 256 start
            // offsets, k-mer index from the owner's hit masks — re-hashed from the LDS stream and written as finished 32 B
            // records, consecutive lanes to consecutive slots.  (Each lane looping over its own hits made every wavefront run
            // the ~95-instruction body as often as its busiest lane had hits: 3-4 times for ~0.6 hits per lane.)
            {
                uint64_t m0 = 0, m1 = 0;
                uint32_t has = 0;
                if (cnt && want_markers) {
                    uint32_t ba = 0, bb = 0;
                    if (!paired) {
                        if (L >= 66 && L <= 400) { has = 1; ba = rel; bb = rel + (uint32_t)(L / 2); }       // sketch.rs:625-656, :923
                    } else if (s2 - s1 >= 33 && e2 - s2 >= 33) {                                             // sketch.rs:659-688
                        has = 1;
                        ba = (uint32_t)((int64_t)(s1 + bias) - a0);
                        bb = (uint32_t)((int64_t)(s2 + bias) - a0);
                    }
                    if (has) {
                        const uint64_t wa64 = win64(sF, ba), wb64 = win64(sF, bb);
                        uint32_t ea, oa, eb, ob;
                        evenodd16(wa64, ea, oa);
                        evenodd16(wb64, eb, ob);
                        m0 = (uint64_t)ea | ((uint64_t)eb << 32);
                        m1 = (uint64_t)oa | ((uint64_t)ob << 32);
                    }
                }
                s_m0[tid] = m0;
                s_m1[tid] = m1;
                s_first[tid] = (before + x - cnt) | (has << 31);
                if (tid == 0) s_first[RTPB] = total;
            }
            // The pass's survivors as a list, hit number -> (lane, k-mer index), written by the lanes that own them (a lane has
            // 0.6 hits on average: a short loop over its own mask words) into the mask rows no record of this pass reaches.
            // Without room for the whole list (records near READ_HALO, or c so small that a pass has thousands of hits) the
            // cooperative pass below finds owner and k-mer by a search over the lanes' start offsets and a walk over the owner's
            // masks instead — what it always did before round 3: ~150 instructions per wavefront and pass more.
            uint32_t* const s_owner = &s_mask[0][0] + rows_used * RTPB;
            const bool listed = total <= ((uint32_t)MASKW - rows_used) * (uint32_t)RTPB;
            if (listed && cnt) {
                uint32_t p = before + x - cnt;
                for (uint32_t w = 0; w < nw; w++) {
                    uint32_t m = s_mask[w][tid];
                    while (m) {
                        const uint32_t bpos = (uint32_t)__clz((int)m);
                        s_owner[p++] = tid | ((w * 32 + bpos) << 10);
                        m &= ~(0x80000000u >> bpos);
                    }
                }
            }
            __syncthreads();
            for (uint32_t hix = tid; hix < total; hix += RTPB) {
                uint32_t lo, i = 0, f;
                if (listed) {
                    const uint32_t ow = s_owner[hix];
                    lo = ow & 0x3FFu;
                    i = ow >> 10;
                    f = s_first[lo];
                } else {
                    lo = 0;
                    uint32_t hi = RTPB;                                         // owner: the last lane whose first hit is <= hix
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((s_first[mid] & 0x7FFFFFFFu) <= hix) lo = mid; else hi = mid;
                    }
                    // (lanes without hits share the start offset of the next lane with hits: the LAST lane with that offset that
                    //  actually owns hix is the one whose successor starts beyond it — the search above returns exactly that lane)
                    f = s_first[lo];
                    uint32_t j = hix - (f & 0x7FFFFFFFu);                       // the j-th hit of lane `lo`
                    for (uint32_t w = 0; w < MASKW; w++) {
                        uint32_t m = s_mask[w][lo];
                        const uint32_t c = (uint32_t)__popc(m);
                        if (j >= c) { j -= c; continue; }
                        for (; j; j--) m &= ~(0x80000000u >> __clz((int)m));   // drop the j highest set bits
                        i = w * 32 + (uint32_t)__clz((int)m);
                        break;
                    }
                }
                const uint32_t o = base_prev + hix;
                if (o < slot_cap) {
                    const uint64_t ft = win64(sF, s_rel[lo] + i);
                    const uint64_t fk = ft >> (64 - 2 * K), rk = revcomp_top<K>(ft);
#ifndef SYLPH_READS_COOP_PLAIN_HASH
                    const uint64_t h = mm_hash64_gfx950(fk < rk ? fk : rk);      // (round 6, VERDICT r05 #4d: the spelling the loop uses — v_lshl_add_u64, v_bitop3)
#else
                    const uint64_t h = mm_hash64(fk < rk ? fk : rk);
#endif
                    if constexpr (HV == 2) {
                        if (h >= thr) {                                          // not a hit after all: out of its record's mask
                            atomicAnd(&s_mask[i >> 5][lo], ~(0x80000000u >> (i & 31u)));
                            s_redo = 1u;
                            continue;
                        }
                    }
                    uint64_t rid = rec_base + pass + lo;
                    if (f >> 31) {
                        rid |= RID_MARKER_BIT;
                        if (paired) rid |= min(emission_rank(i, s_nh[lo], avx2_compat), RID_RANK_MAX) << RID_RANK_SHIFT;   // (sketch_session.h)
                    }
                    slot_rec[out0 + o] = OccRec{h, rid, s_m0[lo], s_m1[lo]};
                    if (slot_key) slot_key[out0 + o] = (uint32_t)(h >> key_sh);   // what finish() partitions by (replay_lds.hip)
                }
            }
            __syncthreads();   // s_wave and s_mask are reused by the next pass
            if constexpr (HV != 2) break;
            else {
                if (!s_redo) break;                                     // (uniform: read behind the barrier)
                __syncthreads();
                if (tid == 0) s_redo = 0;                               // the next writer is several barriers away
            }
            }
            base_prev += total;
        }
        if (tid == 0) {
            if (blk_list) {
                spill_slot_of_blk[blk] = it;
            } else {
                blk_count[blk] = base_prev;      // the true count: a block above slot_cap is redone into a spill region
                if (base_prev > slot_cap) {
                    const uint32_t s = atomicAdd(&state->spill.n_tiles, 1u);
                    if (s < SPILL_MAX_TILES) state->spill.tiles[s] = blk;
                }
            }
        }
        __syncthreads();   // sF / s_off are rewritten by the next block
    }
}

// out[blk_off[b] + i] = slot[b * slot_cap + i] (or its spill region): the occurrences of the batch in file order
__global__ __launch_bounds__(64) void compact_occ_kernel(const OccRec* __restrict__ slot_rec, const uint32_t* __restrict__ blk_count,
                                                         const uint32_t* __restrict__ blk_off, uint32_t n_blk, uint32_t slot_cap,
                                                         uint32_t spill_cap, const OccRec* __restrict__ spill_rec,
                                                         const uint32_t* __restrict__ spill_slot_of_blk,
                                                         uint64_t* __restrict__ out_hash, OccRec* __restrict__ out_rec) {
    for (uint32_t b = blockIdx.x; b < n_blk; b += gridDim.x) {
        const uint32_t n = blk_count[b], d = blk_off[b];
        const OccRec* sr = slot_rec + (uint64_t)b * slot_cap;
        if (n > slot_cap) sr = spill_rec + (uint64_t)spill_slot_of_blk[b] * spill_cap;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const OccRec r = sr[i];
            out_hash[d + i] = r.hash;   // the sort key array of the session
            out_rec[d + i] = r;
        }
    }
}

uint32_t grid_for(uint64_t n, uint32_t tpb = 256) { return (uint32_t)((n + tpb - 1) / tpb); }

// total = sum of the blocks' occurrence counts, written to blk_off[n_blk] — next to the two flag words of the state, so that
// everything the host wants to know about the batch leaves in ONE 12-byte copy (one workgroup: n_blk is a few ten thousand)
__global__ __launch_bounds__(1024) void block_total_kernel(const uint32_t* __restrict__ blk_count, uint32_t n_blk, uint32_t* __restrict__ total) {
    __shared__ uint32_t s[16];
    uint32_t v = 0;
    for (uint32_t i = threadIdx.x; i < n_blk; i += 1024) v += blk_count[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; w++) t += s[w];
        *total = t;
    }
}

// layout of sk->slot_meta: [blk_rec (n_blk+1) | blk_count (n_blk+1) | spill_slot_of_blk (n_blk+1) | blk_off (n_blk+1) | ReadsState]
struct SlotMeta {
    uint32_t *blk_rec, *blk_count, *spill_slot, *blk_off;
    ReadsState* state;
    SlotMeta(sylph_sketch* sk, uint32_t n_blk) {
        blk_rec = sk->slot_meta.as<uint32_t>();
        blk_count = blk_rec + (n_blk + 1);
        spill_slot = blk_count + (n_blk + 1);
        blk_off = spill_slot + (n_blk + 1);
        state = reinterpret_cast<ReadsState*>(blk_off + (n_blk + 1));
    }
};

// the region's occurrences -> the session's dense file-order arrays (hash + OccRec), behind what is there already
void compact_region(sylph_sketch* sk, uint32_t n_blk, uint32_t slot_cap, uint32_t n, uint32_t spill_cap, const OccRec* spill_rec) {
    sylph_ctx* ctx = sk->ctx;
    SlotMeta m(sk, n_blk);
    exclusive_sum_u32(ctx, m.blk_count, m.blk_off, (size_t)n_blk + 1);
    const uint64_t need = sk->n_occ + n;
    sk->hash.grow_keep(need * 8, sk->n_occ * 8, ctx->stream);
    sk->recs.grow_keep(need * sizeof(OccRec), sk->n_occ * sizeof(OccRec), ctx->stream);
    {
        ScopedKernelTimer t(ctx, "compact");
        hipLaunchKernelGGL(compact_occ_kernel, dim3(std::min<uint32_t>(n_blk, 1u << 16)), dim3(64), 0, ctx->stream,
                           sk->slot_rec.as<OccRec>(), m.blk_count, m.blk_off, n_blk, slot_cap, spill_cap, spill_rec, m.spill_slot,
                           sk->hash.as<uint64_t>() + sk->n_occ, sk->recs.as<OccRec>() + sk->n_occ);
        SY_HIP(hipGetLastError());
    }
    sk->n_occ = need;
}

}  // namespace

// The verdict of a deferred batch, read now: the block total + the two flags (what push_short_reads reads when it does not
// defer).  A bad verdict redoes the batch through the checked push; afterwards the session is in the state an ordinary push
// would have left it in.
void resolve_deferred_slots(sylph_sketch* sk) {
    if (!sk->pend.live || !sk->pend.deferred) return;
    sylph_ctx* ctx = sk->ctx;
    SlotMeta m(sk, sk->pend.n_blk);
    hipLaunchKernelGGL(block_total_kernel, dim3(1), dim3(1024), 0, ctx->stream, m.blk_count, sk->pend.n_blk, m.blk_off + sk->pend.n_blk);
    uint32_t res[3] = {0, 0, 0};
    ctx->read_back(res, m.blk_off + sk->pend.n_blk, 12);
    if (res[1] || res[2]) { redo_deferred_batch(sk); return; }
    sk->pend.deferred = false;
    sk->pend.n = res[0];
    if (res[0] == 0) sk->pend = PendingSlots{};
}

void flush_pending_slots(sylph_sketch* sk) {
    resolve_deferred_slots(sk);
    if (!sk->pend.live) return;
    compact_region(sk, sk->pend.n_blk, sk->pend.slot_cap, sk->pend.n, 0, nullptr);
    sk->pend = PendingSlots{};
}

// Short-read path of sylph_sketch_push: returns true with the batch's occurrences taken over by the session — left in their
// slots (sk->pend) when this is the session's first batch and no block overflowed, else appended to the dense arrays (hash +
// OccRec, file order); returns false — having taken nothing — when the batch is not for this kernel (a record longer than
// READ_HALO, or more overflowing blocks than spill regions), and the caller runs the position kernel + annotate instead.
bool push_short_reads(sylph_sketch* sk, const uint8_t* d_bases, uint32_t phase, const uint64_t* d_off, uint64_t n_records, uint64_t n_bases,
                      int enc) {
    sylph_ctx* ctx = sk->ctx;
    // the kernel works in 16 B-aligned coordinates: bias = bases between the aligned address below d_bases and d_bases
    // (a packed stream holds 4 bases per byte and may begin `phase` bases into its first byte)
    const uint32_t bias = enc == SYLPH_ENC_2BIT ? (uint32_t)((uintptr_t)d_bases & 15) * 4u + phase : (uint32_t)((uintptr_t)d_bases & 15);
    const uint8_t* bases_al = d_bases - ((uintptr_t)d_bases & 15);
    const uint64_t n_al = n_bases + bias;
    // block size: about RTPB records per workgroup
    // Equally long records fill the RTPB lanes of every block exactly.  With ragged records the number that start inside a
    // block scatters around its mean (sigma ~ 6 for 35-151 bp reads) and every block above RTPB pays a whole second pass for a
    // handful of records: aim 7 % lower, so that such blocks are rare (c3r: 0.83 -> 0.70 ms per 0.62 Gbp; sweep 85-100 %).
    // Ragged input CAN take the kernel's 512-lane variant (see reads_kernel; SYLPH_HIP_READS_RAGGED_TPB=512) where a block of that many
    // records still fits the stream's LDS window.  It is not the default: measured 8 % SLOWER on c3r (profiles/r06_ab_ragged.txt: 804 against
    // 878 Gbp/s; 754 at five wavefronts per SIMD) — the lane-steps it saves in the hash loop are less than what eight wavefronts waiting
    // for each other at the pass's barriers cost.
    static const int env_ragged_tpb = [] { const char* e = getenv("SYLPH_HIP_READS_RAGGED_TPB"); return e ? atoi(e) : RTPB; }();
    const bool ragged = (n_bases % n_records) != 0;
    const int tpb = (ragged && env_ragged_tpb == RTPB_RAGGED && (uint64_t)RTPB_RAGGED * 93 / 100 * n_bases / n_records <= (uint64_t)RT_MAX) ? RTPB_RAGGED : RTPB;
    const uint64_t target = ragged ? (uint64_t)tpb * 93 / 100 : (uint64_t)tpb;
    uint32_t rt = (uint32_t)std::min<uint64_t>(RT_MAX, std::max<uint64_t>(RT_MIN, target * n_bases / n_records));
    rt = (rt + 15u) & ~15u;
    const uint32_t n_blk = (uint32_t)(n_al / rt) + 1;
    const uint64_t expect = (uint64_t)rt / sk->c;
    const uint32_t spill_cap = rt + RH;
    const uint32_t slot_cap = (uint32_t)std::min<uint64_t>(spill_cap, expect + expect * 3 / 4 + 48);
    const size_t lds_bytes = ((size_t)(rt + 2 * RH) / 16 + 3 + RPAD) * 4;
    const uint64_t thr = UINT64_MAX / (uint64_t)sk->c;
    // the slots belong to the SESSION (from the context's pool): they may outlive this call (sk->pend)
    sk->slot_rec.reserve((size_t)n_blk * slot_cap * sizeof(OccRec));
    sk->slot_key.reserve((size_t)n_blk * slot_cap * 4);
    // (the total — blk_off[n_blk] — and the two flag words of the state sit side by side and leave in ONE 12-byte copy)
    sk->slot_meta.reserve(((size_t)n_blk + 1) * 4 * 4 + sizeof(ReadsState) + 16);
    SlotMeta m(sk, n_blk);
    static_assert(offsetof(ReadsState, long_record) == 0 && offsetof(ReadsState, spill) == 4 && offsetof(SpillState, n_tiles) == 0,
                  "long_record and spill.n_tiles are the first two words");
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    // the k-mer loop's hash / threshold spelling: 0 the compiler's own, 1 mm_hash64_gfx950 + exact 64-bit test, 2 the last hash step and
    // the test on the high word only (a superset the survivors' pass prunes).  Same results all three (tests run each).
    static const int hv_env = getenv("SYLPH_HIP_HASH_VARIANT") ? atoi(getenv("SYLPH_HIP_HASH_VARIANT")) : SYLPH_READS_HASH_DEFAULT;
    const int hv_want = ctx->reads_hash >= 0 ? ctx->reads_hash : hv_env;
    // hi(T) + 1 + slack has to stay a 32-bit number (c = 1: every k-mer passes, T = 2^64 - 1)
    const uint32_t slack = ctx->reads_slack;
    const int hv = (hv_want == 2 && ((thr >> 32) + 1ull + slack) > 0xFFFFFFFFull) ? 1 : hv_want;
    const int key_sh = key_shift(sk->c);
    // positions [it_b, it_e) of the dealing of n_it blocks (reads_kernel)
    auto launch = [&](uint32_t n_it, uint32_t it_b, uint32_t it_e, uint32_t cap, OccRec* sr, uint32_t* skey, const uint32_t* list) {
        if (it_e <= it_b) return;
        const uint32_t grid = ctx->reads_wg_per_cu ? (uint32_t)std::min<uint64_t>(it_e - it_b, (uint64_t)cus * ctx->reads_wg_per_cu) : it_e - it_b;
#define SY_LAUNCH_READS_T(KK, HH, EE, TT)                                                                                             \
    hipLaunchKernelGGL((reads_kernel<KK, HH, EE, TT>), dim3(grid), dim3(TT), lds_bytes, ctx->stream, bases_al, bias, n_al, d_off, n_records, \
                       m.blk_rec, n_it, it_b, it_e, rt, thr, slack, sk->avx2_compat, sk->paired, sk->no_dedup ? 0 : 1, sk->rec_base, cap, sr, skey, key_sh,   \
                       m.blk_count, m.state, list, m.spill_slot)
#define SY_LAUNCH_READS(KK, HH, EE) do { if (tpb == RTPB_RAGGED) SY_LAUNCH_READS_T(KK, HH, EE, RTPB_RAGGED); else SY_LAUNCH_READS_T(KK, HH, EE, RTPB); } while (0)
        if (enc == SYLPH_ENC_2BIT) {
            if (sk->k == 31) { if (hv == 2) SY_LAUNCH_READS(31, 2, 1); else SY_LAUNCH_READS(31, 1, 1); }
            else { if (hv == 2) SY_LAUNCH_READS(21, 2, 1); else SY_LAUNCH_READS(21, 1, 1); }
        } else if (sk->k == 31) { if (hv == 2) SY_LAUNCH_READS(31, 2, 0); else if (hv) SY_LAUNCH_READS(31, 1, 0); else SY_LAUNCH_READS(31, 0, 0); }
        else { if (hv == 2) SY_LAUNCH_READS(21, 2, 0); else if (hv) SY_LAUNCH_READS(21, 1, 0); else SY_LAUNCH_READS(21, 0, 0); }
#undef SY_LAUNCH_READS_T
#undef SY_LAUNCH_READS
        SY_HIP(hipGetLastError());
    };
    {
        HostPhase ph(ctx, "push: reads kernel");
        {
            ScopedKernelTimer t(ctx, "annotate");   // the record lookup this kernel replaces
            hipLaunchKernelGGL(block_records_kernel, dim3(grid_for(n_blk + 1)), dim3(256), 0, ctx->stream, d_off, n_records, bias, rt,
                               n_blk + 1, m.blk_rec, m.blk_count + n_blk, reinterpret_cast<uint32_t*>(m.state));
        }
        ctx->seed_gate();
        {
            // In a pipeline's turn the sample's last positions are a launch of their own and the turn's event sits in front of it: the next
            // sample's kernel starts while this one's last workgroups drain (two seeding kernels share the chip for those few per cent of
            // one — the work is the same, the event-to-wait latency and the drain of a 26,000-workgroup grid are not paid between them).
            // (The tail is a launch of its own for the timers too: one pair of events around both would count the time the tail waits behind
            //  the turn's event — the next sample's kernel is running then — as this kernel's duration.)
            const uint32_t n_round = ((n_blk + 7) / 8) * 8;
            uint32_t cut = n_round;
            if (ctx->turn.done && !ctx->turn.recorded && ctx->reads_tail_pct && n_round >= 64)
                cut = (uint32_t)((uint64_t)n_round * (100 - ctx->reads_tail_pct) / 100) & ~7u;
            {
                ScopedKernelTimer t(ctx, "seeds");
                launch(n_blk, 0, cut, slot_cap, sk->slot_rec.as<OccRec>(), sk->slot_key.as<uint32_t>(), nullptr);
            }
            ctx->seed_done();
            if (cut < n_round) {
                ScopedKernelTimer t(ctx, "seeds");
                launch(n_blk, cut, n_round, slot_cap, sk->slot_rec.as<OccRec>(), sk->slot_key.as<uint32_t>(), nullptr);
            }
        }
        // deferred verdict (sketch_session.h PendingSlots): the caller keeps the batch valid until finish, this is the session's first
        // batch and nothing forces the dense arrays — no block total, no read-back, no wait; finish reads the flags with its own tail
        const uint64_t n_expect = (n_bases > n_records * (uint64_t)(sk->k - 1) ? n_bases - n_records * (uint64_t)(sk->k - 1) : 0) / sk->c;
        if (sk->borrow_until_finish && sk->n_occ == 0 && sk->rec_base == 0 && ctx->finish_mode == 0 && sk->c >= 2 && n_expect >= 4096 &&
            (uint64_t)n_blk * slot_cap < (1ull << 31)) {
            sk->pend = PendingSlots{};
            sk->pend.live = true;
            sk->pend.deferred = true;
            sk->pend.n_blk = n_blk;
            sk->pend.slot_cap = slot_cap;
            sk->pend.n = n_blk * slot_cap;               // upper bound: no block holds more than its slots (else the verdict says so)
            sk->pend.n_expect = (uint32_t)std::min<uint64_t>(n_expect, sk->pend.n);
            sk->pend.bases = d_bases; sk->pend.phase = phase; sk->pend.off = d_off;
            sk->pend.n_records = n_records; sk->pend.n_bases = n_bases; sk->pend.enc = enc;
            if (ctx->profile) ctx->stats["deferred"].launches++;     // (tests ask sylph_ctx_kernel_stats whether this road was taken)
            return true;
        }
        hipLaunchKernelGGL(block_total_kernel, dim3(1), dim3(1024), 0, ctx->stream, m.blk_count, n_blk, m.blk_off + n_blk);
    }
    uint32_t res[3] = {0, 0, 0};   // total occurrences, long_record flag, overflowing blocks
    SY_HIP(hipMemcpyAsync(ctx->pinned, m.blk_off + n_blk, 12, hipMemcpyDeviceToHost, ctx->stream));
    SY_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(res, ctx->pinned, 12);
    if (!ctx->pending.empty()) profile_collect(ctx);
    if (res[1] || res[2] > SPILL_MAX_TILES) return false;
    const uint32_t n = res[0];
    if (n == 0) return true;
    // The first batch of a session stays in its slots: a sample that arrives in one batch is partitioned from there (no
    // compaction pass over its 32 B records at all).  Not with overflowing blocks (their occurrences live in spill regions), not
    // behind occurrences that are in the dense arrays already, not when the device-wide finish was asked for (it wants them dense).
    if (res[2] == 0 && sk->n_occ == 0 && ctx->finish_mode != 1 && sk->c >= 2) {
        sk->pend.live = true;
        sk->pend.n_blk = n_blk;
        sk->pend.slot_cap = slot_cap;
        sk->pend.n = n;
        return true;
    }
    const OccRec* sp_r = nullptr;
    if (res[2]) {   // a few blocks (low-complexity reads) are redone with room for every position
        DevBuf& b_x = ctx->scratch[7];   // spill records
        b_x.reserve((size_t)res[2] * spill_cap * sizeof(OccRec));
        OccRec* xr = b_x.as<OccRec>();
        ScopedKernelTimer ts(ctx, "seeds_spill");
        ScopedKernelTimer t(ctx, "seeds");
        launch(res[2], 0, res[2], spill_cap, xr, nullptr, m.state->spill.tiles);
        sp_r = xr;
    }
    compact_region(sk, n_blk, slot_cap, n, spill_cap, sp_r);
    return true;
}

}  // namespace sylph
