// hash_variants.hip — round 5, VERDICT r04 #3: the k-mer loop of reads_kernel (csrc/reads.hip: two window extracts, canonical choice,
// mm_hash64, threshold mask) with the hash / threshold written four ways, timed alone on the whole chip (5 waves per SIMD, like the
// kernel) on synthetic stream words held in LDS.  Reports ms per 8e8 k-mers and, with rocprofv3 --pmc SQ_INSTS_VALU, instructions
// per k-mer.  Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -I sylph_amd/csrc tools/hash_variants.hip -o /tmp/hash_variants
//   V0  mm_hash64 as the compiler lowers it (64-bit multiplies)
//   V1  mm_hash64_gfx950: what reads_kernel runs (v_lshl_add_u64 chains, the NOT folded into a v_bitop3)
//   V2  V1 with the LAST step on the high word only — h < T needs hi(h) unless hi(h) == hi(T): the mask is a superset (hi' <= hi(T),
//       carry from the low word ignored), the 1-in-200 candidates are re-hashed exactly anyway (the cooperative pass) and would have
//       to be re-TESTED there (2 false candidates per 2^32 k-mers: a tombstone path that the slots do not have today)
//   V3  every 64-bit step spelled out on 32-bit halves (v_alignbit / v_add_co / v_addc): exact
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "device_common.h"

using namespace sylph;

__device__ __forceinline__ uint64_t hash_halves(uint64_t key) {
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32), c;
    auto add_shl = [&](int n) {      // (hi:lo) += (hi:lo) << n, 0 < n < 32
        const uint32_t slo = lo << n, shi = __builtin_amdgcn_alignbit(hi, lo, 32 - n);
        const uint32_t l2 = lo + slo;
        c = l2 < lo ? 1u : 0u;
        hi = hi + shi + c;
        lo = l2;
    };
    auto xor_shr = [&](int n) {      // (hi:lo) ^= (hi:lo) >> n, 0 < n < 32
        lo ^= __builtin_amdgcn_alignbit(hi, lo, n);
        hi ^= hi >> n;
    };
    add_shl(21); lo = ~lo; hi = ~hi;
    xor_shr(24);
    { const uint32_t l0 = lo, h0 = hi; add_shl(3); const uint32_t slo = l0 << 8, shi = __builtin_amdgcn_alignbit(h0, l0, 24); const uint32_t l2 = lo + slo; hi = hi + shi + (l2 < lo ? 1u : 0u); lo = l2; }
    xor_shr(14);
    { const uint32_t l0 = lo, h0 = hi; add_shl(2); const uint32_t slo = l0 << 4, shi = __builtin_amdgcn_alignbit(h0, l0, 28); const uint32_t l2 = lo + slo; hi = hi + shi + (l2 < lo ? 1u : 0u); lo = l2; }
    xor_shr(28);
    add_shl(31);
    return ((uint64_t)hi << 32) | lo;
}

template <int V>
__device__ __forceinline__ void hash_and_mask(uint64_t canon, uint64_t thr, uint32_t& mask) {
    if constexpr (V == 2) {
        // steps 1..6 as V1, then hi(h) without the low word's carry; candidate <=> hi' <= hi(T)
        uint64_t t = lshl_add_u64<0>(canon << 21, canon);
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
        const uint32_t hi = (uint32_t)(t >> 32) + __builtin_amdgcn_alignbit((uint32_t)(t >> 32), (uint32_t)t, 1);
        const uint32_t thi = (uint32_t)(thr >> 32) + 1u;
        asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(hi), "s"(thi) : "vcc");
    } else {
        const uint64_t h = V == 0 ? mm_hash64(canon) : V == 1 ? mm_hash64_gfx950(canon) : hash_halves(canon);
        asm("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(h), "s"(thr) : "vcc");
    }
}

template <int K, int T, int V>
__device__ __forceinline__ void kmer_step(uint32_t A0, uint32_t A1, uint32_t A2, uint32_t Bm, uint32_t B0, uint32_t B1, uint32_t B2, uint64_t thr, uint32_t& mask) {
    constexpr int D = 64 - 2 * K;
    uint32_t fhi, flo, rhi, rlo;
    if constexpr (T == 0) { fhi = A0; flo = A1; }
    else { fhi = __builtin_amdgcn_alignbit(A0, A1, 32 - 2 * T); flo = __builtin_amdgcn_alignbit(A1, A2, 32 - 2 * T); }
    constexpr int OFF = 2 * T - D;
    if constexpr (OFF < 0) { rlo = __builtin_amdgcn_alignbit(B0, Bm, OFF + 32); rhi = __builtin_amdgcn_alignbit(B1, B0, OFF + 32); }
    else if constexpr (OFF == 0) { rlo = B0; rhi = B1; }
    else { rlo = __builtin_amdgcn_alignbit(B1, B0, OFF); rhi = __builtin_amdgcn_alignbit(B2, B1, OFF); }
    const uint64_t f = ((uint64_t)fhi << 32) | flo, rc = ((uint64_t)rhi << 32) | rlo;
    hash_and_mask<V>((f < rc ? f : rc) >> D, thr, mask);
}
template <int K, int T0, int V>
__device__ __forceinline__ void steps8(uint32_t A0, uint32_t A1, uint32_t A2, uint32_t Bm, uint32_t B0, uint32_t B1, uint32_t B2, uint64_t thr, uint32_t& mask) {
    kmer_step<K, T0 + 0, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask); kmer_step<K, T0 + 1, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 2, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask); kmer_step<K, T0 + 3, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 4, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask); kmer_step<K, T0 + 5, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
    kmer_step<K, T0 + 6, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask); kmer_step<K, T0 + 7, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
}

// one lane = one "read" of n_grp groups of 16 k-mers over LDS stream words, as reads_kernel's hot loop
template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void loop_kernel(uint32_t n_grp, uint64_t thr, uint32_t* __restrict__ out) {
    __shared__ uint32_t sF[256 * 10 + 64];
    for (uint32_t i = threadIdx.x; i < 256 * 10 + 64; i += 256) sF[i] = (i + blockIdx.x * 977u) * 2654435761u ^ (i * 40503u);
    __syncthreads();
    const uint32_t w0 = threadIdx.x * 9, sh = 32u - (threadIdx.x & 15u) * 2u;
    uint32_t raw = sF[w0], nxt = sF[w0 + 1], j = 0;
    auto next_word = [&]() {
        const uint32_t a = (uint32_t)((((uint64_t)raw << 32) | nxt) >> sh);
        raw = nxt;
        nxt = sF[w0 + 2 + (j++ % 8)];
        return a;
    };
    uint32_t A0 = next_word(), A1 = next_word(), A2 = next_word();
    uint32_t Bm = 0, B0 = rcword(A0), B1 = rcword(A1), B2 = rcword(A2);
    uint32_t acc = 0;
    for (uint32_t g = 0; g < n_grp; g++) {
        uint32_t mask = 0;
        steps8<31, 0, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
        steps8<31, 8, V>(A0, A1, A2, Bm, B0, B1, B2, thr, mask);
        A0 = A1; A1 = A2; A2 = next_word();
        Bm = B0; B0 = B1; B1 = B2; B2 = rcword(A2);
        acc += mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V>
void run(const char* what, uint32_t* out) {
    const uint32_t n_grp = 120 / 16 + 1;                 // a 150 bp read: 120 k-mers -> 7.5 groups (8 here)
    const uint32_t blocks = 26042;                       // as many workgroups as a 1 Gbp batch has blocks of 256 reads
    const uint64_t thr = UINT64_MAX / 200;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(loop_kernel<V>, dim3(blocks), dim3(256), 0, 0, n_grp, thr, out);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep) best = ms < best ? ms : best;
    }
    const double kmers = (double)blocks * 256 * n_grp * 16;
    printf("%-58s %7.4f ms per launch of %.3e k-mers = %7.4f ms per 8.0e8 k-mers (one 1 Gbp sample of 2 x 150 bp)\n", what, best, kmers, best * 8.0e8 / kmers);
}

int main() {
    uint32_t* out;
    (void)hipMalloc(&out, (size_t)26042 * 256 * 4);
    run<0>("V0 mm_hash64 (compiler's multiplies)", out);
    run<1>("V1 mm_hash64_gfx950 (reads_kernel today)", out);
    run<2>("V2 V1, last step + threshold on the high word (superset)", out);
    run<3>("V3 32-bit halves throughout (exact)", out);
    return 0;
}
