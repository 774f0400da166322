// valu_rates.hip — measures issue cost (cycles per wave-instruction on one SIMD, 1 wave/SIMD and 2 waves/SIMD) of the
// 64-bit integer building blocks of mm_hash64 on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// 4 independent chains so latency does not dominate
#define DEFK(name, body)                                                                \
    __global__ void name(uint64_t* out, uint64_t* cyc, int iters) {                     \
        uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + 1, b = a ^ 0x1234567, c = a + 77, d = a * 3;  \
        uint32_t x0 = (uint32_t)a, x1 = x0 * 3, x2 = x0 + 9, x3 = x0 ^ 5, y0 = x0 + 1, y1 = x1 + 1, y2 = x2 + 1, y3 = x3 + 1; \
        uint64_t t0 = __builtin_readcyclecounter();                                      \
        for (int i = 0; i < iters; i++) { REP64(body) }                                  \
        uint64_t t1 = __builtin_readcyclecounter();                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3; \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                 \
    }
DEFK(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");)
DEFK(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 3, %1\n v_lshl_add_u64 %1, %1, 2, %2\n v_lshl_add_u64 %2, %2, 3, %3\n v_lshl_add_u64 %3, %3, 1, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(k_lshlrev_b64, asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshlrev_b64 %2, 7, %2\n v_lshlrev_b64 %3, 9, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(k_lshrrev_b64, asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 5, %1\n v_lshrrev_b64 %2, 7, %2\n v_lshrrev_b64 %3, 9, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(k_add_co_pair, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %2, vcc, %2, %6\n v_addc_co_u32 %3, vcc, %3, %7, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3) : "vcc");)
DEFK(k_alignbit, asm volatile("v_alignbit_b32 %0, %4, %0, 7\n v_alignbit_b32 %1, %5, %1, 9\n v_alignbit_b32 %2, %6, %2, 11\n v_alignbit_b32 %3, %7, %3, 13" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));)
DEFK(k_xor_b32, asm volatile("v_xor_b32 %0, %4, %0\n v_xor_b32 %1, %5, %1\n v_xor_b32 %2, %6, %2\n v_xor_b32 %3, %7, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));)
DEFK(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %4, %0\n v_mul_lo_u32 %1, %5, %1\n v_mul_lo_u32 %2, %6, %2\n v_mul_lo_u32 %3, %7, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));)
DEFK(k_lshl_add_u32, asm volatile("v_lshl_add_u32 %0, %4, 3, %0\n v_lshl_add_u32 %1, %5, 3, %1\n v_lshl_add_u32 %2, %6, 3, %2\n v_lshl_add_u32 %3, %7, 3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));)
DEFK(k_cmp_lt_u64, asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_u64 vcc, %2, %3\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "vcc");)
DEFK(k_add3_u32, asm volatile("v_add3_u32 %0, %4, %0, %1\n v_add3_u32 %1, %5, %1, %2\n v_add3_u32 %2, %6, %2, %3\n v_add3_u32 %3, %7, %3, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));)
DEFK(k_perm, asm volatile("v_perm_b32 %0, %4, %0, %1\n v_perm_b32 %1, %5, %1, %2\n v_perm_b32 %2, %6, %2, %3\n v_perm_b32 %3, %7, %3, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));)

template <class K>
void run(const char* name, K kern, int n_instr_per_rep) {
    uint64_t *out, *cyc;
    (void)hipMalloc(&out, 64 << 20);
    (void)hipMalloc(&cyc, 8 * 8192);
    const int iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8, tpb = 256;            // 8 blocks x 4 waves per CU = 8 waves per SIMD, whole chip
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(tpb), 0, 0, out, cyc, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(tpb), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * (tpb / 64) * iters * 64.0 * n_instr_per_rep;
    const double per_simd_per_s = wave_instr / (ms * 1e-3) / (256.0 * 4.0);
    printf("%-16s %8.3f ms  %.3e wave-instr/s/SIMD  = %.2f cycles per wave-instr at 2.4 GHz\n", name, ms, per_simd_per_s, 2.4e9 / per_simd_per_s);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run("v_xor_b32", k_xor_b32, 4);
    run("v_mad_u64_u32", k_mad_u64_u32, 4);
    run("v_lshl_add_u64", k_lshl_add_u64, 4);
    run("v_lshlrev_b64", k_lshlrev_b64, 4);
    run("v_lshrrev_b64", k_lshrrev_b64, 4);
    run("add_co+addc", k_add_co_pair, 4);
    run("v_alignbit_b32", k_alignbit, 4);
    run("v_mul_lo_u32", k_mul_lo_u32, 4);
    run("v_lshl_add_u32", k_lshl_add_u32, 4);
    run("cmp_lt_u64+cnd", k_cmp_lt_u64, 4);
    run("v_add3_u32", k_add3_u32, 4);
    run("v_perm_b32", k_perm, 4);
    return 0;
}
