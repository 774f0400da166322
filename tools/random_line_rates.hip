// random_line_rates.hip — round 5, VERDICT r04 #8 (the probe's granule): what the chip delivers for RANDOM reads of 64 / 32 / 16 bytes
// per lane from a 16 GiB array, at the probe's shape (one lane per probe, two probes in flight per lane, 1.9 M probes per launch = one
// sample table; and 15 M = eight tables).  If a 32-byte half-line came back at twice the rate of a 64-byte line, splitting the index
// lines would pay; if the rate is per REQUEST, it would not.  Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/random_line_rates.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

template <int BYTES>
__global__ __launch_bounds__(256) void probe_like(const uint4* __restrict__ lines, uint64_t n_lines, uint32_t n_probes, uint32_t salt, uint32_t* __restrict__ out) {
    constexpr int Q = BYTES / 16;                                   // 16-byte loads per probe
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_probes; i += gridDim.x * 256 * 2) {
        const uint32_t j = i + gridDim.x * 256;
        const uint64_t a = (mix(((uint64_t)salt << 32) | i) % n_lines) * 4, b = (mix(((uint64_t)salt << 32) | j) % n_lines) * 4;
        uint4 va[Q], vb[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) va[q] = lines[a + q];
#pragma unroll
        for (int q = 0; q < Q; q++) vb[q] = j < n_probes ? lines[b + q] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < Q; q++) acc += va[q].x ^ va[q].w ^ vb[q].y ^ vb[q].z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BYTES>
void run(const uint4* lines, uint64_t n_lines, uint32_t n_probes, uint32_t* out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(probe_like<BYTES>, dim3(1024), dim3(256), 0, 0, lines, n_lines, n_probes, (uint32_t)rep + 1, out);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep) best = ms < best ? ms : best;
    }
    printf("%2d B per probe, %9u probes per launch: %8.4f ms = %6.2f G probes/s = %7.1f GB/s of requested bytes\n", BYTES, n_probes, best, n_probes / (best * 1e-3) / 1e9,
           (double)n_probes * BYTES / (best * 1e-3) / 1e9);
}

int main() {
    const uint64_t bytes = 16ull << 30, n_lines = bytes / 64;
    uint4* lines;
    uint32_t* out;
    if (hipMalloc(&lines, bytes) != hipSuccess) { printf("no memory\n"); return 1; }
    (void)hipMalloc(&out, 64);
    (void)hipMemset(lines, 0x5a, bytes);
    (void)hipDeviceSynchronize();
    for (uint32_t n : {1882506u, 15060048u}) {
        run<64>(lines, n_lines, n, out);
        run<32>(lines, n_lines, n, out);
        run<16>(lines, n_lines, n, out);
    }
    return 0;
}
