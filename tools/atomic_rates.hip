// Microbenchmark: returning global atomicAdd on N counters from 4M uniformly random lanes (the bucket histogram / cursor
// pattern of the partition in replay_lds.hip).  Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/atomic_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void hist(uint32_t* cnt, uint32_t n_cnt, uint32_t n, uint32_t* sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = (uint32_t)(mix(i) % n_cnt);
    const uint32_t r = atomicAdd(&cnt[b], 1u);
    if (r == 0xFFFFFFFFu) *sink = r;
}
__global__ void hist_noret(uint32_t* cnt, uint32_t n_cnt, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicAdd(&cnt[(uint32_t)(mix(i) % n_cnt)], 1u);
}
__global__ void scatter(uint32_t* cnt, uint32_t n_cnt, uint32_t n, uint32_t* out, uint32_t cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = (uint32_t)(mix(i) % n_cnt);
    const uint32_t r = atomicAdd(&cnt[b], 1u);
    out[(uint64_t)b * cap + (r & (cap - 1))] = i;
}
struct Rec32 { uint64_t a, b, c, d; };
__global__ void scatter32(uint32_t* cnt, uint32_t n_cnt, uint32_t n, Rec32* out, uint32_t cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = mix(i);
    const uint32_t b = (uint32_t)(h % n_cnt);
    const uint32_t r = atomicAdd(&cnt[b], 1u);
    out[(uint64_t)b * cap + (r & (cap - 1))] = Rec32{h, i, h ^ i, h + i};
}
int main() {
    const uint32_t n = 4u << 20;
    uint32_t *cnt, *sink, *out;
    hipMalloc(&cnt, 1u << 24); hipMalloc(&sink, 4); hipMalloc(&out, (size_t)(1u << 16) * 512 * 4);
    Rec32* out32; hipMalloc(&out32, (size_t)(1u << 16) * 256 * 32);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint32_t n_cnt : {1u, 64u, 1024u, 8192u, 32768u, 65536u, 1u << 20}) {
        float ms[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(cnt, 0, 1u << 24);
            hipEventRecord(a); hist<<<n / 256, 256>>>(cnt, n_cnt, n, sink); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms[0], a, b);
            hipEventRecord(a); hist_noret<<<n / 256, 256>>>(cnt, n_cnt, n); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms[1], a, b);
            if (n_cnt <= 65536) {
                hipMemset(cnt, 0, 1u << 24);
                hipEventRecord(a); scatter<<<n / 256, 256>>>(cnt, n_cnt, n, out, 512); hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms[2], a, b);
                hipMemset(cnt, 0, 1u << 24);
                hipEventRecord(a); scatter32<<<n / 256, 256>>>(cnt, n_cnt, n, out32, 256); hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms[3], a, b);
            }
        }
        printf("counters %8u: returning %.3f ms  non-returning %.3f ms  returning+scatter %.3f ms  returning+32 B record scatter %.3f ms  (4M lanes)\n", n_cnt, ms[0], ms[1], ms[2], ms[3]);
    }
    return 0;
}
