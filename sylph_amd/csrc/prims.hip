// prims.hip — device-wide sort / scan primitives (rocPRIM, AMD's native primitive library) behind thin wrappers.
// These move O(seeds) = O(bases / c) data; the hand-written kernels (seeds.hip, replay.hip, contain.hip) move
// O(bases) and O(probes).  Kept in one translation unit because rocPRIM headers dominate compile time.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace sylph {

template <class F>
static void with_temp(sylph_ctx* ctx, F&& call) {
    size_t bytes = 0;
    SY_HIP(call(nullptr, bytes));
    ctx->tmp_sort.reserve(bytes ? bytes : 16);
    SY_HIP(call(ctx->tmp_sort.p, bytes));
}

void sort_pairs_u64_u32(sylph_ctx* ctx, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout,
                        size_t n, int begin_bit, int end_bit) {
    if (n == 0) return;
    ScopedKernelTimer t(ctx, "sort");
    with_temp(ctx, [&](void* tmp, size_t& bytes) {
        return rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit,
                                         ctx->stream);
    });
}

void sort_pairs_u32_u64(sylph_ctx* ctx, const uint32_t* kin, uint32_t* kout, const uint64_t* vin, uint64_t* vout,
                        size_t n, int begin_bit, int end_bit) {
    if (n == 0) return;
    ScopedKernelTimer t(ctx, "sort");
    with_temp(ctx, [&](void* tmp, size_t& bytes) {
        return rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit,
                                         ctx->stream);
    });
}

void sort_pairs_u32_u32(sylph_ctx* ctx, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                        size_t n, int begin_bit, int end_bit) {
    if (n == 0) return;
    ScopedKernelTimer t(ctx, "sort");
    with_temp(ctx, [&](void* tmp, size_t& bytes) {
        return rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit,
                                         ctx->stream);
    });
}

void sort_keys_u32(sylph_ctx* ctx, const uint32_t* kin, uint32_t* kout, size_t n, int begin_bit, int end_bit) {
    if (n == 0) return;
    ScopedKernelTimer t(ctx, "sort");
    with_temp(ctx, [&](void* tmp, size_t& bytes) {
        return rocprim::radix_sort_keys(tmp, bytes, kin, kout, n, (unsigned)begin_bit, (unsigned)end_bit, ctx->stream);
    });
}

void sort_keys_u64(sylph_ctx* ctx, const uint64_t* kin, uint64_t* kout, size_t n, int begin_bit, int end_bit) {
    if (n == 0) return;
    ScopedKernelTimer t(ctx, "sort");
    with_temp(ctx, [&](void* tmp, size_t& bytes) {
        return rocprim::radix_sort_keys(tmp, bytes, kin, kout, n, (unsigned)begin_bit, (unsigned)end_bit, ctx->stream);
    });
}

void exclusive_sum_u32(sylph_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n) {
    if (n == 0) return;
    with_temp(ctx, [&](void* tmp, size_t& bytes) {
        return rocprim::exclusive_scan(tmp, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), ctx->stream);
    });
}

void inclusive_max_u32(sylph_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n) {
    if (n == 0) return;
    with_temp(ctx, [&](void* tmp, size_t& bytes) {
        return rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::maximum<uint32_t>(), ctx->stream);
    });
}

}  // namespace sylph
