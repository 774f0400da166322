// capi.hip — context, error reporting and profiling hooks of the C ABI (include/sylph_hip.h).
#include "common.h"

namespace sylph {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

ScopedKernelTimer::ScopedKernelTimer(sylph_ctx* c, const char* family) : ctx(c), fam(family) {
    if (!ctx->profile) return;
    auto get = [&]() {
        hipEvent_t e;
        if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        return e;
    };
    a = get();
    b = get();
    if (a) (void)hipEventRecord(a, ctx->stream);
}

ScopedKernelTimer::~ScopedKernelTimer() {
    if (!a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->pending.push_back({fam, a, b});
}

void profile_collect(sylph_ctx* ctx) {
    for (auto& p : ctx->pending) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto& s = ctx->stats[p.fam];
            s.ms += ms;
            s.launches++;
        }
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->pending.clear();
}

}  // namespace sylph

using namespace sylph;

extern "C" {

int sylph_version(void) { return 100; }   // 0.1.0

const char* sylph_last_error(void) { return g_err; }

void sylph_free(void* p) { free(p); }

int sylph_ctx_create(int device, void* stream, sylph_ctx** out) {
    return guarded([&] {
        SY_REQUIRE(out, "null argument");
        int ndev = 0;
        SY_HIP(hipGetDeviceCount(&ndev));
        SY_REQUIRE(ndev > 0, "no HIP device visible");
        if (device < 0) SY_HIP(hipGetDevice(&device));
        SY_REQUIRE(device < ndev, "device %d out of range (have %d)", device, ndev);
        hipDeviceProp_t prop;
        SY_HIP(hipGetDeviceProperties(&prop, device));
        SY_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0 || getenv("SYLPH_HIP_ALLOW_ANY_ARCH"),
                   "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        DeviceGuard dg(device);
        sylph_ctx* ctx = new sylph_ctx();
        ctx->device = device;
        try {
            if (stream) ctx->stream = (hipStream_t)stream;
            else {
                SY_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
                ctx->own_stream = true;
            }
            ctx->counters.reserve(64);
        } catch (...) { delete ctx; throw; }
        *out = ctx;
    });
}

void sylph_ctx_destroy(sylph_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int sylph_ctx_synchronize(sylph_ctx* ctx) {
    return guarded([&] {
        SY_REQUIRE(ctx, "null ctx");
        std::lock_guard<std::mutex> lock(ctx->mu);
        SY_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int sylph_ctx_profile(sylph_ctx* ctx, int enable) {
    return guarded([&] {
        SY_REQUIRE(ctx, "null ctx");
        std::lock_guard<std::mutex> lock(ctx->mu);
        profile_collect(ctx);
        ctx->stats.clear();
        ctx->profile = enable != 0;
    });
}

int sylph_ctx_kernel_stats(sylph_ctx* ctx, const char* family, double* total_ms, uint64_t* launches) {
    return guarded([&] {
        SY_REQUIRE(ctx && family, "null argument");
        std::lock_guard<std::mutex> lock(ctx->mu);
        profile_collect(ctx);
        auto it = ctx->stats.find(family);
        if (total_ms) *total_ms = it == ctx->stats.end() ? 0.0 : it->second.ms;
        if (launches) *launches = it == ctx->stats.end() ? 0 : it->second.launches;
    });
}

}  // extern "C"
