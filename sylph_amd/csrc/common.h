// common.h — shared host-side plumbing for the gfx950 sketch/profile engine (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sylph_hip.h"

namespace sylph {

void set_error(const char* fmt, ...);

struct HipError { hipError_t e; const char* what; const char* file; int line; };

#define SY_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) throw ::sylph::HipError{_e, #expr, __FILE__, __LINE__};    \
    } while (0)

struct ArgError { std::string msg; };
#define SY_REQUIRE(cond, ...)                                     \
    do {                                                          \
        if (!(cond)) {                                            \
            char _b[512];                                         \
            snprintf(_b, sizeof(_b), __VA_ARGS__);                \
            throw ::sylph::ArgError{_b};                          \
        }                                                         \
    } while (0)

// Translate C++ exceptions into ABI status codes; nothing unwinds across extern "C".
template <class F>
int guarded(F&& f) {
    try {
        f();
        return SYLPH_OK;
    } catch (const HipError& e) {
        set_error("HIP error %d (%s) at %s:%d: %s", (int)e.e, hipGetErrorString(e.e), e.file, e.line, e.what);
        (void)hipGetLastError();
        return e.e == hipErrorOutOfMemory ? SYLPH_ERR_NOMEM : SYLPH_ERR_HIP;
    } catch (const ArgError& e) {
        set_error("%s", e.msg.c_str());
        return SYLPH_ERR_INVALID;
    } catch (const std::bad_alloc&) {
        set_error("host allocation failed");
        return SYLPH_ERR_NOMEM;
    } catch (...) {
        set_error("unknown internal error");
        return SYLPH_ERR_INVALID;
    }
}

// Grow-only device buffer.  Sketch/DB state lives in these for the lifetime of a session; HBM is 288 GB, so
// capacity is doubled rather than trimmed.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // Ensure capacity >= bytes; contents are NOT preserved.
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        release();
        size_t want = bytes + bytes / 4 + 256;
        SY_HIP(hipMalloc(&p, want));
        cap = want;
    }
    // Ensure capacity >= bytes keeping the first `keep` bytes (device-to-device copy on `s`).
    void grow_keep(size_t bytes, size_t keep, hipStream_t s) {
        if (bytes <= cap) return;
        size_t want = bytes * 2 + 256;
        void* np = nullptr;
        SY_HIP(hipMalloc(&np, want));
        if (keep && p) {
            SY_HIP(hipMemcpyAsync(np, p, keep, hipMemcpyDeviceToDevice, s));
            SY_HIP(hipStreamSynchronize(s));
        }
        if (p) (void)hipFree(p);
        p = np;
        cap = want;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct KernelStat { double ms = 0; uint64_t launches = 0; };

}  // namespace sylph

struct sylph_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mu;                          // serialises calls on this ctx
    // profiling
    bool profile = false;
    std::map<std::string, sylph::KernelStat> stats;
    struct Pending { std::string fam; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    // scratch
    sylph::DevBuf tmp_sort;                 // rocPRIM temporary storage
    sylph::DevBuf scratch[8];
    sylph::DevBuf counters;                 // small device words (survivor counters etc.)
    void* pinned = nullptr;                 // small pinned host mirror for counters
};

namespace sylph {

// RAII timer: when ctx->profile is on, brackets the enclosed launches with a hipEvent pair on ctx->stream.
struct ScopedKernelTimer {
    sylph_ctx* ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char* fam;
    ScopedKernelTimer(sylph_ctx* c, const char* family);
    ~ScopedKernelTimer();
};
void profile_collect(sylph_ctx* ctx);       // resolves pending event pairs into ctx->stats (synchronises them)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) SY_HIP(hipSetDevice(dev));
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- device primitives implemented with rocPRIM in prims.hip (kept in one TU: rocPRIM is slow to compile) ----
void sort_pairs_u64_u32(sylph_ctx* ctx, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout,
                        size_t n, int begin_bit, int end_bit);
void sort_pairs_u32_u64(sylph_ctx* ctx, const uint32_t* kin, uint32_t* kout, const uint64_t* vin, uint64_t* vout,
                        size_t n, int begin_bit, int end_bit);
void sort_keys_u64(sylph_ctx* ctx, const uint64_t* kin, uint64_t* kout, size_t n, int begin_bit, int end_bit);
void exclusive_sum_u32(sylph_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n);   // out[i] = sum in[0..i)
void inclusive_max_u32(sylph_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n);

inline int bit_length(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

}  // namespace sylph
