// common.h — shared host-side plumbing for the gfx950 sketch/profile engine (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sylph_hip.h"

struct sylph_ctx;

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

struct StateError { std::string msg; };   // -> SYLPH_ERR_STATE
#define SY_REQUIRE_STATE(cond, ...)                               \
    do {                                                          \
        if (!(cond)) {                                            \
            char _b[512];                                         \
            snprintf(_b, sizeof(_b), __VA_ARGS__);                \
            throw ::sylph::StateError{_b};                        \
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
    } catch (const StateError& e) {
        set_error("%s", e.msg.c_str());
        return SYLPH_ERR_STATE;
    } catch (const std::bad_alloc&) {
        set_error("host allocation failed");
        return SYLPH_ERR_NOMEM;
    } catch (...) {
        set_error("unknown internal error");
        return SYLPH_ERR_INVALID;
    }
}

// Device memory is recycled through a per-context pool: hipMalloc/hipFree cost 0.1-1 ms each and hipFree
// synchronises the device, which at ~20 allocations per sample would dwarf the kernels (the whole 1 Gbp sketch is
// ~5 ms of GPU time).  Blocks return to the pool on release and are handed out again to any request they fit;
// everything runs on the one ctx stream, so stream order makes the reuse safe.  HBM is 288 GB: capacity is
// rounded up generously rather than trimmed.
void* pool_acquire(sylph_ctx* ctx, size_t bytes, size_t* cap_out);
void pool_release(sylph_ctx* ctx, void* p, size_t cap);

// Grow-only device buffer backed by the ctx pool (or plain hipMalloc when ctx == nullptr).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    sylph_ctx* ctx = nullptr;
    DevBuf() = default;
    explicit DevBuf(sylph_ctx* c) : ctx(c) {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            if (ctx) pool_release(ctx, p, cap);
            else (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
    }
    void alloc(size_t want) {
        if (ctx) p = pool_acquire(ctx, want, &cap);
        else { SY_HIP(hipMalloc(&p, want)); cap = want; }
    }
    // Ensure capacity >= bytes; contents are NOT preserved.
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        release();
        alloc(bytes + bytes / 4 + 256);
    }
    // Ensure capacity >= bytes keeping the first `keep` bytes (device-to-device copy on `s`).
    void grow_keep(size_t bytes, size_t keep, hipStream_t s) {
        if (bytes <= cap) return;
        void* old = p;
        const size_t old_cap = cap;
        p = nullptr;
        cap = 0;
        alloc(bytes * 2 + 256);
        if (keep && old) SY_HIP(hipMemcpyAsync(p, old, keep, hipMemcpyDeviceToDevice, s));
        if (old) {
            if (ctx) pool_release(ctx, old, old_cap);   // stream-ordered: the copy above is queued before any reuse
            else { SY_HIP(hipStreamSynchronize(s)); (void)hipFree(old); }
        }
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct KernelStat { double ms = 0; uint64_t launches = 0; };

}  // namespace sylph

#ifndef SYLPH_READS_TAIL_PCT
#define SYLPH_READS_TAIL_PCT 10     // (round 6; 0 until then: profiles/r06_ab_latency.txt — 5, 15, 20, 30 are all slower on the exact pair set)
#endif
struct sylph_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mu;                          // serialises calls on this ctx
    int finish_mode = 0;                    // 0 auto, 1 generic, 2 bucket-only (sylph_ctx_set_option "finish")
    uint32_t bucket_target = 128;           // mean occurrences per replay bucket aimed for ("bucket_target")
    uint32_t plain_records = 1;             // marker-less single-end batches keep no occurrence records ("plain_records" = 0: always write them)
    uint32_t shard_reduce = 0;              // "shard_reduce": 0 = the sharded batch's hits travel by all-to-all (default), 1 = by ONE all-gather (north_star's wording; shard_plan.h plan_hits_gather)
    uint32_t fail_next_peer_copy = 0;       // fault injection for the tests ("fail_next_peer_copy"): the next sylph_db_replicate INTO this context finds no device-to-device road
    uint32_t fail_next_shard_probe = 0;     // fault injection for the tests ("fail_next_shard_probe"): the next sharded probe on this context throws
    uint32_t index_lambda = 0;              // 0 = the default for the line size (contain_index.h DEFAULT_INDEX_LAMBDA: 4 per 64-byte line); postings per bucket line of a database index aimed for ("index_lambda"; r04: 3 -> 4, 38.5 -> 29.1 GB at GTDB scale for +3 % probe time alone)
    uint64_t index_pass_max = 1ull << 30;   // postings sorted per pass of the index build ("index_pass_max"; tests lower it)
    // "One seeding kernel at a time" among the contexts of a pipeline (pipeline.hip arms this before a push, under its seed_mu): the
    // stream waits for `gate` — the event behind the previous sample's seeding kernel — immediately before ITS seeding kernel, not before
    // the bookkeeping kernels in front of it, and records `done` immediately behind it: what lies between two samples' seeding kernels
    // on the GPU is one event and one wait (r05: it was the record-lookup kernel and three more event packets, 50-60 us per sample).
    struct SeedTurn { hipEvent_t gate = nullptr, done = nullptr; bool recorded = false; } turn;
    void seed_gate() { if (turn.gate) { (void)hipStreamWaitEvent(stream, turn.gate, 0); turn.gate = nullptr; } }
    void seed_done() { if (turn.done && !turn.recorded && hipEventRecord(turn.done, stream) == hipSuccess) turn.recorded = true; }
    std::string profile_only;                 // "profile_only": comma-separated families the kernel timers are limited to ("" = all of them)
    uint32_t reads_tail_pct = SYLPH_READS_TAIL_PCT;   // "reads_tail_pct": share of a sample's blocks launched behind its turn's event (pipelines only; 0 = one launch)
    int reads_hash = -1;                      // "reads_hash": the read kernel's hash / threshold spelling, -1 = the build's default (reads.hip)
    uint32_t reads_slack = 0;                 // "reads_slack": tests only — widens the high-word candidate test of reads_hash = 2
    uint32_t reads_wg_per_cu = 0;             // "reads_wg_per_cu": 0 = one workgroup per block of reads (measured best: 0.75 ms vs 0.84 ms with 8 looping workgroups per CU), n = n looping workgroups per CU
    uint64_t push_chunk_bytes = 64ull << 20;  // bytes of bases per chunk of a host batch ("push_chunk_bytes"; tests lower it)
    int seeds_mode = 0;                     // 0 auto: read-per-lane kernel for short reads, else ordered slots; 1 unordered kernel + radix sort; 2 ordered slots only ("seeds")
    std::atomic<int> refs{1};               // the creator + every live session / db; freed when it drops to 0
    // profiling
    bool profile = false;
    std::map<std::string, sylph::KernelStat> stats;
    struct Pending { std::string fam; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    // device memory pool (free blocks), see pool_acquire
    std::vector<std::pair<size_t, void*>> pool_free;
    size_t pool_bytes = 0;
    // scratch
    sylph::DevBuf tmp_sort;                 // rocPRIM temporary storage
    sylph::DevBuf scratch[8];
    sylph::DevBuf counters;                 // small device words (survivor counters etc.)
    void* pinned = nullptr;                 // 4 KiB pinned host page for small read-backs
    // small synchronous device->host read through the pinned page (pageable D2H copies are staged and slow)
    void read_back(void* dst, const void* dev_src, size_t bytes);
    // Bulk transfers between caller-owned PAGEABLE host memory and the device always go through two library-owned
    // pinned staging buffers.  Handing pageable pointers to hipMemcpyAsync makes the runtime pin the caller's pages;
    // when the caller then frees them (munmap) the driver evicts this process's GPU queues for ~20 ms.
    static constexpr size_t STAGE_BYTES = 32u << 20;
    void* stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;       // host batches travel on this stream while ctx->stream computes (sketch.hip)
    hipEvent_t copy_ev[2] = {nullptr, nullptr};
    void d2h(void* dst, const void* dev_src, size_t bytes);     // synchronous on return
    void h2d(void* dev_dst, const void* src, size_t bytes);     // queued on `stream`; src may be reused on return
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
void profile_collect(sylph_ctx* ctx);
void ctx_unref(sylph_ctx* ctx);             // drops one reference; destroys the ctx at zero       // resolves pending event pairs into ctx->stats (synchronises them)

// SYLPH_HIP_TRACE=1: print the wall time of each host-side phase (stream-synchronised) to stderr.
struct HostPhase {
    sylph_ctx* ctx;
    const char* name;
    double t0 = 0;
    static bool enabled() { static const bool e = getenv("SYLPH_HIP_TRACE") != nullptr; return e; }
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    HostPhase(sylph_ctx* c, const char* n) : ctx(c), name(n) { if (enabled()) { (void)hipStreamSynchronize(c->stream); t0 = now(); } }
    ~HostPhase() { if (enabled()) { (void)hipStreamSynchronize(ctx->stream); fprintf(stderr, "[sylph_hip] %-28s %8.3f ms\n", name, now() - t0); } }
};

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
void sort_pairs_u32_u32(sylph_ctx* ctx, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                        size_t n, int begin_bit, int end_bit);
// Ordered K1: tiles whose survivors did not fit their slots (low-complexity reads, tandem repeats).  Up to SPILL_MAX_TILES
// of them are redone into full-size spill regions; more than that sends the batch to the unordered kernel + radix sort.
constexpr uint32_t SPILL_MAX_TILES = 256;
struct SpillState { uint32_t n_tiles; uint32_t tiles[SPILL_MAX_TILES]; };
struct ReadsState { uint32_t long_record; SpillState spill; };   // reads.hip: a record too long for the short-read kernel was seen

void sort_keys_u32(sylph_ctx* ctx, const uint32_t* kin, uint32_t* kout, size_t n, int begin_bit, int end_bit);
void sort_keys_u64(sylph_ctx* ctx, const uint64_t* kin, uint64_t* kout, size_t n, int begin_bit, int end_bit);
void exclusive_sum_u32(sylph_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n);   // out[i] = sum in[0..i)
void inclusive_max_u32(sylph_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n);

inline int bit_length(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

}  // namespace sylph
