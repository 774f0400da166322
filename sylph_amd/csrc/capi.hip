// capi.hip — context, error reporting and profiling hooks of the C ABI (include/sylph_hip.h).
#include <algorithm>
#include <memory>

#include <thread>

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
    if (!ctx->profile_only.empty()) {            // ",a,b," holds ",family,"?
        const size_t n = strlen(family);
        size_t at = 0;
        bool found = false;
        while ((at = ctx->profile_only.find(family, at)) != std::string::npos) {
            if (ctx->profile_only[at - 1] == ',' && ctx->profile_only[at + n] == ',') { found = true; break; }
            at += n;
        }
        if (!found) return;
    }
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

void* pool_acquire(sylph_ctx* ctx, size_t bytes, size_t* cap_out) {
    // size classes: 64 KiB granules below 8 MiB, 2 MiB granules above
    const size_t g = bytes < (8u << 20) ? (64u << 10) : (2u << 20);
    const size_t want = ((bytes + g - 1) / g) * g;
    int best = -1;
    for (size_t i = 0; i < ctx->pool_free.size(); i++) {
        const size_t c = ctx->pool_free[i].first;
        if (c >= want && c <= 2 * want + (4u << 20) && (best < 0 || c < ctx->pool_free[best].first)) best = (int)i;
    }
    if (best >= 0) {
        void* p = ctx->pool_free[best].second;
        *cap_out = ctx->pool_free[best].first;
        ctx->pool_free.erase(ctx->pool_free.begin() + best);
        return p;
    }
    void* p = nullptr;
    const double t0 = HostPhase::enabled() ? HostPhase::now() : 0;
    hipError_t e = hipMalloc(&p, want);
    if (HostPhase::enabled())
        fprintf(stderr, "[sylph_hip] pool miss: hipMalloc(%zu KiB) %.3f ms (pool holds %zu free blocks)\n", want >> 10,
                HostPhase::now() - t0, ctx->pool_free.size());
    if (e != hipSuccess) {   // give cached blocks back to the driver and retry once
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        for (auto& b : ctx->pool_free) (void)hipFree(b.second);
        ctx->pool_free.clear();
        SY_HIP(hipMalloc(&p, want));
    }
    ctx->pool_bytes += want;
    *cap_out = want;
    return p;
}

void pool_release(sylph_ctx* ctx, void* p, size_t cap) { ctx->pool_free.emplace_back(cap, p); }

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

void sylph_ctx::read_back(void* dst, const void* dev_src, size_t bytes) {
    SY_REQUIRE(bytes <= 4096, "read_back too large");
    static const bool slowlog = getenv("SYLPH_HIP_SLOWLOG") != nullptr;
    const double t0 = slowlog ? HostPhase::now() : 0;
    SY_HIP(hipMemcpyAsync(pinned, dev_src, bytes, hipMemcpyDeviceToHost, stream));
    const double t1 = slowlog ? HostPhase::now() : 0;
    SY_HIP(hipStreamSynchronize(stream));
    if (slowlog && HostPhase::now() - t0 > 3.0)
        fprintf(stderr, "[sylph_hip] slow read_back: issue %.3f ms, wait %.3f ms\n", t1 - t0, HostPhase::now() - t1);
    memcpy(dst, pinned, bytes);
    // every timing event recorded so far has completed: fold them into the totals and recycle the event objects
    // (creating fresh hipEvents per launch costs far more than the kernels being timed)
    if (!pending.empty()) sylph::profile_collect(this);
}

static void ensure_stage(sylph_ctx* c) {
    for (int i = 0; i < 2; i++) {
        if (!c->stage[i]) {
            SY_HIP(hipHostMalloc(&c->stage[i], sylph_ctx::STAGE_BYTES, hipHostMallocDefault));
            SY_HIP(hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
        }
    }
}

void sylph_ctx::d2h(void* dst, const void* dev_src, size_t bytes) {
    if (!bytes) { SY_HIP(hipStreamSynchronize(stream)); return; }
    ensure_stage(this);
    static const int mode = getenv("SYLPH_HIP_D2H_MODE") ? atoi(getenv("SYLPH_HIP_D2H_MODE")) : 0;
    if (mode == 1) {   // experiment: plain stream syncs, no events
        for (size_t done = 0; done < bytes;) {
            const size_t n = std::min(STAGE_BYTES, bytes - done);
            SY_HIP(hipMemcpyAsync(stage[0], (const char*)dev_src + done, n, hipMemcpyDeviceToHost, stream));
            SY_HIP(hipStreamSynchronize(stream));
            memcpy((char*)dst + done, stage[0], n);
            done += n;
        }
        return;
    }
    // ping-pong: the device fills one pinned buffer while the host drains the other
    size_t issued = 0, done = 0, len[2] = {0, 0};
    int head = 0, tail = 0, inflight = 0;
    while (done < bytes) {
        while (issued < bytes && inflight < 2) {
            len[tail] = std::min(STAGE_BYTES, bytes - issued);
            SY_HIP(hipMemcpyAsync(stage[tail], (const char*)dev_src + issued, len[tail], hipMemcpyDeviceToHost, stream));
            SY_HIP(hipEventRecord(stage_ev[tail], stream));
            issued += len[tail];
            tail ^= 1;
            inflight++;
        }
        SY_HIP(hipEventSynchronize(stage_ev[head]));
        memcpy((char*)dst + done, stage[head], len[head]);
        done += len[head];
        head ^= 1;
        inflight--;
    }
    if (!pending.empty()) sylph::profile_collect(this);
}

// A staging copy of 32 MiB on ONE thread runs at 10-20 GB/s out of the page cache — under what the link takes: four threads for the large ones
// (round 6: the 400 MB of a gzip pair's compressed bytes travelled in 16-19 ms; the helpers are only started for chunks of 8 MiB and more).
static void stage_copy(void* dst, const void* src, size_t n) {
    constexpr int T = 4;
    if (n < (8u << 20)) { memcpy(dst, src, n); return; }
    const size_t per = ((n / T) + 4095) & ~(size_t)4095;
    std::thread helpers[T - 1];
    int started = 0;
    try {
        for (int t = 1; t < T; t++) {
            const size_t a = (size_t)t * per;
            if (a >= n) break;
            helpers[started] = std::thread([=] { memcpy((char*)dst + a, (const char*)src + a, std::min(per, n - a)); });
            started++;
        }
    } catch (...) {}                                            // (no thread to be had: the rest of the chunk on this one)
    memcpy(dst, src, std::min(per, n));
    for (int t = started + 1; t < T; t++) { const size_t a = (size_t)t * per; if (a < n) memcpy((char*)dst + a, (const char*)src + a, std::min(per, n - a)); }
    for (int t = 0; t < started; t++) helpers[t].join();
}

void sylph_ctx::h2d(void* dev_dst, const void* src, size_t bytes) {
    if (!bytes) return;
    ensure_stage(this);
    size_t done = 0;
    int slot = 0;
    bool used[2] = {false, false};
    while (done < bytes) {
        const size_t n = std::min(STAGE_BYTES, bytes - done);
        if (used[slot]) SY_HIP(hipEventSynchronize(stage_ev[slot]));   // previous copy out of this buffer finished
        stage_copy(stage[slot], (const char*)src + done, n);
        SY_HIP(hipMemcpyAsync((char*)dev_dst + done, stage[slot], n, hipMemcpyHostToDevice, stream));
        SY_HIP(hipEventRecord(stage_ev[slot], stream));
        used[slot] = true;
        done += n;
        slot ^= 1;
    }
    // both staging buffers must be free again before another h2d/d2h reuses them
    for (int i = 0; i < 2; i++)
        if (used[i]) SY_HIP(hipEventSynchronize(stage_ev[i]));
}

extern "C" {

int sylph_version(void) { return 100; }   // 0.1.0
int sylph_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

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
        ctx->tmp_sort.ctx = ctx;
        ctx->counters.ctx = ctx;
        for (auto& b : ctx->scratch) b.ctx = ctx;
        try {
            SY_HIP(hipHostMalloc(&ctx->pinned, 4096, hipHostMallocDefault));
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
    if (ctx) sylph::ctx_unref(ctx);   // sessions / databases still alive keep the context (and its pool) alive
}

}  // extern "C"

void sylph::ctx_unref(sylph_ctx* ctx) {
    if (--ctx->refs > 0) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    ctx->tmp_sort.release();
    ctx->counters.release();
    for (auto& b : ctx->scratch) b.release();
    for (auto& b : ctx->pool_free) (void)hipFree(b.second);
    ctx->pool_free.clear();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (int i = 0; i < 2; i++) {
        if (ctx->stage[i]) (void)hipHostFree(ctx->stage[i]);
        if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (int i = 0; i < 2; i++) if (ctx->copy_ev[i]) (void)hipEventDestroy(ctx->copy_ev[i]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" {

int sylph_ctx_synchronize(sylph_ctx* ctx) {
    return guarded([&] {
        SY_REQUIRE(ctx, "null ctx");
        std::lock_guard<std::mutex> lock(ctx->mu);
        SY_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int sylph_pinned_alloc(uint64_t bytes, void** out) {
    return guarded([&] {
        SY_REQUIRE(out, "null argument");
        *out = nullptr;
        SY_HIP(hipHostMalloc(out, std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
    });
}

void sylph_pinned_free(void* p) {
    if (p) (void)hipHostFree(p);
}

}  // extern "C"

struct sylph_upload {
    sylph_ctx* ctx = nullptr;
    sylph::DevBuf dev;
    uint64_t bytes = 0, at = 0, chunk_cap = 0;
    void* chunk[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    int cur = -1;                            // the chunk handed out and not yet committed
    int next = 0;
    hipStream_t stream = nullptr;            // copies travel on a stream of their own (the context's stream may be busy with an index build)
};

extern "C" {

int sylph_upload_begin(sylph_ctx* ctx, uint64_t bytes, uint64_t chunk_bytes, sylph_upload** out) {
    return guarded([&] {
        SY_REQUIRE(ctx && out, "null argument");
        DeviceGuard dg(ctx->device);
        std::unique_ptr<sylph_upload> u(new sylph_upload());
        u->ctx = ctx;
        u->bytes = bytes;
        u->chunk_cap = std::max<uint64_t>(1u << 20, std::min<uint64_t>(chunk_bytes ? chunk_bytes : (128ull << 20), 1ull << 30));
        u->dev.alloc(bytes + 64);             // (plain hipMalloc: a buffer of this size does not belong in the context's pool)
        ctx->refs++;                          // taken BEFORE anything that may throw: sylph_upload_destroy below gives it back
        try {
            SY_HIP(hipStreamCreateWithFlags(&u->stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; i++) {
                SY_HIP(hipHostMalloc(&u->chunk[i], u->chunk_cap, hipHostMallocDefault));
                SY_HIP(hipEventCreateWithFlags(&u->ev[i], hipEventDisableTiming));
            }
        } catch (...) { sylph_upload_destroy(u.release()); throw; }
        *out = u.release();
    });
}

int sylph_upload_chunk(sylph_upload* u, void** chunk, uint64_t* cap) {
    return guarded([&] {
        SY_REQUIRE(u && chunk && cap, "null argument");
        SY_REQUIRE_STATE(u->cur < 0, "sylph_upload_chunk: the previous chunk has not been committed");
        DeviceGuard dg(u->ctx->device);
        const int i = u->next;
        if (u->used[i]) SY_HIP(hipEventSynchronize(u->ev[i]));      // the copy that last read this chunk
        u->cur = i;
        *chunk = u->chunk[i];
        *cap = u->chunk_cap;
    });
}

int sylph_upload_commit(sylph_upload* u, uint64_t n) {
    return guarded([&] {
        SY_REQUIRE(u, "null argument");
        SY_REQUIRE_STATE(u->cur >= 0, "sylph_upload_commit without a chunk");
        SY_REQUIRE(n <= u->chunk_cap && u->at + n <= u->bytes, "sylph_upload_commit: %llu bytes do not fit (%llu of %llu uploaded)",
                   (unsigned long long)n, (unsigned long long)u->at, (unsigned long long)u->bytes);
        DeviceGuard dg(u->ctx->device);
        const int i = u->cur;
        if (n) SY_HIP(hipMemcpyAsync(u->dev.as<char>() + u->at, u->chunk[i], n, hipMemcpyHostToDevice, u->stream));
        SY_HIP(hipEventRecord(u->ev[i], u->stream));
        u->used[i] = true;
        u->at += n;
        u->cur = -1;
        u->next = i ^ 1;
    });
}

int sylph_upload_finish(sylph_upload* u, const void** device_ptr) {
    return guarded([&] {
        SY_REQUIRE(u && device_ptr, "null argument");
        SY_REQUIRE_STATE(u->cur < 0, "sylph_upload_finish: a chunk is still out");
        SY_REQUIRE(u->at == u->bytes, "sylph_upload_finish: %llu of %llu bytes uploaded", (unsigned long long)u->at, (unsigned long long)u->bytes);
        DeviceGuard dg(u->ctx->device);
        SY_HIP(hipStreamSynchronize(u->stream));
        *device_ptr = u->dev.p;
    });
}

// the same uploader once more (a feed that sends one sample's text after the other): the page-locked chunks, the stream and — where
// it is large enough — the device buffer stay; what the previous round uploaded is no longer valid
int sylph_upload_restart(sylph_upload* u, uint64_t bytes) {
    return guarded([&] {
        SY_REQUIRE(u, "null argument");
        SY_REQUIRE_STATE(u->cur < 0, "sylph_upload_restart: a chunk is still out");
        DeviceGuard dg(u->ctx->device);
        SY_HIP(hipStreamSynchronize(u->stream));
        // grows when it must, and SHRINKS behind an unusually large sample (round 6): a buffer that only grew stayed at the
        // size of the largest text an engine ever sent — up to ~18 GB each, times the engines of a `profile`, beside a 29 GB index
        const bool too_small = bytes + 64 > u->dev.cap;
        const bool wasteful = u->dev.cap > (4ull << 30) && u->dev.cap / 4 > bytes + 64;
        if (too_small || wasteful) {
            SY_HIP(hipStreamSynchronize(u->ctx->stream));      // kernels of the context may still read the old buffer
            u->dev.release();
            u->dev.alloc(bytes + bytes / 8 + 64);
        }
        u->bytes = bytes;
        u->at = 0;
        u->used[0] = u->used[1] = false;
        u->next = 0;
    });
}

void sylph_upload_destroy(sylph_upload* u) {
    if (!u) return;
    (void)hipSetDevice(u->ctx->device);
    if (u->stream) { (void)hipStreamSynchronize(u->stream); (void)hipStreamDestroy(u->stream); }
    for (int i = 0; i < 2; i++) {
        if (u->chunk[i]) (void)hipHostFree(u->chunk[i]);
        if (u->ev[i]) (void)hipEventDestroy(u->ev[i]);
    }
    u->dev.release();
    sylph_ctx* ctx = u->ctx;
    delete u;
    sylph::ctx_unref(ctx);
}

int sylph_ctx_set_option(sylph_ctx* ctx, const char* key, const char* value) {
    return guarded([&] {
        SY_REQUIRE(ctx && key && value, "null argument");
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (!strcmp(key, "finish")) {
            if (!strcmp(value, "auto")) ctx->finish_mode = 0;
            else if (!strcmp(value, "generic")) ctx->finish_mode = 1;
            else if (!strcmp(value, "bucket")) ctx->finish_mode = 2;
            else SY_REQUIRE(false, "finish must be auto|generic|bucket");
        } else if (!strcmp(key, "seeds")) {
            if (!strcmp(value, "auto") || !strcmp(value, "ordered")) ctx->seeds_mode = 0;
            else if (!strcmp(value, "unordered")) ctx->seeds_mode = 1;
            else if (!strcmp(value, "slots")) ctx->seeds_mode = 2;
            else SY_REQUIRE(false, "seeds must be auto|slots|unordered");
        } else if (!strcmp(key, "bucket_target")) {
            const long v = strtol(value, nullptr, 10);
            SY_REQUIRE(v >= 16 && v <= 256, "bucket_target must be in [16, 256]");
            ctx->bucket_target = (uint32_t)v;
        } else if (!strcmp(key, "plain_records")) {
            ctx->plain_records = (uint32_t)strtol(value, nullptr, 10) ? 1u : 0u;   // A/B knob: 0 = occurrence records for every batch
        } else if (!strcmp(key, "shard_reduce")) {
            if (!strcmp(value, "alltoall")) ctx->shard_reduce = 0;
            else if (!strcmp(value, "allgather")) ctx->shard_reduce = 1;
            else SY_REQUIRE(false, "shard_reduce must be alltoall|allgather");
        } else if (!strcmp(key, "stream_priority")) {
            // The context's own stream is made again at another priority (hipStreamCreateWithPriority): "high" lets the small kernels of a
            // profile context through where the sketch contexts' seeding kernels keep the chip full (pipeline.hip).
            SY_REQUIRE(ctx->own_stream, "stream_priority: the context runs on the caller's stream");
            int least = 0, greatest = 0;
            DeviceGuard dg(ctx->device);
            SY_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            int prio = 0;
            if (!strcmp(value, "high")) prio = greatest;
            else if (!strcmp(value, "low")) prio = least;
            else SY_REQUIRE(!strcmp(value, "normal"), "stream_priority must be high|normal|low");
            SY_HIP(hipStreamSynchronize(ctx->stream));
            hipStream_t s = nullptr;
            SY_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
            (void)hipStreamDestroy(ctx->stream);
            ctx->stream = s;
        } else if (!strcmp(key, "fail_next_peer_copy")) {
            ctx->fail_next_peer_copy = (uint32_t)strtol(value, nullptr, 10);     // tests only: sylph_db_replicate into this context takes the host road
        } else if (!strcmp(key, "fail_next_shard_probe")) {
            ctx->fail_next_shard_probe = (uint32_t)strtol(value, nullptr, 10);   // tests only: one rank of a sharded batch fails between the collectives
        } else if (!strcmp(key, "push_chunk_bytes")) {
            const long long v = strtoll(value, nullptr, 10);
            SY_REQUIRE(v >= 64 && v <= (1ll << 31), "push_chunk_bytes must be in [64, 2^31]");
            ctx->push_chunk_bytes = (uint64_t)v;
        } else if (!strcmp(key, "reads_wg_per_cu")) {
            const long v = strtol(value, nullptr, 10);
            SY_REQUIRE(v >= 0 && v <= 64, "reads_wg_per_cu must be in [0, 64]");
            ctx->reads_wg_per_cu = (uint32_t)v;
        } else if (!strcmp(key, "profile_only")) {
            // the families sylph_ctx_profile times from now on: "seeds" or "seeds,probe"; "" or "all" = every family.  (Every timed family
            // costs two event records per launch group on the stream: bench.py keeps only the dominant kernel's in its timed region.)
            ctx->profile_only = (!value[0] || !strcmp(value, "all")) ? std::string() : "," + std::string(value) + ",";
        } else if (!strcmp(key, "reads_tail_pct")) {
            const long v = strtol(value, nullptr, 10);
            SY_REQUIRE(v >= 0 && v <= 50, "reads_tail_pct must be in [0, 50]");
            ctx->reads_tail_pct = (uint32_t)v;
        } else if (!strcmp(key, "reads_hash")) {
            const long v = strtol(value, nullptr, 10);
            SY_REQUIRE(v >= -1 && v <= 2, "reads_hash must be -1 (default), 0, 1 or 2");
            ctx->reads_hash = (int)v;
        } else if (!strcmp(key, "reads_slack")) {
            ctx->reads_slack = (uint32_t)strtoul(value, nullptr, 0);
        } else if (!strcmp(key, "index_lambda")) {
            const long v = strtol(value, nullptr, 10);
            SY_REQUIRE(v >= 1 && v <= 8, "index_lambda must be in [1, 8]");
            ctx->index_lambda = (uint32_t)v;
        } else if (!strcmp(key, "cu_mask")) {
            // "lo:hi" — the context's OWN stream is recreated on the compute units [lo, hi) of the device's CU-mask numbering (the
            // driver deals consecutive mask bits to the XCDs in turn, so a range of 8 n bits is n CUs on every XCD); "all" undoes it.
            // An A/B knob for the pipeline (seeding on one part of the chip, the profile stream on the rest): profiles/r04_ab_cu_mask.txt.
            SY_REQUIRE(ctx->own_stream, "cu_mask: the context runs on a caller-supplied stream");
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
            long lo = 0, hi = cus;
            if (strcmp(value, "all")) {
                char* end = nullptr;
                lo = strtol(value, &end, 10);
                SY_REQUIRE(end && *end == ':', "cu_mask must be lo:hi or all");
                hi = strtol(end + 1, nullptr, 10);
            }
            SY_REQUIRE(lo >= 0 && lo < hi && hi <= cus, "cu_mask range [%ld, %ld) outside the device's %d CUs", lo, hi, cus);
            DeviceGuard dg(ctx->device);
            SY_HIP(hipStreamSynchronize(ctx->stream));
            std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
            for (long i = lo; i < hi; i++) mask[(size_t)i >> 5] |= 1u << (i & 31);
            hipStream_t ns = nullptr;
            SY_HIP(hipExtStreamCreateWithCUMask(&ns, (uint32_t)mask.size(), mask.data()));
            (void)hipStreamDestroy(ctx->stream);
            ctx->stream = ns;
        } else if (!strcmp(key, "index_pass_max")) {
            const long long v = strtoll(value, nullptr, 10);
            SY_REQUIRE(v >= 1 && v <= (1ll << 31), "index_pass_max must be in [1, 2^31]");
            ctx->index_pass_max = (uint64_t)v;
        } else {
            SY_REQUIRE(false, "unknown option %s", key);
        }
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
