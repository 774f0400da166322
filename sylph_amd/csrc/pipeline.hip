// pipeline.hip — a stream of samples through both stages, overlapped inside the library.
//
// The reference runs its samples on the rayon pool: sketch.rs:313,371 sketches files on parallel workers, contain.rs:267-289
// walks sample chunks x genomes on all threads.  On the GPU the same overlap has to come from several HIP streams: the seeding
// kernel is VALU-bound and leaves HBM idle, the dedup/count and profile stages are strings of small memory- or latency-bound
// dispatches — with the samples of several steps in flight the small kernels of one sample run beside the seeding kernel of
// another, and result copies and host-side assembly disappear behind them.  Rounds 1-2 got that overlap from Python threads in
// bench.py; this file moves it behind the C ABI so that a Rust (or any) host gets it with three calls:
//
//   sylph_pipeline_submit*()   hand over a sample (batches of records, or a session the caller has been pushing into)
//   sylph_pipeline_next()      the oldest outstanding sample's results: table size, containment counts, coverage vectors
//
// Inside: `n_workers` sketch threads, each with a context of its own (stream + memory pool), take samples in submission
// order (begin -> push -> finish_device); ONE profile thread owns the database's context, waits for the oldest sketched sample
// and probes it TOGETHER with every consecutive sample that is ready by then (up to `max_batch` tables per launch: one sample
// per launch leaves the probe at 22 % of the HBM roofline, eight reach 30 %), into a pinned result block of the pipeline.
// Results come back in submission order, so nothing the caller sees depends on the interleaving of the threads.
// With a sharded database (cfg.comm) the batches must be the same on every rank: the profile thread then waits for exactly
// `max_batch` samples (or sylph_pipeline_flush) and runs the exchange of shard.hip.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <thread>

#include "contain_index.h"
#include "sketch_session.h"

using namespace sylph;

namespace {

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

struct Block {   // one pinned result block (+ its layout), shared by the samples of one probe batch
    HostBlock mem;
    uint32_t width = 4, users = 0;
    uint64_t n_covs = 0;
};

enum class JobState { Queued, Sketching, Sketched, Profiled };

struct Job {
    uint64_t seq = 0, tag = 0;
    std::vector<sylph_read_batch> batches;
    int mem = SYLPH_MEM_DEVICE, enc = SYLPH_ENC_ASCII;
    sylph_sketch* sk = nullptr;
    bool adopted = false;        // the caller's session (sylph_pipeline_submit_session)
    int worker = -1;             // the worker whose context owns `sk` (adopted: none)
    JobState state = JobState::Queued;
    int status = SYLPH_OK;
    std::string error;
    const uint64_t* dev_k = nullptr;
    const uint32_t* dev_c = nullptr;
    uint64_t n_table = 0, dup_removed = 0;
    std::vector<uint64_t> host_k;    // want_table
    std::vector<uint32_t> host_c;
    Block* block = nullptr;
    uint32_t slot_in_block = 0;
    double t_submit = 0, t_sketch0 = 0, t_sketch1 = 0, t_profile0 = 0, t_done = 0;
    uint32_t batch_size = 0;
};

}  // namespace

struct sylph_pipeline {
    sylph_db* db = nullptr;
    sylph_comm* comm = nullptr;
    uint32_t n_workers = 3, depth = 8, max_batch = 8;
    uint32_t c = 200, k = 31;
    int reads_mode = SYLPH_READS_PAIRED, no_dedup = 0, seed_mode = SYLPH_SEED_AVX2_COMPAT, want_table = 0;
    double min_number_kmers = 0;
    std::vector<sylph_ctx*> wctx;
    std::vector<std::thread> wthreads;
    std::thread pthread;
    std::mutex mu;
    std::condition_variable cv_work, cv_sketched, cv_done;
    std::deque<Job*> to_sketch;                  // submitted, not yet taken by a worker
    std::deque<Job*> order;                      // every outstanding job in submission order (front = next to be returned)
    std::vector<std::vector<sylph_sketch*>> trash;   // per worker: sessions to destroy on the thread that owns their context
    std::vector<std::unique_ptr<Block>> blocks;
    Job* returned = nullptr;                     // the job whose views the caller holds (released by the next call)
    uint64_t next_seq = 0;
    uint64_t flush_upto = 0;                     // sharded: jobs with seq < flush_upto may go in a partial batch
    bool stop = false;
    // tuning (sylph_pipeline_set_option): see the option table in include/sylph_hip.h
    std::atomic<uint32_t> serial_outside{0};     // "serialize_outside" = 1: the r01-r04 placement of that wait (A/B; profiles/r05_ab_seed_turn.txt)
    std::atomic<uint32_t> serialize_seeding{1};  // 1 (default, r04: +3 %, profiles/r04_ab_pipeline_sweep.txt): one worker at a time runs its seeding kernel —
                                                 // two VALU-bound seeding kernels side by side only slow each other; the others are in their dedup/count tails
    std::string dedup_fpr, dedup_capacity;       // "dedup_fpr" / "dedup_capacity": handed to every session the pipeline opens (sylph_sketch_set_option; a10.hip)
    // the profile thread waits up to batch_wait_us for min_batch ready tables while more are being sketched.  r04: 2 (+1.2 %).  Round 6: 1 — with
    // the seeding tail below, tables probed one by one are 0.5 % (default flags: 2.4 %) FASTER, and the samples no longer complete in pairs:
    // completion intervals p99 2.0 -> 1.3 ms (profiles/r06_ab_latency.txt)
    uint32_t min_batch = 1, batch_wait_us = 400;
    std::mutex seed_mu;
    // serialize_seeding on the DEVICE: a push with a deferred verdict returns as soon as its kernels are queued, so holding a
    // mutex around it orders nothing on the GPU — the next worker's stream waits for the event the previous worker recorded
    // behind its seeding kernel instead (hosts never block on it)
    std::vector<hipEvent_t> seed_ev;             // one per worker
    hipEvent_t last_seed_ev = nullptr;           // guarded by seed_mu
    // A pipeline over several replicas of one database (sylph_pipeline_create_multi: one per GPU of the node) is only a router:
    // `children` are ordinary pipelines, one per replica; `route` holds the replica of every outstanding sample in submission
    // order, so that sylph_pipeline_next returns the samples in the order they were submitted whichever GPU had them.
    std::vector<sylph_pipeline*> children;
    std::deque<uint32_t> route;                  // guarded by mu
    uint32_t last_replica = 0;                   // replica of the sample sylph_pipeline_next returned last

    Block* take_block() {                        // mu held
        for (auto& b : blocks)
            if (b->users == 0) return b.get();
        blocks.emplace_back(new Block());
        return blocks.back().get();
    }
    void fail(Job* j, int rc) {
        j->status = rc;
        j->error = sylph_last_error();
    }
    void sketch_job(Job* j, int w) {
        j->t_sketch0 = now_s();
        int rc = SYLPH_OK;
        if (!j->sk) {
            rc = sylph_sketch_begin(wctx[w], c, k, reads_mode, no_dedup, seed_mode, &j->sk);
            j->worker = w;
            // device batches are borrowed until sylph_pipeline_next has returned the sample: the seeding verdict may wait for finish
            if (rc == SYLPH_OK && j->mem == SYLPH_MEM_DEVICE && j->batches.size() == 1) rc = sylph_sketch_set_option(j->sk, "borrow_until_finish", "1");
            std::string fpr, cap;
            { std::lock_guard<std::mutex> lk(mu); fpr = dedup_fpr; cap = dedup_capacity; }
            if (rc == SYLPH_OK && !fpr.empty()) rc = sylph_sketch_set_option(j->sk, "dedup_fpr", fpr.c_str());
            if (rc == SYLPH_OK && !cap.empty()) rc = sylph_sketch_set_option(j->sk, "dedup_capacity", cap.c_str());
            for (size_t i = 0; rc == SYLPH_OK && i < j->batches.size(); i++) {
                const sylph_read_batch& b = j->batches[i];
                std::unique_lock<std::mutex> seed_lock(seed_mu, std::defer_lock);
                const bool serial = serialize_seeding.load() != 0;      // read ONCE per push: the option may change between the two uses below
                sylph_ctx* cx = wctx[w];
                if (serial) {
                    seed_lock.lock();
                    // the library waits and records around the seeding kernel itself (common.h SeedTurn), not around the whole push
                    hipEvent_t prev = (last_seed_ev && last_seed_ev != seed_ev[(size_t)w]) ? last_seed_ev : nullptr;
                    if (serial_outside.load()) {            // A/B knob: rounds 1-4 waited in front of the whole push and recorded behind it
                        if (prev) (void)hipStreamWaitEvent(cx->stream, prev, 0);
                    } else {
                        cx->turn.gate = prev;
                        cx->turn.done = seed_ev[(size_t)w];
                        cx->turn.recorded = false;
                    }
                }
                rc = sylph_sketch_push_enc(j->sk, b.bases, b.rec_off, b.n_records, b.n_bases, j->mem, j->enc);
                if (serial) {
                    // (a push that launched no seeding kernel: nothing to order; one whose kernel did not record: behind the whole push)
                    if (cx->turn.recorded || hipEventRecord(seed_ev[(size_t)w], cx->stream) == hipSuccess) last_seed_ev = seed_ev[(size_t)w];
                    cx->turn = sylph_ctx::SeedTurn{};
                }
            }
        }
        if (rc == SYLPH_OK) rc = sylph_sketch_finish_device(j->sk, &j->dev_k, &j->dev_c, &j->n_table, &j->dup_removed);
        if (rc != SYLPH_OK) fail(j, rc);
        j->t_sketch1 = now_s();
    }
    void worker_main(int w) {
        (void)hipSetDevice(wctx[w]->device);
        for (;;) {
            Job* j = nullptr;
            std::vector<sylph_sketch*> dead;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !to_sketch.empty() || !trash[w].empty(); });
                dead.swap(trash[w]);
                if (!to_sketch.empty()) {
                    j = to_sketch.front();
                    to_sketch.pop_front();
                    j->state = JobState::Sketching;
                } else if (stop && dead.empty()) return;
            }
            for (sylph_sketch* s : dead) sylph_sketch_destroy(s);
            if (!j) continue;
            sketch_job(j, w);
            {
                std::lock_guard<std::mutex> lk(mu);
                j->state = JobState::Sketched;
            }
            cv_sketched.notify_all();
        }
    }
    // the profile thread: oldest sketched sample + every consecutive one that is ready, one probe launch
    void profile_main() {
        (void)hipSetDevice(db->ctx->device);
        size_t next_idx_seq = 0;                 // seq of the first job not yet profiled
        for (;;) {
            std::vector<Job*> batch;
            Block* blk = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                auto ready = [&] {
                    batch.clear();
                    // sharded: the batches must be the same on every rank, whatever the timing of its workers — exactly the next
                    // max_batch samples, or (after sylph_pipeline_flush) everything up to the flush point
                    uint64_t want = max_batch;
                    if (comm && flush_upto > next_idx_seq) want = std::min<uint64_t>(want, flush_upto - next_idx_seq);
                    for (Job* j : order) {
                        if (j->seq < next_idx_seq) continue;
                        if (j->state != JobState::Sketched || batch.size() >= want) break;
                        batch.push_back(j);
                    }
                    if (batch.empty()) return false;
                    if (comm && (batch.size() < want || (want < max_batch && flush_upto < next_idx_seq + want))) return false;
                    return true;
                };
                cv_sketched.wait(lk, [&] { return stop || ready(); });
                if (!ready()) return;            // stop, and nothing left to do (destroy drains the outstanding samples first)
                if (!comm && batch_wait_us && batch.size() < std::min(min_batch, max_batch)) {
                    // a fuller launch is cheaper per table (one probe + one sort + one copy for all of them): wait a little for
                    // the samples that are still being sketched — never for samples nobody has submitted yet
                    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(batch_wait_us);
                    auto enough = [&] {
                        if (stop || !ready()) return true;
                        if (batch.size() >= std::min(min_batch, max_batch)) return true;
                        size_t coming = 0;
                        for (Job* j : order) coming += j->state == JobState::Queued || j->state == JobState::Sketching;
                        return coming == 0;
                    };
                    while (!enough() && cv_sketched.wait_until(lk, deadline) != std::cv_status::timeout) {}
                    if (!ready()) return;
                }
                blk = take_block();
                blk->users = (uint32_t)batch.size();
            }
            const double t0 = now_s();
            std::vector<sylph_sample_ref> refs;
            std::vector<Job*> good;
            for (Job* j : batch)
                if (j->status == SYLPH_OK) { refs.push_back(sylph_sample_ref{j->dev_k, j->dev_c, j->n_table}); good.push_back(j); }
            int rc = SYLPH_OK;
            uint32_t width = 4, n_covs = 0;
            if (!good.empty() || comm) {
                rc = guarded([&] {
                    n_covs = comm ? contain_batch_sharded_impl(db, comm, refs.data(), (uint32_t)refs.size(), SYLPH_MEM_DEVICE, min_number_kmers, &width, &blk->mem)
                                  : contain_batch_impl(db, refs.data(), (uint32_t)refs.size(), SYLPH_MEM_DEVICE, min_number_kmers, &width, &blk->mem);
                });
            }
            if (rc == SYLPH_OK && want_table) {
                rc = guarded([&] {
                    for (Job* j : good) {
                        sylph_ctx* cx = j->sk->ctx;
                        std::lock_guard<std::mutex> lock(cx->mu);
                        DeviceGuard dg(cx->device);
                        j->host_k.resize(j->n_table);
                        j->host_c.resize(j->n_table);
                        if (j->n_table) {
                            cx->d2h(j->host_k.data(), j->dev_k, j->n_table * 8);
                            cx->d2h(j->host_c.data(), j->dev_c, j->n_table * 4);
                        }
                    }
                });
            }
            const double t1 = now_s();
            {
                std::lock_guard<std::mutex> lk(mu);
                blk->width = width;
                blk->n_covs = n_covs;
                uint32_t slot = 0;
                for (Job* j : batch) {
                    j->t_profile0 = t0;
                    j->t_done = t1;
                    j->batch_size = (uint32_t)batch.size();
                    j->block = blk;
                    if (j->status == SYLPH_OK) {
                        if (rc != SYLPH_OK) fail(j, rc);
                        else j->slot_in_block = slot++;
                    }
                    j->state = JobState::Profiled;
                }
                next_idx_seq = batch.back()->seq + 1;
            }
            cv_done.notify_all();
        }
    }
    // mu held.  Returns a session the CALLER of this function must destroy after dropping the lock (an adopted session lives
    // on a context of the pipeline's user), or nullptr (sessions of the workers go back to the thread that owns their context).
    sylph_sketch* release_returned() {
        Job* j = returned;
        returned = nullptr;
        if (!j) return nullptr;
        if (j->block && j->block->users) j->block->users--;
        sylph_sketch* dead = j->sk;
        const bool adopted = j->adopted;
        const int w = j->worker;
        delete j;
        if (!dead) return nullptr;
        if (!adopted && w >= 0) { trash[w].push_back(dead); cv_work.notify_all(); return nullptr; }
        return dead;
    }
};

namespace {

int submit_job(sylph_pipeline* p, std::unique_ptr<Job> j) {
    bool full = false;
    const int rc = guarded([&] {
        std::unique_lock<std::mutex> lk(p->mu);
        SY_REQUIRE(!p->stop, "pipeline is shutting down");
        if (p->order.size() >= p->depth) { full = true; return; }
        j->seq = p->next_seq++;
        j->t_submit = now_s();
        Job* raw = j.release();
        p->order.push_back(raw);
        p->to_sketch.push_back(raw);
        lk.unlock();
        p->cv_work.notify_one();
    });
    if (full) {
        set_error("sylph_pipeline_submit: %u samples are outstanding already (depth): call sylph_pipeline_next first", p->depth);
        return SYLPH_ERR_STATE;
    }
    return rc;
}

}  // namespace

extern "C" {

int sylph_pipeline_create(sylph_db* db, const sylph_pipeline_config* cfg, sylph_pipeline** out) {
    return guarded([&] {
        SY_REQUIRE(db && cfg && out, "null argument");
        SY_REQUIRE(cfg->struct_size == sizeof(sylph_pipeline_config), "sylph_pipeline_config.struct_size is %u, this library expects %zu",
                   cfg->struct_size, sizeof(sylph_pipeline_config));
        SY_REQUIRE(cfg->c >= 1 && (cfg->k == 21 || cfg->k == 31), "bad c / k");
        SY_REQUIRE(cfg->reads_mode == SYLPH_READS_SINGLE || cfg->reads_mode == SYLPH_READS_PAIRED, "bad reads_mode %d", cfg->reads_mode);
        SY_REQUIRE(cfg->seed_mode == SYLPH_SEED_SCALAR || cfg->seed_mode == SYLPH_SEED_AVX2_COMPAT, "bad seed_mode %d", cfg->seed_mode);
        SY_REQUIRE(cfg->n_workers <= 16 && cfg->max_batch <= 64 && cfg->depth <= 4096, "n_workers <= 16, max_batch <= 64, depth <= 4096");
        SY_REQUIRE((cfg->comm != nullptr) == (db->world > 1 || !db->bounds.empty()), "a sharded database needs cfg.comm (and only a sharded one takes it)");
        std::unique_ptr<sylph_pipeline> p(new sylph_pipeline());
        p->db = db;
        p->comm = cfg->comm;
        p->n_workers = cfg->n_workers ? cfg->n_workers : 3;
        p->max_batch = cfg->max_batch ? cfg->max_batch : 8;
        p->depth = cfg->depth ? cfg->depth : p->n_workers + 5;
        SY_REQUIRE(!p->comm || p->depth >= p->max_batch, "sharded: depth (%u) must reach max_batch (%u), the fixed batch of the exchange", p->depth, p->max_batch);
        p->c = cfg->c; p->k = cfg->k;
        p->reads_mode = cfg->reads_mode; p->no_dedup = cfg->no_dedup; p->seed_mode = cfg->seed_mode; p->want_table = cfg->want_table;
        p->min_number_kmers = cfg->min_number_kmers;
        p->trash.resize(p->n_workers);
        try {
            for (uint32_t w = 0; w < p->n_workers; w++) {
                sylph_ctx* cx = nullptr;
                if (sylph_ctx_create(db->ctx->device, nullptr, &cx) != SYLPH_OK) throw ArgError{sylph_last_error()};
                p->wctx.push_back(cx);
                hipEvent_t ev = nullptr;
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) throw ArgError{"hipEventCreate failed"};
                p->seed_ev.push_back(ev);
            }
        } catch (...) {
            for (sylph_ctx* cx : p->wctx) sylph_ctx_destroy(cx);
            for (hipEvent_t ev : p->seed_ev) (void)hipEventDestroy(ev);
            throw;
        }
        db->ctx->refs++;                         // the profile thread works on the database's context
        sylph_pipeline* raw = p.release();
        for (uint32_t w = 0; w < raw->n_workers; w++) raw->wthreads.emplace_back([raw, w] { raw->worker_main((int)w); });
        raw->pthread = std::thread([raw] { raw->profile_main(); });
        *out = raw;
    });
}

// One sample loop over the GPUs of a node, in ONE process (the reference's sample loop uses the whole machine through its rayon
// pool: contain.rs:252-295, sketch.rs:313,371).  dbs[i]: the database's replica on GPU i (sylph_db_replicate: the index copied over
// xGMI, not uploaded N times).  Every replica gets a pipeline of its own (cfg: per replica); a sample goes to the replica on the
// device its memory lives on (device batches, sessions) or to the least busy one (host batches), and comes back in submission order.
int sylph_pipeline_create_multi(sylph_db* const* dbs, uint32_t n_dbs, const sylph_pipeline_config* cfg, sylph_pipeline** out) {
    return guarded([&] {
        SY_REQUIRE(dbs && cfg && out && n_dbs >= 1 && n_dbs <= 64, "null argument, or not 1..64 replicas");
        SY_REQUIRE(cfg->struct_size == sizeof(sylph_pipeline_config), "sylph_pipeline_config.struct_size is %u, this library expects %zu",
                   cfg->struct_size, sizeof(sylph_pipeline_config));
        SY_REQUIRE(!cfg->comm, "replicas take no communicator (a sharded database runs one pipeline per rank)");
        for (uint32_t i = 0; i < n_dbs; i++) {
            SY_REQUIRE(dbs[i] && dbs[i]->world == 1 && dbs[i]->bounds.empty(), "replica %u is null or a shard", i);
            SY_REQUIRE(dbs[i]->n_genomes == dbs[0]->n_genomes && dbs[i]->n_kmers == dbs[0]->n_kmers, "replica %u is not a copy of replica 0", i);
        }
        std::unique_ptr<sylph_pipeline> p(new sylph_pipeline());
        p->db = dbs[0];
        p->n_workers = 0;
        try {
            for (uint32_t i = 0; i < n_dbs; i++) {
                sylph_pipeline* c = nullptr;
                if (sylph_pipeline_create(dbs[i], cfg, &c) != SYLPH_OK) throw ArgError{sylph_last_error()};
                p->children.push_back(c);
            }
        } catch (...) {
            for (sylph_pipeline* c : p->children) sylph_pipeline_destroy(c);
            throw;
        }
        p->depth = p->children[0]->depth * n_dbs;
        p->max_batch = p->children[0]->max_batch;
        *out = p.release();
    });
}

}  // extern "C"

namespace {

// multi: the replica a sample goes to — among those on `device` (-1: any) the one with the fewest samples outstanding
int pick_replica(sylph_pipeline* p, int device) {
    int best = -1;
    uint32_t best_out = 0;
    for (size_t i = 0; i < p->children.size(); i++) {
        sylph_pipeline* c = p->children[i];
        if (device >= 0 && c->db->ctx->device != device) continue;
        const uint32_t o = sylph_pipeline_outstanding(c);
        if (o >= c->depth) continue;
        if (best < 0 || o < best_out) { best = (int)i; best_out = o; }
    }
    return best;
}
// submits through `send` (a call on the chosen child) and records the route; mu orders submissions against each other
template <class Send>
int route_submit(sylph_pipeline* p, int device, const char* what, Send&& send) {
    std::lock_guard<std::mutex> lk(p->mu);
    const int r = pick_replica(p, device);
    if (r < 0) {
        bool any = false;
        for (sylph_pipeline* c : p->children) any |= device < 0 || c->db->ctx->device == device;
        if (!any) { set_error("%s: no replica of the database lives on device %d, where the sample's memory is", what, device); return SYLPH_ERR_INVALID; }
        set_error("%s: every replica%s has its `depth` samples outstanding: call sylph_pipeline_next first", what, device >= 0 ? " on the sample's device" : "");
        return SYLPH_ERR_STATE;
    }
    const int rc = send(p->children[(size_t)r]);
    if (rc == SYLPH_OK) p->route.push_back((uint32_t)r);
    return rc;
}

}  // namespace

extern "C" {

int sylph_pipeline_replica_of_last(sylph_pipeline* p) {
    if (!p) return -1;
    std::lock_guard<std::mutex> lk(p->mu);
    return p->children.empty() ? 0 : (int)p->last_replica;
}

int sylph_pipeline_submit(sylph_pipeline* p, const sylph_read_batch* batches, uint32_t n_batches, int mem, int enc, uint64_t tag) {
    if (!p || (n_batches && !batches)) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    if (!p->children.empty()) {
        int device = -1;
        if (mem == SYLPH_MEM_DEVICE && n_batches) {
            hipPointerAttribute_t at;
            const void* ptr = batches[0].bases ? (const void*)batches[0].bases : (const void*)batches[0].rec_off;
            if (hipPointerGetAttributes(&at, ptr) != hipSuccess) { (void)hipGetLastError(); set_error("sylph_pipeline_submit: cannot tell the device of the batch's memory"); return SYLPH_ERR_INVALID; }
            device = at.device;
        }
        return route_submit(p, device, "sylph_pipeline_submit", [&](sylph_pipeline* c) { return sylph_pipeline_submit(c, batches, n_batches, mem, enc, tag); });
    }
    std::unique_ptr<Job> j(new Job());
    j->batches.assign(batches, batches + n_batches);
    j->mem = mem; j->enc = enc; j->tag = tag;
    return submit_job(p, std::move(j));
}

int sylph_pipeline_submit_session(sylph_pipeline* p, sylph_sketch* sk, uint64_t tag) {
    if (!p || !sk) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    if (!p->children.empty())
        return route_submit(p, sk->ctx->device, "sylph_pipeline_submit_session", [&](sylph_pipeline* c) { return sylph_pipeline_submit_session(c, sk, tag); });
    if (sk->ctx->device != p->db->ctx->device) { set_error("the session lives on another device than the database"); return SYLPH_ERR_INVALID; }
    std::unique_ptr<Job> j(new Job());
    j->sk = sk; j->adopted = true; j->tag = tag;
    return submit_job(p, std::move(j));
}

int sylph_pipeline_flush(sylph_pipeline* p) {
    return guarded([&] {
        SY_REQUIRE(p, "null argument");
        for (sylph_pipeline* c : p->children) (void)sylph_pipeline_flush(c);
        { std::lock_guard<std::mutex> lk(p->mu); p->flush_upto = p->next_seq; }
        p->cv_sketched.notify_all();
    });
}

int sylph_pipeline_next(sylph_pipeline* p, sylph_pipeline_result* out) {
    if (p && out && !p->children.empty()) {
        uint32_t r;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (p->route.empty()) { set_error("sylph_pipeline_next: nothing is outstanding"); return SYLPH_ERR_INVALID; }
            r = p->route.front();
        }
        // (the replica's own pipeline hands its samples back in ITS submission order, which is the order they were routed in)
        const int rc = sylph_pipeline_next(p->children[r], out);
        if (rc == SYLPH_OK) { std::lock_guard<std::mutex> lk(p->mu); p->route.pop_front(); p->last_replica = r; }
        return rc;
    }
    return guarded([&] {
        SY_REQUIRE(p && out, "null argument");
        std::unique_lock<std::mutex> lk(p->mu);
        if (sylph_sketch* dead = p->release_returned()) { lk.unlock(); sylph_sketch_destroy(dead); lk.lock(); }
        SY_REQUIRE(!p->order.empty(), "sylph_pipeline_next: nothing is outstanding");
        Job* j = p->order.front();
        if (p->comm && j->state != JobState::Profiled && p->flush_upto <= j->seq) {
            // sharded: the profile thread waits for exactly max_batch samples (the probe batches must be the same on every rank);
            // with fewer outstanding and no flush nobody would ever wake this call — refuse instead of hanging a one-thread caller
            uint64_t unprofiled = 0;
            for (Job* o : p->order) unprofiled += o->state != JobState::Profiled;
            SY_REQUIRE_STATE(unprofiled >= p->max_batch,
                             "sylph_pipeline_next: %llu sample(s) outstanding, the sharded probe batch needs %u: submit more or call sylph_pipeline_flush",
                             (unsigned long long)unprofiled, p->max_batch);
        }
        p->cv_done.wait(lk, [&] { return j->state == JobState::Profiled; });
        p->order.pop_front();
        p->returned = j;
        memset(out, 0, sizeof(*out));
        out->tag = j->tag;
        out->status = j->status;
        out->error = j->status == SYLPH_OK ? nullptr : j->error.c_str();
        out->n_table = j->n_table;
        out->dup_removed = j->dup_removed;
        out->dev_kmers = j->dev_k;
        out->dev_counts = j->dev_c;
        if (p->want_table && j->status == SYLPH_OK) { out->kmers = j->host_k.data(); out->counts = j->host_c.data(); }
        out->t_submit = j->t_submit; out->t_sketch_begin = j->t_sketch0; out->t_sketch_end = j->t_sketch1;
        out->t_profile_begin = j->t_profile0; out->t_done = j->t_done;
        out->probe_batch = j->batch_size;
        if (j->status == SYLPH_OK && j->block) {
            const Block& b = *j->block;
            const char* h = (const char*)b.mem.p;
            const uint64_t G = p->db->n_genomes, row0 = (uint64_t)j->slot_in_block * G;
            out->cov_off = (const uint64_t*)h + row0;                  // G + 1 entries; values index `covs`
            out->contain_count = (const uint32_t*)(h + b.mem.lay.ccount) + row0;
            out->covs = h + b.mem.lay.covs;
            out->cov_width = b.width;
            out->n_covs = out->cov_off[G] - out->cov_off[0];
        }
    });
}

uint32_t sylph_pipeline_outstanding(sylph_pipeline* p) {
    if (!p) return 0;
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->children.empty()) return (uint32_t)p->route.size();
    return (uint32_t)p->order.size();
}

int sylph_pipeline_set_option(sylph_pipeline* p, const char* key, const char* value) {
    if (!p || !key || !value) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    if (!p->children.empty()) {
        for (sylph_pipeline* c : p->children) { const int rc = sylph_pipeline_set_option(c, key, value); if (rc != SYLPH_OK) return rc; }
        return SYLPH_OK;
    }
    {   // the pipeline's own knobs; everything else goes to the workers' contexts
        if (!strcmp(key, "serialize_seeding")) { p->serialize_seeding.store((uint32_t)strtoul(value, nullptr, 10)); return SYLPH_OK; }
        if (!strcmp(key, "serialize_outside")) { p->serial_outside.store((uint32_t)strtoul(value, nullptr, 10)); return SYLPH_OK; }
        uint32_t* dst = !strcmp(key, "min_batch") ? &p->min_batch : !strcmp(key, "batch_wait_us") ? &p->batch_wait_us : nullptr;
        if (dst) {                               // (the profile thread reads both with p->mu held)
            std::lock_guard<std::mutex> lk(p->mu);
            *dst = (uint32_t)strtoul(value, nullptr, 10);
            return SYLPH_OK;
        }
    }
    if (!strcmp(key, "shard_reduce")) return sylph_ctx_set_option(p->db->ctx, key, value);      // (the database's context only: "alltoall" | "allgather")
    if (!strcmp(key, "dedup_fpr") || !strcmp(key, "dedup_capacity")) {      // for the sessions opened from now on (checked by the session)
        std::lock_guard<std::mutex> lk(p->mu);
        (key[6] == 'f' ? p->dedup_fpr : p->dedup_capacity) = value;
        return SYLPH_OK;
    }
    for (sylph_ctx* cx : p->wctx) {
        const int rc = sylph_ctx_set_option(cx, key, value);
        if (rc != SYLPH_OK) return rc;
    }
    if (!strcmp(key, "profile_only") || !strcmp(key, "shard_reduce")) return sylph_ctx_set_option(p->db->ctx, key, value);   // the profile thread's timers / the sharded batch's reduction live there
    return SYLPH_OK;
}

int sylph_pipeline_profile(sylph_pipeline* p, int enable) {
    if (!p) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    if (!p->children.empty()) {
        for (sylph_pipeline* c : p->children) { const int rc = sylph_pipeline_profile(c, enable); if (rc != SYLPH_OK) return rc; }
        return SYLPH_OK;
    }
    for (sylph_ctx* cx : p->wctx) {
        const int rc = sylph_ctx_profile(cx, enable);
        if (rc != SYLPH_OK) return rc;
    }
    return sylph_ctx_profile(p->db->ctx, enable);
}

int sylph_pipeline_kernel_stats(sylph_pipeline* p, const char* family, double* total_ms, uint64_t* launches) {
    if (!p || !family) { set_error("null argument"); return SYLPH_ERR_INVALID; }
    double ms = 0;
    uint64_t n = 0;
    if (!p->children.empty()) {
        for (sylph_pipeline* c : p->children) {
            double m = 0;
            uint64_t l = 0;
            const int rc = sylph_pipeline_kernel_stats(c, family, &m, &l);
            if (rc != SYLPH_OK) return rc;
            ms += m;
            n += l;
        }
        if (total_ms) *total_ms = ms;
        if (launches) *launches = n;
        return SYLPH_OK;
    }
    std::vector<sylph_ctx*> all(p->wctx);
    all.push_back(p->db->ctx);
    for (sylph_ctx* cx : all) {
        double m = 0;
        uint64_t l = 0;
        const int rc = sylph_ctx_kernel_stats(cx, family, &m, &l);
        if (rc != SYLPH_OK) return rc;
        ms += m;
        n += l;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    return SYLPH_OK;
}

void sylph_pipeline_destroy(sylph_pipeline* p) {
    if (!p) return;
    if (!p->children.empty()) {                  // the router: its replicas' pipelines finish what they have and go
        for (sylph_pipeline* c : p->children) sylph_pipeline_destroy(c);
        delete p;
        return;
    }
    sylph_sketch* dead = nullptr;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        dead = p->release_returned();
        p->flush_upto = p->next_seq;             // sharded: what is outstanding goes in a last, partial batch (as after sylph_pipeline_flush;
                                                 // every rank destroys its pipeline with the same number of samples outstanding)
    }
    if (dead) sylph_sketch_destroy(dead);
    p->cv_sketched.notify_all();
    // outstanding samples are finished (their inputs are the caller's memory: nothing may still read them afterwards)
    for (;;) {
        std::unique_lock<std::mutex> lk(p->mu);
        if (p->order.empty()) break;
        Job* j = p->order.front();
        p->cv_done.wait(lk, [&] { return j->state == JobState::Profiled; });
        p->order.pop_front();
        p->returned = j;
        dead = p->release_returned();
        lk.unlock();
        if (dead) sylph_sketch_destroy(dead);
    }
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_work.notify_all();
    p->cv_sketched.notify_all();
    for (auto& t : p->wthreads) t.join();
    if (p->pthread.joinable()) p->pthread.join();
    for (size_t w = 0; w < p->wctx.size(); w++) {
        for (sylph_sketch* s : p->trash[w]) sylph_sketch_destroy(s);
        sylph_ctx_destroy(p->wctx[w]);
    }
    for (hipEvent_t ev : p->seed_ev) (void)hipEventDestroy(ev);
    sylph_ctx* dbctx = p->db->ctx;
    p->blocks.clear();
    delete p;
    ctx_unref(dbctx);
}

}  // extern "C"
