// commands.cpp — minimal restatement of the sketch / contain command drivers (sketch.rs:276-479, contain.rs:115-351)
// around the GPU engine: parse records on the host, push batches through the C ABI, keep the sequential f64
// bookkeeping here, run the statistics on the containment results, print the reference's TSV rows.
#include <functional>
#include <sys/stat.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <condition_variable>
#include <fstream>
#include <future>
#include <map>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>

#include "sylph_host.hpp"

namespace sylph_host {

namespace {

void hip_check(int rc, const char* what) {
    if (rc != SYLPH_OK) throw Error{1, std::string(what) + ": " + sylph_last_error()};
}
void warn(const std::string& m) { fprintf(stderr, "WARN  [sylph_hip] %s\n", m.c_str()); }
void info(const std::string& m) { fprintf(stderr, "INFO  [sylph_hip] %s\n", m.c_str()); }
}  // namespace
bool exact_dedup_accepted(bool flag) {
    const char* e = getenv("SYLPH_HIP_EXACT_DEDUP");
    return flag || (e && *e && strcmp(e, "0") != 0);
}
namespace {

template <class T>
std::vector<T> take(T* p, uint64_t n) {
    std::vector<T> v(p, p + n);
    sylph_free(p);
    return v;
}

std::string basename_of(const std::string& p) {
    const size_t s = p.find_last_of('/');
    return s == std::string::npos ? p : p.substr(s + 1);
}
std::string dirname_of(const std::string& p) {
    const size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string() : p.substr(0, s);
}
void create_dir_all(const std::string& dir) {
    if (dir.empty()) return;
    std::string cur;
    for (size_t i = 0; i <= dir.size(); i++) {
        if (i == dir.size() || dir[i] == '/') {
            if (!cur.empty() && cur != "." && cur != "..") mkdir(cur.c_str(), 0777);
        }
        if (i < dir.size()) cur += dir[i];
    }
}
std::string path_join(const std::string& a, const std::string& b) {
    if (a.empty()) return b;
    return a.back() == '/' ? a + b : a + "/" + b;
}
void parse_line_file(const std::string& file, std::vector<std::string>& out) {   // sketch.rs:252
    std::ifstream f(file);
    if (!f) throw Error{1, "could not open list file " + file};
    std::string line;
    while (std::getline(f, line)) out.push_back(line);
}

// f(i) for i in [0, n) on up to `threads` threads (f must not throw)
template <class F>
void parallel_for(size_t n, uint64_t threads, F&& f) {
    const size_t t = std::max<size_t>(1, std::min<size_t>(threads, (n + 15) / 16));
    if (t <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::atomic<size_t> next{0};
    auto work = [&] { for (size_t i = next++; i < n; i = next++) f(i); };
    std::vector<std::thread> pool;
    for (size_t w = 1; w < t; w++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
}

struct Session {   // RAII
    sylph_sketch* sk = nullptr;
    // dedup_fpr != 0 (pairs): the reference's dup_removal_lsh_full over its cuckoo filter (sketch.rs:839-848); 0: the exact set (:829-838)
    Session(Engine& e, uint64_t c, uint64_t k, bool paired, bool no_dedup, double dedup_fpr = 0.) {
        hip_check(sylph_sketch_begin(e.context(), (uint32_t)c, (uint32_t)k, paired ? SYLPH_READS_PAIRED : SYLPH_READS_SINGLE,
                                     no_dedup ? 1 : 0, SYLPH_SEED_AVX2_COMPAT, &sk), "sylph_sketch_begin");
        if (paired && !no_dedup && dedup_fpr != 0.) {
            char v[64];
            snprintf(v, sizeof(v), "%.17g", dedup_fpr);
            const int rc = sylph_sketch_set_option(sk, "dedup_fpr", v);
            if (rc != SYLPH_OK) { sylph_sketch_destroy(sk); sk = nullptr; hip_check(rc, "sylph_sketch_set_option(dedup_fpr)"); }
        }
    }
    ~Session() { sylph_sketch_destroy(sk); }
    // keep != nullptr: the caller takes the pushed, unfinished session (the profile pipeline finishes it on the device and
    // probes its table where it lies); out's table stays empty
    void finish_or_keep(SequencesSketch& out, sylph_sketch** keep) {
        if (keep) { *keep = sk; sk = nullptr; return; }
        finish(out);
    }
    void finish(SequencesSketch& out) {
        uint64_t* k = nullptr; uint32_t* c = nullptr; uint64_t n = 0, dup = 0;
        hip_check(sylph_sketch_finish(sk, &k, &c, &n, &dup), "sylph_sketch_finish");
        out.kmers = take(k, n);
        out.counts = take(c, n);
    }
};

}  // namespace

// SYLPH_HIP_FEED_TRACE: "[sylph_hip t+123.4 ms] what" — milliseconds since this library was loaded (just behind the dynamic linker)
static const std::chrono::steady_clock::time_point g_loaded = std::chrono::steady_clock::now();
bool fast_exit() { static const bool f = getenv("SYLPH_HIP_CLEAN_EXIT") == nullptr; return f; }
// Work that nobody waits for (unmapping a sample's files, handing inflated copies back) runs on threads of its own; a command that
// leaves through exit() joins them first (they use function-local statics that exit() destroys), one that leaves through _exit does not.
namespace {
// ONE long-lived reaper thread and a queue (round 6): a thread per task, joined only at the end of the command, left a
// finished-but-unjoined thread (its stack mapping) behind every gzip sample — tens of thousands of samples in one command run into
// vm.max_map_count or the thread limit, and std::thread's constructor then throws inside a destructor.  A task that cannot be queued
// (no memory for the node, no thread) runs inline.
struct Reaper {
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::vector<std::function<void()>> queue;
    bool running = false, started = false;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return !queue.empty(); });
            std::vector<std::function<void()>> todo;
            todo.swap(queue);
            running = true;
            lk.unlock();
            for (auto& f : todo) { try { f(); } catch (...) {} }
            todo.clear();
            lk.lock();
            running = false;
            if (queue.empty()) cv_idle.notify_all();
        }
    }
    void add(std::function<void()> f) {
        try {
            std::unique_lock<std::mutex> lk(mu);
            if (!started) { std::thread([this] { loop(); }).detach(); started = true; }   // (it ends with the process)
            queue.push_back(std::move(f));
            lk.unlock();
            cv_work.notify_one();
        } catch (...) { try { f(); } catch (...) {} }
    }
    void drain() {
        std::unique_lock<std::mutex> lk(mu);
        cv_idle.wait(lk, [&] { return queue.empty() && !running; });
    }
};
Reaper& reaper() { static Reaper* r = new Reaper(); return *r; }
void background(std::function<void()> f) { reaper().add(std::move(f)); }
// ... and one thread that writes finished sketches out, in the order they were finished, while their worker is at its next sample
// (a 1 Gbp pair's .sylsp: 48 MB, 25-35 ms of a warm sample's ~120); always drained before the command returns
Reaper& writers() { static Reaper* r = new Reaper(); return *r; }
}  // namespace
void join_background() { reaper().drain(); }
void trace_mark(const char* what) {
    static const bool trace = getenv("SYLPH_HIP_FEED_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[sylph_hip t+%.1f ms] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - g_loaded).count() * 1e3, what);
}

struct Engine::Gate { std::mutex mu; std::condition_variable cv; bool ready = false; };

Engine::Engine(int dev) : device(dev), gate_(new Gate()) {
    init_ = std::thread([this] {
        auto open_gate = [this] { { std::lock_guard<std::mutex> lk(gate_->mu); gate_->ready = true; } gate_->cv.notify_all(); };
        try {
            trace_mark("engine: bring-up begins");
            hip_check(sylph_ctx_create(device, nullptr, &ctx_), "sylph_ctx_create");
            trace_mark("engine: context created (runtime initialised, stream, pinned page)");
            if (getenv("SYLPH_HIP_NO_WARMUP")) { open_gate(); return; }
            // first use of a kernel loads its code object (~25 ms for the sketch kernels, the filter dedup's included): do it here, with
            // a few dummy pairs at a small c (so that the pairs really carry seeds and every kernel of a finish runs)
            sylph_sketch* sk = nullptr;
            hip_check(sylph_sketch_begin(ctx_, 5, 31, SYLPH_READS_PAIRED, 0, SYLPH_SEED_AVX2_COMPAT, &sk), "sylph_sketch_begin");
            int rc = sylph_sketch_set_option(sk, "dedup_fpr", "0.0001");
            std::vector<uint8_t> b(8 * 150, 'A');
            for (size_t i = 0; i < b.size(); i++) b[i] = "ACGT"[(i * 2654435761u >> 7) & 3];
            uint64_t* k = nullptr; uint32_t* c = nullptr; uint64_t n = 0, dup = 0;
            if (warm_text_route.load() && device_feed_enabled()) {
                // the process's first sample is a gzip one on the device route: its text is indexed and pushed by the route's own kernels
                // (csrc/fastq.hip) — the warm-up pairs go THAT way, as FASTQ text, so that those kernels' first launches (~20 ms of host-side
                // binding between the sample's inflate and its push otherwise) happen here; the seeding / dedup kernels behind them are the same
                std::string fq[2];
                for (int m = 0; m < 2; m++) {
                    fq[m] = std::string(16, '\n');
                    for (int r = 0; r < 4; r++) fq[m] += "@w\n" + std::string((const char*)b.data() + (size_t)(m * 4 + r) * 150, 150) + "\n+\n" + std::string(150, 'I') + "\n";
                    fq[m] += std::string(32, '\n');                                         // (readable 16 bytes either side of the text)
                }
                sylph_fastq *fa = nullptr, *fb = nullptr;
                rc = rc || sylph_fastq_index(ctx_, fq[0].data() + 16, fq[0].size() - 48, SYLPH_MEM_HOST, &fa) ||
                     sylph_fastq_index(ctx_, fq[1].data() + 16, fq[1].size() - 48, SYLPH_MEM_HOST, &fb) ||
                     sylph_sketch_set_option(sk, "borrow_until_finish", "1") || sylph_sketch_push_fastq(sk, fa, fb, 0, 4) ||
                     sylph_sketch_finish(sk, &k, &c, &n, &dup);
                sylph_sketch_destroy(sk);
                sk = nullptr;
                sylph_fastq_destroy(fa); sylph_fastq_destroy(fb);
            } else {
                std::vector<uint64_t> off(9);
                for (size_t i = 0; i < off.size(); i++) off[i] = i * 150;
                rc = rc || sylph_sketch_push(sk, b.data(), off.data(), 8, SYLPH_MEM_HOST) || sylph_sketch_finish(sk, &k, &c, &n, &dup);
            }
            sylph_free(k); sylph_free(c);
            if (sk) sylph_sketch_destroy(sk);
            if (rc) hip_check(rc, "warm-up");
            trace_mark("engine: sketch kernels loaded (warm-up sample done)");
            open_gate();             // sessions may be opened from here on ...
            // ... while the packed double buffers of the indexed feed are page-locked (~27 ms; round 5: BEHIND the gate — the first
            // sample of a process travels from pageable memory it was gathered into during the bring-up: PinnedBatch::gather_packed_early).
            // (Only these: the 256 MB ASCII batch of the sequential reader is page-locked by its first add() — most commands never need it.)
            // (using the device FASTQ route's kernels once from here, on a second context, so that their first launches are not the first
            //  sample's was tried: it takes 100-330 ms beside the sample's own work and doubled the sample's inflate — profiles/r06_gz_e2e_trace_final.txt)
            if (defer_pinned.load()) { trace_mark("engine: page-locked feed buffers left to their first user"); return; }
            batch.prealloc_packed();
            trace_mark("engine: packed double buffers page-locked");
            if (device_feed_enabled()) { text.prepare(ctx_); trace_mark("engine: text uploader's chunks page-locked"); }
        } catch (const Error& e) { init_error_ = e.msg; init_code_ = e.code ? e.code : 1; open_gate(); }
    });
}
bool Engine::ready() const {
    std::lock_guard<std::mutex> lk(gate_->mu);
    return gate_->ready;
}
sylph_ctx* Engine::context() {
    {
        std::unique_lock<std::mutex> lk(gate_->mu);
        gate_->cv.wait(lk, [&] { return gate_->ready; });
    }
    if (init_code_) throw Error{init_code_, init_error_};
    return ctx_;
}
void Engine::wait_pinned() {
    (void)context();
    static std::mutex join_mu;                       // (sample threads share the command's first engine only through run_job: one joiner)
    std::lock_guard<std::mutex> lk(join_mu);
    if (init_.joinable()) init_.join();
    if (init_code_) throw Error{init_code_, init_error_};
}
Engine::~Engine() {
    if (init_.joinable()) init_.join();
    if (ctx_) sylph_ctx_destroy(ctx_);
}

namespace {
// Uncompressed 4-line FASTQ (the common case): the files are indexed by parse_threads() workers, whole batches are gathered
// into page-locked memory in parallel and pushed; the record loop of the reference shrinks to its one sequential piece, the
// running mean of the read lengths (f64, file order: sketch.rs:941-943, :825-826), which runs on its own thread meanwhile.
// index_inputs returns nothing when a file is not that simple (gzip, FASTA, blank lines, a malformed record anywhere): the
// caller then runs the sequential reader with needletail's exact record/error semantics.  No GPU call happens before the
// files are indexed, so the engine's background bring-up overlaps with it.
struct IndexedInput { std::unique_ptr<FastqIndex> a, b; };
struct ThreadJoiner {   // joins on every way out of a scope (an Error thrown past a joinable std::thread is std::terminate)
    std::thread& t;
    ~ThreadJoiner() { if (t.joinable()) t.join(); }
};
// build = false: the files are mapped / inflated, not indexed — what the device-side route needs of a gzip file (finish_index does
// the rest on the host when that route is not taken after all)
std::optional<IndexedInput> index_inputs(const std::string& f1, const std::string* f2, bool build = true) {
    if (getenv("SYLPH_HIP_SEQUENTIAL_FEED")) return std::nullopt;
    const unsigned T = parse_threads();
    IndexedInput in;
    std::exception_ptr err_b;
    std::thread tb;
    ThreadJoiner jb{tb};
    if (f2) tb = std::thread([&] {                        // the two mate files are indexed concurrently
        try { in.b.reset(new FastqIndex(*f2, T, build)); } catch (...) { err_b = std::current_exception(); }
    });
    in.a.reset(new FastqIndex(f1, T, build));
    if (tb.joinable()) tb.join();
    if (err_b) std::rethrow_exception(err_b);
    if (build ? (!in.a->ok || (f2 && !in.b->ok)) : (!in.a->text_ready || (f2 && !in.b->text_ready))) return std::nullopt;
    return in;
}
bool finish_index(IndexedInput& in) {
    const unsigned T = parse_threads();
    std::exception_ptr err_b;
    std::thread tb;
    ThreadJoiner jb{tb};
    if (in.b) tb = std::thread([&] { try { in.b->build_index(T); } catch (...) { err_b = std::current_exception(); } });
    in.a->build_index(T);
    if (tb.joinable()) tb.join();
    if (err_b) std::rethrow_exception(err_b);
    return in.a->ok && (!in.b || in.b->ok);
}
bool is_gzip_file(const std::string& f) {
    struct stat st;
    if (stat(f.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false;   // (never read from a pipe here: the bytes would be gone for its reader)
    FILE* fp = fopen(f.c_str(), "rb");
    if (!fp) return false;
    unsigned char h[2] = {0, 0};
    const size_t n = fread(h, 1, 2, fp);
    fclose(fp);
    return n == 2 && h[0] == 0x1f && h[1] == 0x8b;
}
// Batches of whole records (pairs) of at most BATCH_BASES bases / BATCH_RECS records
std::vector<std::pair<size_t, size_t>> cut_batches(size_t n, bool paired, const std::vector<uint64_t>& ca, const std::vector<uint64_t>& cb) {
    std::vector<std::pair<size_t, size_t>> out;
    size_t i0 = 0;
    while (i0 < n) {
        // largest i1 with at most BATCH_BASES bases and BATCH_RECS records in [i0, i1)
        size_t lo = i0 + 1, hi = std::min(n, i0 + PinnedBatch::BATCH_RECS / (paired ? 2 : 1));
        auto bases_upto = [&](size_t i) { return ca[i] - ca[i0] + (paired ? cb[i] - cb[i0] : 0); };
        while (lo < hi) {
            const size_t mid = lo + (hi - lo + 1) / 2;
            if (bases_upto(mid) <= PinnedBatch::BATCH_BASES) lo = mid; else hi = mid - 1;
        }
        out.emplace_back(i0, lo);
        i0 = lo;
    }
    return out;
}
// open_session: creates the sample's session — it waits for the engine's context, so it is called as late as possible: whatever
// batches can be gathered and packed before the context is up (the first sample of a process: ~0.15 s of idle parse threads otherwise)
// are gathered into pageable memory first (round 5)
void sketch_indexed(Engine& e, const std::function<sylph_sketch*()>& open_session, const IndexedInput& in, double& mean_read_length) {
    const unsigned T = parse_threads();
    static const bool trace = getenv("SYLPH_HIP_FEED_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const double t = now();
        fprintf(stderr, "[sylph_hip feed] %-28s %8.3f ms\n", what, (t - t_prev) * 1e3);
        t_prev = t;
    };
    const FastqIndex& a = *in.a;
    const FastqIndex* b = in.b.get();
    const size_t n = b ? std::min(a.n_records(), b->n_records()) : a.n_records();   // lock-step readers: sketch.rs:813-815
    static const std::vector<uint64_t> no_cum;
    const std::vector<uint64_t>& ca = a.cum;                             // prefix sums of the read lengths, from the index workers
    const std::vector<uint64_t>& cb = b ? b->cum : no_cum;
    // the one sequential piece of the reference's record loop (f64, file order: sketch.rs:941-943, :825-826), on its own thread
    // while the batches travel
    double mean = 0.;
    std::thread tm([&] { double counter = 0.; for (size_t i = 0; i < n; i++) { counter += 1.; mean = mean + ((double)a.seq_len[i] - mean) / counter; } });
    ThreadJoiner jm{tm};
    const auto batches = cut_batches(n, b != nullptr, ca, cb);
    size_t early = 0;
    constexpr size_t MAX_EARLY = 16;                                     // 4 Gbp = 1 GB of packed bases in pageable memory at most
    while (!e.ready() && early < batches.size() && early < MAX_EARLY) {
        e.batch.gather_packed_early(a, b, ca, b ? &cb : nullptr, batches[early].first, batches[early].second, T);
        early++;
    }
    if (early) lap("early batches: gather + pack (context still coming up)");
    sylph_sketch* const sk = open_session();
    e.batch.flush(sk);
    if (early) {
        e.batch.push_early(sk);
        if (early < batches.size()) {
            a.release_behind(a.seq_off[batches[early].first]);
            if (b) b->release_behind(b->seq_off[batches[early].first]);
        }
        lap("early batches: pushed");
    }
    // Batches are gathered AND packed to 2 bits per base by the parse threads into one of two page-locked slots while the other
    // slot is on its way to the device: the push of batch i (PCIe + kernels) hides behind the gather of batch i + 1.
    if (early < batches.size()) {
        e.wait_pinned();
        e.batch.gather_packed((int)(early & 1), a, b, ca, b ? &cb : nullptr, batches[early].first, batches[early].second, T);
        lap("batch: gather + pack");
    }
    for (size_t bi = early; bi < batches.size(); bi++) {
        std::exception_ptr err;
        std::thread tg;
        ThreadJoiner jg{tg};
        if (bi + 1 < batches.size())
            tg = std::thread([&, bi] {
                try { e.batch.gather_packed((int)((bi + 1) & 1), a, b, ca, b ? &cb : nullptr, batches[bi + 1].first, batches[bi + 1].second, T); }
                catch (...) { err = std::current_exception(); }
            });
        e.batch.push_packed(sk, (int)(bi & 1));
        if (tg.joinable()) tg.join();
        if (err) std::rethrow_exception(err);
        // everything before the next batch's first record has been gathered: an inflated copy gives those pages back
        if (bi + 1 < batches.size()) {
            a.release_behind(a.seq_off[batches[bi + 1].first]);
            if (b) b->release_behind(b->seq_off[batches[bi + 1].first]);
        }
        lap("batch: push || gather next");
    }
    tm.join();
    mean_read_length = mean;
    lap("running mean (joined)");
}

// Round 5, the device-side route for plain FASTQ (csrc/fastq.hip): the files' TEXT goes to the device through the engine's uploader and
// the library finds the records there; the host keeps the one sequential piece (the running mean of the mate-1 read lengths, from the
// lengths the device hands back).  Taken for every sample whose engine is already up — a process's FIRST sample is indexed and gathered
// on the host while the GPU runtime initialises, which no device can do — unless SYLPH_HIP_FEED_DEVICE=0 (feed.cpp device_feed_enabled).
// Returns false, having pushed
// nothing, when the files are not plain four-line FASTQ (SYLPH_ERR_FORMAT, gzip, FASTA): the caller takes the host route, whose
// record and error semantics are needletail's.
// `text`: the files' text where it lies in host memory already (the inflated copies of gzip files) instead of the files themselves
// Round 6, gzip on the device (csrc/inflate.hip): a sample whose files are gzip sends their COMPRESSED bytes (0.4 B per base instead of
// 2.1) and the library inflates them there — sylph_inflate; SYLPH_HIP_INFLATE_DEVICE=0 keeps the host's inflate.  What the library
// declines (SYLPH_ERR_FORMAT: damaged, not gzip after all, a stream it does not take; SYLPH_ERR_NOMEM) goes the host way, whose
// error semantics are needletail's.
bool device_inflate_enabled() {
    static const bool on = [] {
        if (!device_feed_enabled()) return false;
        if (const char* e = getenv("SYLPH_HIP_INFLATE_DEVICE")) return atoi(e) != 0;
        return true;
    }();
    return on;
}
namespace {
struct MappedFile {                       // the compressed bytes of one file (read-only mapping of the page cache)
    const uint8_t* data = nullptr;
    size_t size = 0;
    explicit MappedFile(const std::string& path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (p != MAP_FAILED) { data = (const uint8_t*)p; size = (size_t)st.st_size; }
        }
        close(fd);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    ~MappedFile() { if (data) munmap(const_cast<uint8_t*>(data), size); }
};
struct InflatedText { sylph_inflated* h = nullptr; ~InflatedText() { sylph_inflated_destroy(h); } };
}  // namespace

bool sketch_fastq_on_device(Engine& e, const std::function<sylph_sketch*()>& open_session, const std::string& f1, const std::string* f2,
                            double& mean_read_length, const IndexedInput* text = nullptr) {
    if (!device_feed_enabled()) return false;
    const bool gz = !text && device_inflate_enabled() && is_gzip_file(f1) && (!f2 || is_gzip_file(*f2));
    if (!gz && !e.ready()) return false;             // (a gzip sample waits for the engine: nothing on the host side is faster than that)
    static const bool trace = getenv("SYLPH_HIP_FEED_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const double t = now();
        fprintf(stderr, "[sylph_hip feed] %-28s %8.3f ms\n", what, (t - t_prev) * 1e3);
        t_prev = t;
    };
    sylph_ctx* ctx = e.context();
    std::vector<std::string> files{f1};
    if (f2) files.push_back(*f2);
    std::vector<TextUploader::Text> texts;
    InflatedText inflated;                           // (destroyed behind the indexes that borrow its text: declared before them)
    if (gz) {
        // both mates in ONE call (sylph_inflate_files): one scan, one decode launch, one chain — the mates share a block's latency
        std::vector<std::unique_ptr<MappedFile>> maps;
        std::vector<const void*> ptrs;
        std::vector<uint64_t> lens;
        for (const auto& f : files) {
            maps.emplace_back(new MappedFile(f));
            if (!maps.back()->data) return false;
            ptrs.push_back(maps.back()->data);
            lens.push_back(maps.back()->size);
        }
        const int rc = sylph_inflate_files(ctx, ptrs.data(), lens.data(), (uint32_t)ptrs.size(), SYLPH_MEM_HOST, &inflated.h);
        if (rc == SYLPH_ERR_FORMAT || rc == SYLPH_ERR_NOMEM) {
            if (trace) fprintf(stderr, "[sylph_hip feed] device inflate declined %s: %s\n", files[0].c_str(), sylph_last_error());
            return false;
        }
        hip_check(rc, "sylph_inflate_files");
        for (uint32_t i = 0; i < ptrs.size(); i++) {
            const void* dev = nullptr;
            uint64_t bytes = 0;
            hip_check(sylph_inflated_file(inflated.h, i, &dev, &bytes), "sylph_inflated_file");
            if (!bytes) return false;
            texts.push_back(TextUploader::Text{(const uint8_t*)dev, bytes});
        }
        lap("device route: gzip inflated on the device");
    } else if (text) {
        // The first text an engine sends pays for the route's device buffers (~40-90 ms of allocations, once); skipping the host's index
        // and gather of an inflated copy buys that back only for a sample of some size: 0.45 against 0.53 s for a 1 Gbp gzip pair, 0.34
        // against 0.30 s for its first mate alone (profiles/r05_feed_device_route.txt)
        const uint64_t bytes = text->a->size + (f2 ? text->b->size : 0);
        if (!e.text.warm() && bytes < (1500ull << 20)) return false;
        std::vector<TextUploader::Mem> mem{{text->a->data, text->a->size}};
        if (f2) mem.push_back({text->b->data, text->b->size});
        if (!e.text.send(ctx, mem, parse_threads(), texts)) return false;
    } else if (!e.text.send(ctx, files, parse_threads(), texts)) return false;
    if (!gz) lap("device route: text uploaded");
    struct Fq { sylph_fastq* f = nullptr; ~Fq() { sylph_fastq_destroy(f); } } fa, fb;
    auto index = [&](const TextUploader::Text& t, Fq& out) {
        const int rc = sylph_fastq_index(ctx, t.dev, t.bytes, SYLPH_MEM_DEVICE, &out.f);
        if (rc == SYLPH_ERR_FORMAT || rc == SYLPH_ERR_NOMEM) return false;   // not for this route / no room for it: the host feed
        hip_check(rc, "sylph_fastq_index");
        return true;
    };
    if (!index(texts[0], fa) || (f2 && !index(texts[1], fb))) return false;
    uint64_t na = 0, nb = 0, bases_a = 0, bases_b = 0;
    hip_check(sylph_fastq_counts(fa.f, &na, &bases_a), "sylph_fastq_counts");
    if (f2) hip_check(sylph_fastq_counts(fb.f, &nb, &bases_b), "sylph_fastq_counts");
    const uint64_t n = f2 ? std::min(na, nb) : na;                         // lock-step readers: sketch.rs:813-815
    lap("device route: records found");
    std::vector<uint32_t> la(n), lb;
    hip_check(sylph_fastq_lengths(fa.f, 0, n, la.data()), "sylph_fastq_lengths");
    // the one sequential piece of the reference's record loop (f64, file order: sketch.rs:941-943, :825-826), on its own thread
    double mean = 0.;
    std::thread tm([&] { double counter = 0.; for (uint64_t i = 0; i < n; i++) { counter += 1.; mean = mean + ((double)la[i] - mean) / counter; } });
    ThreadJoiner jm{tm};
    // pushes of whole records (pairs) below 2^31 bases and 2^30 items: one for anything up to ~2 Gbp
    std::vector<std::pair<uint64_t, uint64_t>> pushes;
    if (bases_a + bases_b < (1ull << 31) && n < (1ull << 30)) {
        if (n) pushes.emplace_back(0, n);
    } else {
        if (f2) { lb.resize(n); hip_check(sylph_fastq_lengths(fb.f, 0, n, lb.data()), "sylph_fastq_lengths"); }
        uint64_t i0 = 0, sum = 0;
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t l = (uint64_t)la[i] + (f2 ? lb[i] : 0);
            if (i > i0 && (sum + l >= (1ull << 31) || i - i0 >= (1ull << 30))) { pushes.emplace_back(i0, i - i0); i0 = i; sum = 0; }
            sum += l;
        }
        if (n > i0) pushes.emplace_back(i0, n - i0);
    }
    sylph_sketch* const sk = open_session();
    e.batch.flush(sk);
    // one push: the batch is the session's own buffer, valid until finish — the seeding verdict may wait for it (one host round trip less)
    if (pushes.size() == 1) hip_check(sylph_sketch_set_option(sk, "borrow_until_finish", "1"), "sylph_sketch_set_option");
    for (const auto& p : pushes) {
        const int rc = sylph_sketch_push_fastq(sk, fa.f, f2 ? fb.f : nullptr, p.first, p.second);
        if (rc == SYLPH_ERR_NOMEM) {
            // no room on the device for this route's batch (the whole sample in one piece, up to 2 Gbp): not an error — the host feed, which
            // works through a sample in 256 Mbp batches, takes the sample from its start; the caller opens a fresh session
            if (trace) fprintf(stderr, "[sylph_hip feed] device route: no room for the batch (%s): the host feed takes the sample\n", sylph_last_error());
            tm.join();
            return false;
        }
        hip_check(rc, "sylph_sketch_push_fastq");
    }
    lap("device route: pushed");
    tm.join();
    mean_read_length = mean;
    return true;
}

// The index of the NEXT sample's files is built on a background thread while the current sample is gathered and pushed
// (the files of a sample are independent of everything before them).  get(j) hands over what start(j) began — or builds it
// now; at most `ahead` samples are in flight, so the memory of their mappings and index arrays stays bounded.
class IndexAhead {
   public:
    using Files = std::pair<std::string, std::optional<std::string>>;
    IndexAhead(std::vector<Files> files) : files_(std::move(files)) {}
    ~IndexAhead() { for (auto& kv : fut_) if (kv.second.valid()) kv.second.wait(); }
    void start(size_t j) {
        std::lock_guard<std::mutex> lk(mu_);
        if (j >= files_.size() || fut_.count(j)) return;
        const Files f = files_[j];
        fut_[j] = std::async(std::launch::async, [f] { return index_inputs(f.first, f.second ? &*f.second : nullptr); });
    }
    std::optional<IndexedInput> get(size_t j) {
        std::future<std::optional<IndexedInput>> f;
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = fut_.find(j);
            if (it != fut_.end()) { f = std::move(it->second); fut_.erase(it); }
        }
        if (f.valid()) return f.get();
        const Files& fl = files_.at(j);
        return index_inputs(fl.first, fl.second ? &*fl.second : nullptr);
    }
   private:
    std::vector<Files> files_;
    std::mutex mu_;
    std::map<size_t, std::future<std::optional<IndexedInput>>> fut_;
};

// MemAvailable of /proc/meminfo in bytes (0 if unknown)
size_t mem_available() {
    std::ifstream f("/proc/meminfo");
    std::string key;
    unsigned long long kb = 0;
    while (f >> key) {
        if (key == "MemAvailable:") { f >> kb; return (size_t)kb * 1024; }
        f.ignore(1 << 20, '\n');
    }
    return 0;
}
// what ONE index may hold in anonymous memory when `concurrent` of them can be alive at the same time: half of what is available
void set_feed_budget(size_t concurrent) {
    if (const char* e = getenv("SYLPH_HIP_FEED_MEM_GB")) { set_index_memory_budget((size_t)(atof(e) * (1ull << 30))); return; }
    const size_t avail = mem_available();
    set_index_memory_budget(avail ? avail / 2 / std::max<size_t>(1, concurrent) : 0);
}
}  // namespace

namespace {
// `pre`: the files' index if somebody built it ahead of time (nullptr: build it here; an empty optional: the files are not for
// the block-parallel feed)
std::optional<SequencesSketch> sketch_sequences_needle_impl(Engine& e, const std::string& read_file, uint64_t c, uint64_t k,
                                                            std::optional<std::string> sample_name, bool no_dedup,
                                                            std::optional<IndexedInput>* pre, sylph_sketch** keep = nullptr, bool try_device = false);
std::optional<SequencesSketch> sketch_pair_sequences_impl(Engine& e, const std::string& read_file1, const std::string& read_file2,
                                                          uint64_t c, uint64_t k, std::optional<std::string> sample_name,
                                                          bool no_dedup, double dedup_fpr, std::optional<IndexedInput>* pre,
                                                          sylph_sketch** keep = nullptr, bool try_device = false);
}  // namespace

// sketch.rs:897-959
std::optional<SequencesSketch> sketch_sequences_needle(Engine& e, const std::string& read_file, uint64_t c, uint64_t k,
                                                       std::optional<std::string> sample_name, bool no_dedup) {
    return sketch_sequences_needle_impl(e, read_file, c, k, std::move(sample_name), no_dedup, nullptr);
}
namespace {
std::optional<SequencesSketch> sketch_sequences_needle_impl(Engine& e, const std::string& read_file, uint64_t c, uint64_t k,
                                                            std::optional<std::string> sample_name, bool no_dedup,
                                                            std::optional<IndexedInput>* pre, sylph_sketch** keep, bool try_device) {
    // (pre holds un-indexed text — an inflated gzip file — when its index says neither ok nor failed: the device route takes the text
    //  from memory; if it declines, the index is finished on the host)
    const bool text_only = pre && *pre && !(*pre)->a->ok;
    if (try_device || text_only) {
        std::optional<Session> so;
        double mean = 0.;
        if (sketch_fastq_on_device(e, [&] { so.emplace(e, c, k, false, no_dedup); return so->sk; }, read_file, nullptr, mean, text_only ? &**pre : nullptr)) {
            SequencesSketch out;
            so->finish_or_keep(out, keep);
            out.file_name = read_file; out.c = c; out.k = k; out.paired = false;
            out.sample_name = std::move(sample_name);
            out.mean_read_length = mean;
            return out;
        }
        if (text_only && !finish_index(**pre)) pre->reset();
    }
    std::optional<IndexedInput> own;
    if (!pre) { own = index_inputs(read_file, nullptr); pre = &own; }
    if (auto& in = *pre) {   // uncompressed 4-line FASTQ: block-parallel feed
        std::optional<Session> so;
        double mean = 0.;
        sketch_indexed(e, [&] { so.emplace(e, c, k, false, no_dedup); return so->sk; }, *in, mean);
        Session& s = *so;
        {
            SequencesSketch out;
            s.finish_or_keep(out, keep);
            out.file_name = read_file; out.c = c; out.k = k; out.paired = false;
            out.sample_name = std::move(sample_name);
            out.mean_read_length = mean;
            return out;
        }
    }
    std::unique_ptr<ChunkStream> reader;
    try { reader.reset(new ChunkStream(read_file)); }
    catch (const Error&) { warn(read_file + " is not a valid fasta/fastq file; skipping."); return std::nullopt; }   // :911-914
    Session s(e, c, k, false, no_dedup);
    double mean_read_length = 0., counter = 0.;
    const uint8_t* seq = nullptr;
    uint32_t len = 0;
    for (;;) {
        const ChunkStream::Kind kind = reader->next(seq, len);
        if (kind == ChunkStream::ERR) { warn("File " + read_file + " is not a valid fasta/fastq file"); break; }   // :945
        if (kind == ChunkStream::END) break;
        e.batch.add(s.sk, seq, len, nullptr, 0, false);
        counter += 1.;                                                       // :941-943
        mean_read_length = mean_read_length + ((double)len - mean_read_length) / counter;
    }
    e.batch.flush(s.sk);
    SequencesSketch out;
    s.finish_or_keep(out, keep);
    out.file_name = read_file; out.c = c; out.k = k; out.paired = false;
    out.sample_name = std::move(sample_name);
    out.mean_read_length = mean_read_length;
    return out;
}
}  // namespace

// sketch.rs:771-895.  dedup_fpr == 0: the exact marker set (:829-838); else the set behind a cuckoo filter of that false-positive
// probability (:839-848; the session option "dedup_fpr", csrc/a10.hip).
std::optional<SequencesSketch> sketch_pair_sequences(Engine& e, const std::string& read_file1, const std::string& read_file2,
                                                     uint64_t c, uint64_t k, std::optional<std::string> sample_name,
                                                     bool no_dedup, double dedup_fpr) {
    return sketch_pair_sequences_impl(e, read_file1, read_file2, c, k, std::move(sample_name), no_dedup, dedup_fpr, nullptr);
}
namespace {
std::optional<SequencesSketch> sketch_pair_sequences_impl(Engine& e, const std::string& read_file1, const std::string& read_file2,
                                                          uint64_t c, uint64_t k, std::optional<std::string> sample_name,
                                                          bool no_dedup, double dedup_fpr, std::optional<IndexedInput>* pre,
                                                          sylph_sketch** keep, bool try_device) {
    const bool text_only = pre && *pre && !(*pre)->a->ok;
    if (try_device || text_only) {
        std::optional<Session> so;
        double mean = 0.;
        if (sketch_fastq_on_device(e, [&] { so.emplace(e, c, k, true, no_dedup, dedup_fpr); return so->sk; }, read_file1, &read_file2, mean, text_only ? &**pre : nullptr)) {
            SequencesSketch out;
            so->finish_or_keep(out, keep);
            out.file_name = read_file1; out.c = c; out.k = k; out.paired = true;
            out.sample_name = std::move(sample_name);
            out.mean_read_length = mean;
            return out;
        }
        if (text_only && !finish_index(**pre)) pre->reset();
    }
    std::optional<IndexedInput> own;
    if (!pre) { own = index_inputs(read_file1, &read_file2); pre = &own; }
    if (auto& in = *pre) {
        std::optional<Session> so;
        double mean = 0.;
        sketch_indexed(e, [&] { so.emplace(e, c, k, true, no_dedup, dedup_fpr); return so->sk; }, *in, mean);
        Session& s = *so;
        {
            SequencesSketch out;
            s.finish_or_keep(out, keep);
            out.file_name = read_file1; out.c = c; out.k = k; out.paired = true;
            out.sample_name = std::move(sample_name);
            out.mean_read_length = mean;
            return out;
        }
    }
    std::unique_ptr<ChunkStream> r1, r2;   // the two mate files are parsed/inflated concurrently by their reader threads
    try { r1.reset(new ChunkStream(read_file1)); r2.reset(new ChunkStream(read_file2)); }
    catch (const Error&) {
        throw Error{1, "Paired end reading failed for '" + read_file1 + "' and '" + read_file2 +
                           "'. Make sure the files are present or the sequences are valid."};   // :781-784
    }
    Session s(e, c, k, true, no_dedup, dedup_fpr);
    double mean_read_length = 0., counter = 0.;
    const uint8_t *s1 = nullptr, *s2 = nullptr;
    uint32_t l1 = 0, l2 = 0;
    for (;;) {
        const ChunkStream::Kind k1 = r1->next(s1, l1);
        if (k1 == ChunkStream::ERR) { e.batch.flush(s.sk); return std::nullopt; }   // :878-880
        const ChunkStream::Kind k2 = r2->next(s2, l2);
        if (k1 == ChunkStream::END) break;                                   // :881-883
        if (k2 != ChunkStream::REC) continue;                                // mate 2 missing/invalid: pair skipped
        e.batch.add(s.sk, s1, l1, s2, l2, true);
        counter += 1.;                                                       // :824-826 (mate-1 length only)
        mean_read_length = mean_read_length + ((double)l1 - mean_read_length) / counter;
    }
    e.batch.flush(s.sk);
    SequencesSketch out;
    s.finish_or_keep(out, keep);
    out.file_name = read_file1; out.c = c; out.k = k; out.paired = true;
    out.sample_name = std::move(sample_name);
    out.mean_read_length = mean_read_length;
    return out;
}
}  // namespace

// A batch of parsed genomes sketched by ONE sylph_sketch_genomes call (seeding, genome-wide duplicate removal and the spacing
// filter all run on the device; sketch.rs:550-622 / :481-548 per genome).  Files are parsed on the host and appended until
// the batch holds BATCH_BASES; results are split back into GenomeSketch records in input order.
namespace {
struct GenomeBatch {
    static constexpr uint64_t BATCH_BASES = 1ull << 30;
    Engine& e;
    uint64_t c, k, min_spacing;
    bool pseudotax;
    std::vector<GenomeSketch>& out;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> off{0}, goff{0};
    std::vector<GenomeSketch> pending;   // names + gn_size of the genomes in the batch
    GenomeBatch(Engine& en, uint64_t c_, uint64_t k_, uint64_t sp, bool pt, std::vector<GenomeSketch>& o)
        : e(en), c(c_), k(k_), min_spacing(sp), pseudotax(pt), out(o) {}

    // One genome file read into memory: records concatenated, one offset per record.  Touches nothing of the batch, so any
    // number of files can be parsed (and inflated) at the same time (the reference reads its genome files on the rayon pool,
    // sketch.rs:422-476); warnings travel with the result so that they come out in file order.
    struct Parsed {
        std::string file;
        std::vector<uint8_t> bases;
        std::vector<uint64_t> ends;                 // end of every record in `bases`
        std::vector<std::string> ids;               // first record only unless `individual`
        std::vector<std::string> warnings;
        bool ok = false;
    };
    static Parsed parse_file(const std::string& ref_file, bool individual) {
        Parsed p;
        p.file = ref_file;
        std::unique_ptr<FastxReader> reader;
        try { reader.reset(new FastxReader(ref_file)); }
        catch (const Error&) { p.warnings.push_back(ref_file + " is not a valid fasta/fastq file; skipping."); return p; }
        FastxRecord rec;
        try {
            while (reader->next(rec)) {
                if (individual || p.ids.empty()) p.ids.push_back(rec.id);
                p.bases.insert(p.bases.end(), rec.seq.begin(), rec.seq.end());
                p.ends.push_back(p.bases.size());
            }
        } catch (const Error&) {                                             // :586-589: the whole file is dropped
            p.warnings.push_back("File " + ref_file + " is not a valid fasta/fastq file");
            return p;
        }
        p.ok = true;
        return p;
    }

    // sketch_genome (individual = false) or sketch_genome_individual (true) up to the k-mer work; false = file skipped
    bool add_file(const std::string& ref_file, bool individual) { return append(parse_file(ref_file, individual), individual); }

    // files [0, n) parsed on up to `threads` threads, a window of them at a time, appended (and flushed) in file order
    void add_files(const std::vector<std::string>& files, bool individual, uint64_t threads) {
        const size_t window = std::max<size_t>(1, std::min<size_t>(threads, 64)) * 2;
        for (size_t lo = 0; lo < files.size(); lo += window) {
            const size_t n = std::min(window, files.size() - lo);
            std::vector<Parsed> parsed(n);
            std::atomic<size_t> next{0};
            auto work = [&] { for (size_t i = next++; i < n; i = next++) parsed[i] = parse_file(files[lo + i], individual); };
            std::vector<std::thread> pool;
            for (size_t w = 1; w < std::min<size_t>(std::max<uint64_t>(threads, 1), n); w++) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
            for (auto& p : parsed) append(std::move(p), individual);
        }
    }

    bool append(Parsed p, bool individual) {
        for (const auto& w : p.warnings) warn(w);
        if (!p.ok) return false;
        const std::string& ref_file = p.file;
        // one sylph_sketch_genomes call holds < 2^32 bases: a file that would push the batch over the limit starts a new batch,
        // and a single genome beyond it is skipped with a warning instead of aborting the whole run
        constexpr uint64_t LIMIT = (1ull << 32) - 4096;
        if (p.bases.size() >= LIMIT) {
            warn(ref_file + " holds " + std::to_string(p.bases.size()) + " bases, more than one device batch (2^32): skipping it");
            return false;
        }
        if (bases.size() + p.bases.size() >= LIMIT) flush();
        const uint64_t bases0 = bases.size();
        bases.insert(bases.end(), p.bases.begin(), p.bases.end());
        for (size_t r = 0; r < p.ends.size(); r++) {
            off.push_back(bases0 + p.ends[r]);
            if (individual) {
                GenomeSketch g;
                g.file_name = ref_file; g.first_contig_name = p.ids[r]; g.gn_size = p.ends[r] - (r ? p.ends[r - 1] : 0);
                pending.push_back(std::move(g));
                goff.push_back(off.size() - 1);
            }
        }
        if (!individual) {
            GenomeSketch whole;
            whole.file_name = ref_file;
            if (!p.ids.empty()) whole.first_contig_name = p.ids.front();
            whole.gn_size = p.bases.size();
            pending.push_back(std::move(whole));
            goff.push_back(off.size() - 1);
        }
        if (bases.size() >= BATCH_BASES) flush();
        return true;
    }

    void flush() {
        if (pending.empty()) return;
        const uint64_t G = pending.size();
        std::vector<uint64_t> koff(G + 1), toff(G + 1);
        uint64_t *gk = nullptr, *tr = nullptr;
        hip_check(sylph_sketch_genomes(e.context(), bases.data(), off.data(), off.size() - 1, goff.data(), G, (uint32_t)c, (uint32_t)k,
                                       SYLPH_SEED_AVX2_COMPAT, min_spacing, pseudotax ? 1 : 0, SYLPH_MEM_HOST, &gk, koff.data(), &tr,
                                       toff.data()),
                  "sylph_sketch_genomes");
        struct Free { uint64_t* p; ~Free() { sylph_free(p); } } f1{gk}, f2{tr};
        for (uint64_t g = 0; g < G; g++) {
            GenomeSketch& s = pending[g];
            s.genome_kmers.assign(gk + koff[g], gk + koff[g + 1]);
            if (pseudotax) s.pseudotax_tracked_nonused_kmers = std::vector<uint64_t>(tr + toff[g], tr + toff[g + 1]);
            s.c = c; s.k = k; s.min_spacing = min_spacing;
            out.push_back(std::move(s));
        }
        pending.clear(); bases.clear(); off.assign(1, 0); goff.assign(1, 0);
    }
};
}  // namespace

// sketch.rs:550-622
std::optional<GenomeSketch> sketch_genome(Engine& e, uint64_t c, uint64_t k, const std::string& ref_file, uint64_t min_spacing,
                                          bool pseudotax) {
    std::vector<GenomeSketch> out;
    GenomeBatch b(e, c, k, min_spacing, pseudotax, out);
    if (!b.add_file(ref_file, false)) return std::nullopt;
    b.flush();
    return std::move(out.front());
}

// sketch.rs:481-548
std::vector<GenomeSketch> sketch_genome_individual(Engine& e, uint64_t c, uint64_t k, const std::string& ref_file,
                                                   uint64_t min_spacing, bool pseudotax) {
    std::vector<GenomeSketch> out;
    GenomeBatch b(e, c, k, min_spacing, pseudotax, out);
    b.add_file(ref_file, true);
    b.flush();
    return out;
}

// sketch.rs:276-479
int sketch(Engine& e, const SketchArgs& args) {
    std::vector<std::string> read_inputs, genome_inputs, first_pairs, second_pairs;
    const bool nothing = args.files.empty() && !args.list_sequence && args.first_pair.empty() && args.second_pair.empty() &&
                         args.genomes.empty() && args.reads.empty() && !args.list_genomes && !args.list_reads &&
                         !args.list_first_pair && !args.list_second_pair;
    if (nothing) throw Error{1, "No input sequences found; see sylph sketch -h for help. Exiting."};   // :144-157
    if (args.fpr < 0. || args.fpr >= 1.) throw Error{1, "Invalid FPR for sketching. Must be in [0,1)."};   // :158-161
    std::vector<std::string> all_files;
    if (args.list_sequence) parse_line_file(*args.list_sequence, all_files);
    all_files.insert(all_files.end(), args.files.begin(), args.files.end());
    for (const auto& f : all_files) {                                        // :164-189
        if (is_fastq(f)) read_inputs.push_back(f);
        else if (is_fasta(f)) genome_inputs.push_back(f);
        else warn(f + " does not have a fasta/fastq/gzip type extension; skipping");
    }
    genome_inputs.insert(genome_inputs.end(), args.genomes.begin(), args.genomes.end());   // :191-216
    read_inputs.insert(read_inputs.end(), args.reads.begin(), args.reads.end());
    if (args.list_reads) parse_line_file(*args.list_reads, read_inputs);
    if (args.list_genomes) parse_line_file(*args.list_genomes, genome_inputs);
    if (args.first_pair.size() != args.second_pair.size()) throw Error{1, "Different number of paired sequences. Exiting."};
    first_pairs = args.first_pair;
    second_pairs = args.second_pair;
    if (args.list_first_pair) parse_line_file(*args.list_first_pair, first_pairs);
    if (args.list_second_pair) parse_line_file(*args.list_second_pair, second_pairs);
    if (first_pairs.size() != second_pairs.size()) throw Error{1, "Different number of paired sequences. Exiting."};
    std::optional<std::vector<std::string>> sample_names;                    // :260-274
    if (args.list_sample_names) { sample_names.emplace(); parse_line_file(*args.list_sample_names, *sample_names); }
    else if (args.sample_names) sample_names = args.sample_names;
    if (sample_names && sample_names->size() != first_pairs.size() + read_inputs.size())
        throw Error{1, "Sample name length is not equal to the number of reads. Exiting"};   // :288-292
    // a10: pairs are deduplicated as the reference does — --fpr != 0 (default 1e-4, cmdline.rs:77): the set behind a cuckoo filter
    // (sketch.rs:733-769; the session option "dedup_fpr"); --fpr 0: the exact set (:690-731).  --exact-dedup / SYLPH_HIP_EXACT_DEDUP=1
    // (not in the reference) force the exact set whatever --fpr says.  (--no-dedup never consults the filter: sketch.rs:744.)
    const double pair_fpr = exact_dedup_accepted(args.exact_dedup) ? 0. : args.fpr;

    // Samples are independent (sketch.rs:313,371 runs them on the rayon pool, `-t`): a pool of `-t` worker threads, each with
    // its own GPU context (calls on one context are serialised) and its own page-locked batch, takes them in input order.
    // The parsing / inflating of different samples overlaps; the GPU work of one sample is ~2 ms per Gbp.
    create_dir_all(args.sample_output_dir);
    const size_t n_jobs = first_pairs.size() + read_inputs.size();
    // (a command's first sample is gathered into pageable memory while the GPU runtime comes up and the device route's uploader serves the
    //  samples BEHIND it: with one sample nobody ever wants the page-locked feed buffers — ~80 ms of hipHostMalloc beside the sample's own
    //  push, and as much again when the process is torn down)
    if (n_jobs <= 1) e.defer_pinned.store(true);
    // --gpus N|all (round 6): the workers are dealt to the node's GPUs — worker w runs on device w mod N, with its own context, page-locked
    // batch and uploader there; a sample never leaves its GPU, nothing is exchanged ("replicas only" for the sketch stage —
    // what the reference's rayon pool does with the machine's cores, sketch.rs:313, :371).  At least one worker per GPU.
    int n_gpus = 1;
    if (args.gpus != 1) {
        const int have = std::max(1, sylph_device_count());
        n_gpus = args.gpus < 0 ? have : std::min(args.gpus, have);
        if (args.gpus > have) warn("--gpus " + std::to_string(args.gpus) + ": this node has " + std::to_string(have) + " GPU(s); using them all");
    }
    // (tests on a one-GPU box: SYLPH_HIP_FAKE_GPUS=N deals the workers as for N GPUs and maps every one of them to device 0)
    const char* fake = getenv("SYLPH_HIP_FAKE_GPUS");
    const int n_deal = fake ? std::max(1, atoi(fake)) : n_gpus;
    const size_t n_workers = std::max<size_t>(1, std::min<size_t>(std::max<size_t>(std::min<size_t>(args.threads, MAX_SAMPLE_THREADS), (size_t)(args.gpus != 1 || fake ? n_deal : 1)), n_jobs));
    set_parse_share((unsigned)n_workers);
    // the next sample of every worker is indexed while the current one is gathered and pushed
    std::vector<IndexAhead::Files> job_files;
    for (size_t j = 0; j < first_pairs.size(); j++) job_files.push_back({first_pairs[j], second_pairs[j]});
    for (const auto& r : read_inputs) job_files.push_back({r, std::nullopt});
    IndexAhead ahead(job_files);
    set_feed_budget(4 * n_workers);            // two files per sample, the current and the next sample of every worker
    std::atomic<size_t> indexes_obtained{0};
    set_no_more_inflates(false);
    // a finished sketch is written by the writers' thread while its worker takes the next sample; the first error is rethrown by the command
    std::mutex write_mu;
    std::optional<Error> write_error;
    auto write_out = [&](const std::string& path, SequencesSketch&& sk, const std::string& what, std::function<void(const SequencesSketch&, const std::string&)> timing) {
        auto keep = std::make_shared<SequencesSketch>(std::move(sk));
        auto task = [&write_mu, &write_error, keep, path, what, timing] {
            try {
                write_sylsp(path, *keep);
                trace_mark("sketch: .sylsp written");
                info("Sketching " + path + " complete.");
                timing(*keep, what);
            } catch (const Error& er) {
                std::lock_guard<std::mutex> lk(write_mu);
                if (!write_error) write_error = er;
            }
        };
        if (n_jobs > 1) writers().add(task); else task();
    };
    struct DrainWriters { ~DrainWriters() { writers().drain(); } } drain_writers;     // (also on the way out of an exception: the tasks refer to this frame)
    auto run_job = [&](Engine& eng, size_t j) {
        // an engine that is up takes plain FASTQ by the device route (no host index at all); its first sample, whose index is built while
        // the GPU runtime initialises, and everything the device route declines go the host way
        const auto& jf = job_files[j];
        const bool gz = device_feed_enabled() && is_gzip_file(jf.first);
        const bool gz_dev = gz && device_inflate_enabled();                      // round 6: the compressed bytes travel, the device inflates
        const bool dev = device_feed_enabled() && (gz_dev || (eng.ready() && !gz));
        std::optional<IndexedInput> pre;
        if (gz && !gz_dev) pre = index_inputs(jf.first, jf.second ? &*jf.second : nullptr, false);   // inflated on the host, its text then sent as it is
        else if (!dev) pre = ahead.get(j);
        // (a gzip sample on the device route needs none of the page-locked feed buffers: an engine still in its bring-up leaves them to
        //  whoever wants them first — ~80 ms of hipHostMalloc that would run beside the sample's own allocations and copies)
        if (gz_dev && !eng.ready()) { eng.defer_pinned.store(true); eng.warm_text_route.store(true); }
        trace_mark(dev ? "sketch: the sample goes the device route" : "sketch: the sample's files are indexed (or not indexable)");
        if (++indexes_obtained == n_jobs) set_no_more_inflates(true);   // nobody will want a recycled inflate buffer any more
        // the index goes (2 x 1 GB of mappings to unmap / inflated copies to hand back: 30-60 ms per sample) on a thread of its own, behind the sample
        struct Later { std::optional<IndexedInput>& p; ~Later() { if (p) background([x = std::make_shared<std::optional<IndexedInput>>(std::move(p))]() mutable { x.reset(); }); } } later{pre};
        if (!device_feed_enabled()) ahead.start(j + n_workers);
        const auto t_job = std::chrono::steady_clock::now();
        auto timing = [t_job](const SequencesSketch& sk, const std::string& what) {   // (not a reference message: feed measurements)
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_job).count();
            uint64_t occ = 0;
            for (uint32_t c : sk.counts) occ += c;
            char b[256];
            snprintf(b, sizeof(b), "timing: %s sketched + written in %.3f s (%zu distinct k-mers, %llu counted occurrences)", what.c_str(), sec,
                     sk.kmers.size(), (unsigned long long)occ);
            info(b);
        };
        if (j < first_pairs.size()) {                                        // :311-367
            std::optional<std::string> sample_name;
            if (sample_names) sample_name = (*sample_names)[j];
            auto sk = sketch_pair_sequences_impl(eng, first_pairs[j], second_pairs[j], args.c, args.k, sample_name, args.no_dedup, pair_fpr, dev ? nullptr : &pre, nullptr, dev);
            if (!sk) return;
            const std::string& name = sample_name ? *sk->sample_name : sk->file_name;
            const std::string path = path_join(args.sample_output_dir, basename_of(name)) + ".paired" + SAMPLE_FILE_SUFFIX;
            trace_mark("sketch: the pair is sketched (table on the host)");
            write_out(path, std::move(*sk), first_pairs[j], timing);
        } else {                                                             // :369-420
            const size_t i = j - first_pairs.size();
            std::optional<std::string> sample_name;
            if (sample_names) sample_name = (*sample_names)[j];
            auto sk = sketch_sequences_needle_impl(eng, read_inputs[i], args.c, args.k, sample_name, args.no_dedup, dev ? nullptr : &pre, nullptr, dev);
            if (!sk) return;
            const std::string& name = sample_name ? *sk->sample_name : sk->file_name;
            const std::string path = path_join(args.sample_output_dir, basename_of(name)) + SAMPLE_FILE_SUFFIX;
            write_out(path, std::move(*sk), read_inputs[i], timing);
        }
    };
    if (n_workers <= 1) {
        for (size_t j = 0; j < n_jobs; j++) run_job(e, j);
    } else {
        std::atomic<size_t> next{0};
        std::mutex err_mu;
        std::optional<Error> first_error;
        std::atomic<size_t> worker_no{1};
        auto worker = [&](Engine* eng) {
            try {
                std::unique_ptr<Engine> own;
                if (!eng) {
                    const size_t w = worker_no++;
                    const int dev = n_deal > 1 ? (fake ? e.device : (int)(w % (size_t)n_deal)) : e.device;
                    if (n_deal > 1) info("sketch worker " + std::to_string(w) + " runs on GPU " + std::to_string(fake ? (int)(w % (size_t)n_deal) : dev) + (fake ? " (SYLPH_HIP_FAKE_GPUS: device 0)" : ""));
                    own.reset(new Engine(dev));
                    eng = own.get();
                }
                for (size_t j = next++; j < n_jobs; j = next++) run_job(*eng, j);
            } catch (const Error& er) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (!first_error) first_error = er;
                next = n_jobs;   // stop handing out work
            }
        };
        std::vector<std::thread> pool;
        for (size_t w = 1; w < n_workers; w++) pool.emplace_back(worker, nullptr);
        worker(&e);
        for (auto& t : pool) t.join();
        if (first_error) throw *first_error;
    }
    writers().drain();
    if (write_error) throw *write_error;
    if (!genome_inputs.empty()) {                                            // :422-476
        const std::string path = args.db_out_name + QUERY_FILE_SUFFIX;
        create_dir_all(dirname_of(path));
        std::vector<GenomeSketch> all;
        GenomeBatch batch(e, args.c, args.k, args.min_spacing_kmer, !args.no_pseudotax, all);
        batch.add_files(genome_inputs, args.individual, args.threads);
        batch.flush();
        if (all.empty()) warn("No valid genomes to sketch; " + path + " is not output");
        else { write_syldb(path, all); info("Wrote all genome sketches to " + path); }
    }
    if (!fast_exit()) join_background();
    info("Finished.");
    return 0;
}

// ---- contain ----------------------------------------------------------------------------------------------------

namespace {

// contain.rs:18-94
void print_ani_result(const AniResult& r, const std::string& seq_name, const GenomeSketch& g, bool pseudotax, FILE* out,
                      bool debug_f64 = false) {
    if (debug_f64) {   // --debug-f64 (not in the reference): every float column as %.17g, unclamped, for 1e-6 parity checks
        auto opt = [](const std::optional<double>& v) { char b[40]; if (v) snprintf(b, sizeof(b), "%.17g", *v); else snprintf(b, sizeof(b), "NA"); return std::string(b); };
        fprintf(out, "%s\t%s\t", seq_name.c_str(), g.file_name.c_str());
        if (pseudotax) fprintf(out, "%.17g\t%.17g\t", *r.rel_abund, *r.seq_abund);
        fprintf(out, "%.17g\t%.17g\t%s-%s\t", r.final_est_ani * 100., r.final_est_cov, opt(r.ani_ci_lo).c_str(), opt(r.ani_ci_hi).c_str());
        if (r.lambda_status == AdjustStatus::Lambda) fprintf(out, "%.17g\t", r.lambda);
        else fprintf(out, "%s\t", r.lambda_status == AdjustStatus::High ? "HIGH" : "LOW");
        fprintf(out, "%s-%s\t%.17g\t%.17g\t%zu/%zu\t%.17g\t", opt(r.lambda_ci_lo).c_str(), opt(r.lambda_ci_hi).c_str(), r.median_cov,
                r.mean_cov, r.contain_count, r.n_kmers, r.naive_ani * 100.);
        if (pseudotax) fprintf(out, "%zu\t", *r.kmers_lost);
        fprintf(out, "%s\n", g.first_contig_name.c_str());
        return;
    }
    char final_ani[64];
    snprintf(final_ani, sizeof(final_ani), "%.2f", std::min(r.final_est_ani * 100., 100.));
    char lambda_print[64];
    if (r.lambda_status == AdjustStatus::Lambda) snprintf(lambda_print, sizeof(lambda_print), "%.3f", r.lambda);
    else snprintf(lambda_print, sizeof(lambda_print), "%s", r.lambda_status == AdjustStatus::High ? "HIGH" : "LOW");
    char ci_ani[64] = "NA-NA", ci_lambda[64] = "NA-NA";
    if (r.ani_ci_lo && r.ani_ci_hi) snprintf(ci_ani, sizeof(ci_ani), "%.2f-%.2f", *r.ani_ci_lo * 100., *r.ani_ci_hi * 100.);
    if (r.lambda_ci_lo && r.lambda_ci_hi) snprintf(ci_lambda, sizeof(ci_lambda), "%.2f-%.2f", *r.lambda_ci_lo, *r.lambda_ci_hi);
    if (!pseudotax) {
        fprintf(out, "%s\t%s\t%s\t%.3f\t%s\t%s\t%s\t%.0f\t%.3f\t%zu/%zu\t%.2f\t%s\n", seq_name.c_str(), g.file_name.c_str(),
                final_ani, r.final_est_cov, ci_ani, lambda_print, ci_lambda, r.median_cov, r.mean_cov, r.contain_count, r.n_kmers,
                r.naive_ani * 100., g.first_contig_name.c_str());
    } else {
        fprintf(out, "%s\t%s\t%.4f\t%.4f\t%s\t%.3f\t%s\t%s\t%s\t%.0f\t%.3f\t%zu/%zu\t%.2f\t%zu\t%s\n", seq_name.c_str(),
                g.file_name.c_str(), *r.rel_abund, *r.seq_abund, final_ani, r.final_est_cov, ci_ani, lambda_print, ci_lambda,
                r.median_cov, r.mean_cov, r.contain_count, r.n_kmers, r.naive_ani * 100., *r.kmers_lost,
                g.first_contig_name.c_str());
    }
}

void print_header(bool pseudotax, FILE* out, bool estimate_unknown) {         // contain.rs:461-480
    if (!pseudotax)
        fprintf(out, "Sample_file\tGenome_file\tAdjusted_ANI\tEff_cov\tANI_5-95_percentile\tEff_lambda\tLambda_5-95_percentile\t"
                     "Median_cov\tMean_cov_geq1\tContainment_ind\tNaive_ANI\tContig_name\n");
    else
        fprintf(out, "Sample_file\tGenome_file\tTaxonomic_abundance\tSequence_abundance\tAdjusted_ANI\t%s\tANI_5-95_percentile\t"
                     "Eff_lambda\tLambda_5-95_percentile\tMedian_cov\tMean_cov_geq1\tContainment_ind\tNaive_ANI\tkmers_reassigned\t"
                     "Contig_name\n", estimate_unknown ? "True_cov" : "Eff_cov");
}

}  // namespace

// --estimate-unknown (-u), contain.rs:901-951 get_kmer_identity: k-mer identity of the reads (identity ^ k) from the share of
// multiplicity-1 k-mers.  The estimate itself (eps) is a sum over the table.  The reference's "continuous median" of the counts
// above 1 is a walk over `kmer_counts.values()` in hashbrown's iteration order, which this host does not reproduce: here the
// walk goes over the table in ascending k-mer order (an arbitrary order with respect to the counts, as the hash map's is).
// It only decides whether a short-read sample counts as "depth < 3" and gets the fixed 99.5 % identity — a sample right at
// that limit may take the other branch than `sylph profile -u` does; -I (contain.rs:275) bypasses the walk altogether.
// The integer types are the reference's: `num_not1s` is a u32 that wraps in a release build.
std::optional<double> get_kmer_identity(const SequencesSketch& S, bool estimate_unknown) {
    if (!estimate_unknown) return std::nullopt;
    uint32_t median = 0;
    double mov_avg_median = 0., n = 1.;
    for (const uint32_t count : S.counts) {
        if (count > 1) {
            if (count > median) median += 1; else median -= 1;
            mov_avg_median += (double)median;
            n += 1.;
        }
    }
    mov_avg_median /= n;
    int32_t num_1s = 0;
    uint32_t num_not1s = 0;
    for (const uint32_t count : S.counts) {
        if (count == 1) num_1s += 1; else num_not1s += count;
    }
    const double eps = (double)num_not1s / ((double)num_not1s + (double)num_1s + 0.1);
    {   // which branch the walk chose, and how close the call was (the walk's order is this host's, not hashbrown's: see above).
        // The reference prints neither line: the first only under SYLPH_HIP_DEBUG, the warning only inside the 10 % band.
        const bool near_limit = S.mean_read_length < 400. && std::fabs(mov_avg_median - MED_KMER_FOR_ID_EST) <= 0.1 * MED_KMER_FOR_ID_EST;
        if (getenv("SYLPH_HIP_DEBUG")) {
            char num[160];
            snprintf(num, sizeof(num), " --estimate-unknown: running median of the counts above 1 = %.4f (limit %.1f, mean read length %.1f): ",
                     mov_avg_median, MED_KMER_FOR_ID_EST, S.mean_read_length);
            info(S.file_name + num + ((mov_avg_median < MED_KMER_FOR_ID_EST && S.mean_read_length < 400.) ? "fixed 99.5% read identity" : "read identity estimated from the table"));
        }
        if (near_limit)
            warn(S.file_name + ": the sample's depth is within 10% of the limit that switches --estimate-unknown between the fixed 99.5% read identity and "
                 "the estimated one; that decision depends on the order the table is walked in (ascending k-mers here, hash-map order in "
                 "sylph), so True_cov / Sequence_abundance may differ from `sylph profile -u` for this sample: pass -I to fix the identity");
    }
    if (mov_avg_median < MED_KMER_FOR_ID_EST && S.mean_read_length < 400.) {
        info(S.file_name + " short-read sample has high diversity compared to sequencing depth (approx. avg depth < 3). Using 99.5% as "
             "read accuracy estimate instead of automatic detection for --estimate-unknown.");
        return std::pow(0.995, (double)S.k);
    }
    return eps < 1. ? eps : 1.;
}

// contain.rs:377-390
void estimate_true_cov(std::vector<AniResult>& results, std::optional<double> kmer_id_opt, bool estimate_unknown, double read_length,
                       uint64_t k) {
    double multiplier = 1.;
    if (estimate_unknown) multiplier = read_length / (read_length - (double)k + 1.);
    if (estimate_unknown && kmer_id_opt)
        for (auto& r : results) r.final_est_cov = r.final_est_cov / *kmer_id_opt * multiplier;
}

// contain.rs:392-408: share of the sample's bases the profiled genomes account for
double estimate_covered_bases(const std::vector<AniResult>& results, const std::vector<GenomeSketch>& genomes, const SequencesSketch& S,
                              double read_length, uint64_t k) {
    const double multiplier = read_length / (read_length - (double)k + 1.);
    double num_covered_bases = 0.;
    for (const auto& r : results) num_covered_bases += (double)genomes[r.genome_index].gn_size * r.final_est_cov;
    uint64_t num_total_counts = 0;
    for (const uint32_t count : S.counts) num_total_counts += count;
    const double num_tentative_bases = (double)(S.c * num_total_counts) * multiplier;
    if (num_tentative_bases == 0.) return 0.;
    return std::min(num_covered_bases / num_tentative_bases, 1.);
}

// contain.rs:115-351
int contain(Engine& e, ContainCmdArgs args, bool pseudotax_in, FILE* out) {
    if (pseudotax_in) args.pseudotax = true;
    std::vector<std::string> genome_sketch_files, genome_files, read_sketch_files;
    std::vector<std::vector<std::string>> read_files;
    std::vector<std::string> all_files = args.files;
    if (args.file_list) parse_line_file(*args.file_list, all_files);
    auto ends = [](const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; };
    for (const auto& f : all_files) {                                        // contain.rs:165-198
        if (ends(f, ".syldb") || ends(f, ".sylqueries")) genome_sketch_files.push_back(f);
        else if (ends(f, ".sylsp") || ends(f, ".sylsample")) read_sketch_files.push_back(f);
        else if (is_fasta(f)) genome_files.push_back(f);
        else if (is_fastq(f)) read_files.push_back({f});
        else warn(f + " file extension is not a sketch or a fasta/fastq file.");
    }
    if (args.first_pair.size() != args.second_pair.size())
        throw Error{1, "Different number of paired sequences (-1, -2) for sketching. Exiting."};
    for (size_t i = 0; i < args.first_pair.size(); i++) read_files.push_back({args.first_pair[i], args.second_pair[i]});
    for (const auto& r : args.reads) read_files.push_back({r});
    if (genome_sketch_files.empty() && genome_files.empty())
        throw Error{1, "No genome files found; see sylph query/profile -h for help. Exiting"};
    if (read_sketch_files.empty() && read_files.empty())
        throw Error{1, "No read files found; see sylph query/profile -h for help. Exiting"};

    // get_genome_sketches, contain.rs:482-541
    std::vector<GenomeSketch> genome_sketches;
    std::optional<uint64_t> lowest_genome_c, current_k;
    static const bool copy_load = getenv("SYLPH_HIP_DB_COPY_LOAD") != nullptr;   // A/B + tests: every genome copied into vectors first (rounds 1-3)
    for (const auto& f : genome_sketch_files) {
        auto v = copy_load ? read_syldb(f) : read_syldb_views(f);
        if (v.empty()) continue;
        const uint64_t c = v.front().c, k = v.front().k;
        if (!lowest_genome_c || *lowest_genome_c < c) lowest_genome_c = c;
        if (!current_k) current_k = k;
        else if (*current_k != k) throw Error{1, "Query sketches have inconsistent -k. Exiting."};
        for (auto& g : v) genome_sketches.push_back(std::move(g));
    }
    GenomeBatch batch(e, args.c, args.k, args.min_spacing_kmer, args.pseudotax, genome_sketches);
    std::vector<std::string> genome_files_ok;
    for (const auto& gf : genome_files) {
        if (lowest_genome_c && *lowest_genome_c < args.c) { fprintf(stderr, "ERROR [sylph_hip] Value of -c for contain is %llu -- greater than the smallest value of -c for a genome sketch %llu. Continuing without sketching.\n", (unsigned long long)args.c, (unsigned long long)*lowest_genome_c); continue; }
        if (current_k && *current_k != args.k) { fprintf(stderr, "ERROR [sylph_hip] -k %llu is not equal to -k %llu found in sketches. Continuing without sketching.\n", (unsigned long long)args.k, (unsigned long long)*current_k); continue; }
        genome_files_ok.push_back(gf);
    }
    batch.add_files(genome_files_ok, args.individual, args.threads);
    batch.flush();
    info("Finished obtaining genome sketches.");
    if (genome_sketches.empty()) throw Error{1, "No genome sketches found; see sylph query/profile -h for help. Exiting"};
    if (!genome_sketches.front().has_tracked() && args.pseudotax)
        throw Error{1, "Attempting profiling, but *.syldb was sketched with the --disable-profiling option. Exiting"};   // :234-237

    // database resident in HBM (replaces the per-genome probe loop of contain.rs:284-291).  The k-mers go from where they lie — the
    // mapped .syldb files (views), or the vectors of genomes sketched in this run — through the library's page-locked upload chunks
    // to the device, gathered by all parse threads: no per-genome vector, no flat host copy, the copy of one chunk travels while the
    // next is gathered (round 4; rounds 1-3: 113,104 vectors + one 11 GB concatenation + a staged pageable copy)
    const auto t_db0 = std::chrono::steady_clock::now();
    auto upload_gathered = [&](bool tracked, std::vector<uint64_t>& off, sylph_upload** up_out) -> const uint64_t* {
        off.assign(1, 0);
        for (const auto& g : genome_sketches) off.push_back(off.back() + (tracked ? g.n_tracked() : g.n_kmers()));
        const uint64_t total = off.back() * 8;
        sylph_upload* up = nullptr;
        hip_check(sylph_upload_begin(e.context(), total, 256ull << 20, &up), "sylph_upload_begin");
        *up_out = up;
        const unsigned T = std::max(1u, parse_threads());
        uint64_t at = 0;                                   // bytes of the flat array gathered so far
        size_t g0 = 0;                                     // first genome that still has bytes to give
        while (at < total) {
            void* chunk = nullptr;
            uint64_t cap = 0;
            hip_check(sylph_upload_chunk(up, &chunk, &cap), "sylph_upload_chunk");
            const uint64_t n = std::min<uint64_t>(cap & ~7ull, total - at);
            std::vector<std::thread> th;
            std::exception_ptr err;
            auto piece = [&](unsigned w) {
                const uint64_t b0 = at + n / T * w / 8 * 8, b1 = w + 1 == T ? at + n : at + n / T * (w + 1) / 8 * 8;
                size_t g = (size_t)(std::upper_bound(off.begin() + (long)g0, off.end(), b0 / 8) - off.begin()) - 1;   // genome holding byte b0
                for (uint64_t b = b0; b < b1;) {
                    while (off[g + 1] * 8 <= b) g++;
                    const GenomeSketch& gs = genome_sketches[g];
                    const uint8_t* src = tracked ? gs.tracked_bytes() : gs.kmers_bytes();
                    const uint64_t in_g = b - off[g] * 8, len = std::min<uint64_t>(off[g + 1] * 8 - b, b1 - b);
                    memcpy((uint8_t*)chunk + (b - at), src + in_g, len);
                    b += len;
                }
            };
            for (unsigned w = 1; w < T; w++) th.emplace_back(piece, w);
            piece(0);
            for (auto& t : th) t.join();
            hip_check(sylph_upload_commit(up, n), "sylph_upload_commit");
            at += n;
            while (g0 + 1 < off.size() && off[g0 + 1] * 8 <= at) g0++;
        }
        const void* dev = nullptr;
        hip_check(sylph_upload_finish(up, &dev), "sylph_upload_finish");
        return (const uint64_t*)dev;
    };
    struct UploadGuard { sylph_upload* u = nullptr; ~UploadGuard() { sylph_upload_destroy(u); } };
    std::vector<uint64_t> goff;
    sylph_db* db = nullptr;
    {
        UploadGuard ug, og;
        const uint64_t* d_k = upload_gathered(false, goff, &ug.u);
        // the offsets travel the same way (a few hundred KB)
        sylph_upload* uo = nullptr;
        hip_check(sylph_upload_begin(e.context(), goff.size() * 8, 1u << 20, &uo), "sylph_upload_begin");
        og.u = uo;
        for (size_t i = 0; i < goff.size();) {
            void* chunk = nullptr;
            uint64_t cap = 0;
            hip_check(sylph_upload_chunk(uo, &chunk, &cap), "sylph_upload_chunk");
            const size_t m = std::min<size_t>(goff.size() - i, cap / 8);
            memcpy(chunk, goff.data() + i, m * 8);
            hip_check(sylph_upload_commit(uo, m * 8), "sylph_upload_commit");
            i += m;
        }
        const void* d_off = nullptr;
        hip_check(sylph_upload_finish(uo, &d_off), "sylph_upload_finish");
        hip_check(sylph_db_upload(e.context(), d_k, (const uint64_t*)d_off, genome_sketches.size(), SYLPH_MEM_DEVICE, &db), "sylph_db_upload");
    }
    // (destroyed even on the fast way out: a 29 GB index left to the driver's own clean-up at process exit is released BEHIND the
    //  process — the next command's database load then took 2.7 s instead of 0.7: profiles/r05_db_load_with_forked_profile.txt)
    struct DbGuard { sylph_db* d; ~DbGuard() { sylph_db_destroy(d); } } guard{db};
    if (args.pseudotax) {   // the winner table also ranges over pseudotax_tracked_nonused_kmers (contain.rs:421-428)
        UploadGuard ug, og;
        std::vector<uint64_t> toff;
        const uint64_t* d_t = upload_gathered(true, toff, &ug.u);
        sylph_upload* uo = nullptr;
        hip_check(sylph_upload_begin(e.context(), toff.size() * 8, 1u << 20, &uo), "sylph_upload_begin");
        og.u = uo;
        for (size_t i = 0; i < toff.size();) {
            void* chunk = nullptr;
            uint64_t cap = 0;
            hip_check(sylph_upload_chunk(uo, &chunk, &cap), "sylph_upload_chunk");
            const size_t m = std::min<size_t>(toff.size() - i, cap / 8);
            memcpy(chunk, toff.data() + i, m * 8);
            hip_check(sylph_upload_commit(uo, m * 8), "sylph_upload_commit");
            i += m;
        }
        const void* d_off = nullptr;
        hip_check(sylph_upload_finish(uo, &d_off), "sylph_upload_finish");
        hip_check(sylph_db_attach_tracked(db, d_t, (const uint64_t*)d_off, SYLPH_MEM_DEVICE), "sylph_db_attach_tracked");
    }
    if (getenv("SYLPH_HIP_FEED_TRACE") || getenv("SYLPH_HIP_DEBUG")) {
        char b[200];
        snprintf(b, sizeof(b), "timing: database of %zu genomes (%llu k-mers) uploaded and indexed in %.3f s", genome_sketches.size(),
                 (unsigned long long)goff.back(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_db0).count());
        info(b);
    }

    print_header(args.pseudotax, out, args.estimate_unknown);
    const uint64_t genome_c = genome_sketches[0].c, genome_k = genome_sketches[0].k;
    // ---- one sample's results -> statistics -> (profile: reassignment pass) -> TSV rows.  `cc / coff / covs` are the first-pass
    // views (coverage values cov_width bytes each); the sample table is given where it lies (tk / tc / tn in tmem) for the
    // reassignment probe; S carries the metadata, and the counts on the host when -u needs them.
    auto report = [&](const SequencesSketch& S, const std::string& first_file, const uint32_t* cc, const uint64_t* coff, const void* covs,
                      uint32_t cov_width, const uint64_t* tk, const uint32_t* tc, uint64_t tn, int tmem, sylph_db* rdb) {
        if (genome_k != S.k) throw Error{1, "k parameter for reads != k parameter for genome"};   // contain.rs:608-615
        const std::string seq_name = S.sample_name ? *S.sample_name : S.file_name;   // :775-781
        auto cov_vector = [&](const void* base, uint32_t width, uint64_t lo, uint64_t hi) {
            std::vector<uint32_t> cv(hi - lo);
            if (width == 4) memcpy(cv.data(), (const uint32_t*)base + lo, (hi - lo) * 4);
            else if (width == 2) for (uint64_t i = lo; i < hi; i++) cv[i - lo] = ((const uint16_t*)base)[i];
            else for (uint64_t i = lo; i < hi; i++) cv[i - lo] = ((const uint8_t*)base)[i];
            return cv;
        };
        std::vector<AniResult> stats;
        {   // the statistics of different genomes are independent (contain.rs:284 runs them on the rayon pool): -t threads,
            // results gathered in genome order so that the output does not depend on the interleaving
            std::vector<size_t> with_hits;
            for (size_t g = 0; g < genome_sketches.size(); g++) {
                if (genome_sketches[g].c < S.c) throw Error{1, "c parameter for reads > c parameter for genome"};   // :616-623
                if (cc[g] != 0) with_hits.push_back(g);                  // :654
            }
            std::vector<std::optional<AniResult>> res(with_hits.size());
            parallel_for(with_hits.size(), args.threads, [&](size_t i) {
                const size_t g = with_hits[i];
                res[i] = stats_from_covs(args, cov_vector(covs, cov_width, coff[g], coff[g + 1]), genome_sketches[g].n_kmers(), S.k, std::nullopt);
                if (res[i]) res[i]->genome_index = g;
            });
            for (auto& r : res) if (r) stats.push_back(*r);
        }
        std::optional<double> kmer_id_opt;                               // contain.rs:274-281
        if (args.seq_id) kmer_id_opt = std::pow(*args.seq_id / 100., (double)S.k);
        else kmer_id_opt = get_kmer_identity(S, args.estimate_unknown);
        estimate_true_cov(stats, kmer_id_opt, args.estimate_unknown, S.mean_read_length, S.k);   // :295
        if (args.pseudotax) {
            info(first_file + " taxonomic profiling; reassigning k-mers for " + std::to_string(stats.size()) + " genomes...");
            // winner_table (contain.rs:410-430) + second get_stats pass with the winner map (:300-307, :637-646) on the
            // device: one more probe of the resident postings (genome_kmers + tracked k-mers), passing genomes only.
            std::vector<uint32_t> pg(stats.size());
            std::vector<double> pa(stats.size());
            for (size_t i = 0; i < stats.size(); i++) { pg[i] = (uint32_t)stats[i].genome_index; pa[i] = stats[i].final_est_ani; }
            const uint32_t *cc2 = nullptr, *covs2 = nullptr, *lost2 = nullptr;
            const uint64_t* coff2 = nullptr;
            uint64_t ncov2 = 0;
            hip_check(sylph_db_reassign_view(rdb, tk, tc, tn, tmem, pg.data(), pa.data(), (uint32_t)pg.size(), &cc2, &coff2, &covs2, &ncov2, &lost2),
                      "sylph_db_reassign_view");
            std::vector<AniResult> stats2;
            std::vector<std::optional<AniResult>> res2(stats.size());
            parallel_for(stats.size(), args.threads, [&](size_t i) {
                const size_t g = stats[i].genome_index;
                std::vector<uint32_t> cv(covs2 + coff2[g], covs2 + coff2[g + 1]);
                res2[i] = stats_from_covs(args, std::move(cv), genome_sketches[g].n_kmers(), S.k, (size_t)lost2[g]);
                if (res2[i]) res2[i]->genome_index = g;
            });
            for (size_t i = 0; i < stats.size(); i++) {
                const auto& r = res2[i];
                if (!r) continue;
                // derep_if_reassign_threshold, contain.rs:353-375
                const double thr = std::pow(args.redundant_ani / 100., (double)S.k) * (double)r->n_kmers;
                if ((double)(stats[i].contain_count - r->contain_count) < thr) stats2.push_back(*r);
            }
            stats.swap(stats2);
            estimate_true_cov(stats, kmer_id_opt, args.estimate_unknown, S.mean_read_length, S.k);   // :310
            info(first_file + " has " + std::to_string(stats.size()) + " genomes passing profiling threshold. ");
            double bases_explained = 1.;                                 // :313-317
            if (args.estimate_unknown) {
                bases_explained = estimate_covered_bases(stats, genome_sketches, S, S.mean_read_length, S.k);
                char buf[64];
                snprintf(buf, sizeof buf, "%.2f", bases_explained * 100.);
                info(first_file + " has " + buf + "% of reads detected in database by profile");
            }
            double total_cov = 0, total_seq_cov = 0;                     // contain.rs:319-326
            for (const auto& r : stats) { total_cov += r.final_est_cov; total_seq_cov += r.final_est_cov * (double)genome_sketches[r.genome_index].gn_size; }
            for (auto& r : stats) r.rel_abund = r.final_est_cov / total_cov * 100.;
            for (auto& r : stats) r.seq_abund = r.final_est_cov * (double)genome_sketches[r.genome_index].gn_size / total_seq_cov * 100. * bases_explained;
            std::stable_sort(stats.begin(), stats.end(), [](const AniResult& x, const AniResult& y) { return *y.rel_abund < *x.rel_abund; });   // :330
        } else {
            std::stable_sort(stats.begin(), stats.end(), [](const AniResult& x, const AniResult& y) { return y.final_est_ani < x.final_est_ani; });   // :333
        }
        for (const auto& r : stats) print_ani_result(r, seq_name, genome_sketches[r.genome_index], args.pseudotax, out, args.debug_f64);
    };
    auto finished = [&](const std::vector<std::string>& files) {
        info(std::string(files.size() > 1 ? "Finished paired sample " : "Finished sample ") + files[0] + ".");
    };

    // ---- raw read samples (contain.rs:239-291 sketches and profiles them chunk by chunk on the rayon pool).  Here: up to `-t`
    // sample threads, each with a GPU context of its own, read + index + pack + push their files into sessions, in input order;
    // the sessions go through a sylph_pipeline (finish on the device -> probe, tables never leave HBM), and this thread takes the
    // results in input order and does the statistics and the printing: the feed of sample j + 1 .. overlaps with the profile of
    // sample j and the statistics of sample j - 1.
    const size_t n_raw = read_files.size();
    if (n_raw) {
        struct Prepared { std::optional<SequencesSketch> meta; sylph_sketch* session = nullptr; std::exception_ptr error; };
        std::vector<std::promise<Prepared>> promises(n_raw);
        std::vector<std::future<Prepared>> futures;
        for (auto& p : promises) futures.push_back(p.get_future());
        // --gpus N|all: the database replicated on N GPUs (index copied device to device), sample threads dealt to them in turn,
        // one router pipeline over the replicas (sylph_pipeline_create_multi): the reference's sample loop spans the machine through
        // its rayon pool (contain.rs:252-295), this is the same for the GPUs of a node.  One GPU: everything as before.
        // (SYLPH_HIP_SHARE_GPUS=1: more replicas than devices — they wrap around; how the one-GPU test box runs `--gpus 2`)
        int n_gpus = args.gpus < 0 ? sylph_device_count() : getenv("SYLPH_HIP_SHARE_GPUS") ? args.gpus : std::min(args.gpus, std::max(1, sylph_device_count()));
        n_gpus = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, n_gpus), n_raw));
        std::vector<std::unique_ptr<Engine>> replica_engines;            // (declared before the replicas: destroyed after them)
        struct Replicas { std::vector<sylph_db*> v; ~Replicas() { for (size_t i = 1; i < v.size(); i++) sylph_db_destroy(v[i]); } } replicas;
        std::vector<sylph_db*>& dbs = replicas.v;
        dbs.push_back(db);
        std::vector<int> replica_device{e.device};
        {
            int dev0 = e.device;
            if (dev0 < 0) dev0 = 0;                                      // (Engine(-1) = the current device = 0 in a fresh process)
            replica_device[0] = e.device;
            const auto t_r0 = std::chrono::steady_clock::now();
            for (int g = 1; g < n_gpus; g++) {
                const int dev = (dev0 + g) % std::max(1, sylph_device_count());
                replica_engines.emplace_back(new Engine(dev));
                sylph_db* r = nullptr;
                hip_check(sylph_db_replicate(db, replica_engines.back()->context(), &r), "sylph_db_replicate");
                dbs.push_back(r);
                replica_device.push_back(dev);
            }
            if (n_gpus > 1) {
                char b[160];
                snprintf(b, sizeof(b), "database replicated on %d GPUs (device to device) in %.3f s", n_gpus,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_r0).count());
                info(b);
            }
        }
        std::atomic<size_t> worker_no{0};
        const size_t n_workers = std::max<size_t>(1, std::min<size_t>(std::max<size_t>(std::min<size_t>(args.threads, MAX_SAMPLE_THREADS), (size_t)n_gpus), n_raw));
        set_parse_share((unsigned)n_workers);
        const size_t ahead_limit = n_workers + 2;                        // sessions that may wait, sketched, for the profile stage
        std::vector<IndexAhead::Files> job_files;
        for (const auto& r : read_files) job_files.push_back({r[0], r.size() > 1 ? std::optional<std::string>(r[1]) : std::nullopt});
        IndexAhead ahead(job_files);
        set_feed_budget(4 * n_workers);
        std::atomic<size_t> next_job{0}, released{0}, indexes_obtained{0};
        set_no_more_inflates(false);
        std::atomic<bool> cancel{false};
        std::mutex gate_mu;
        std::condition_variable gate_cv;
        // contain.rs:591: raw pairs are sketched with the default filter (DEFAULT_FPR) — unless the exact set is asked for (not in the reference)
        const double raw_pair_fpr = exact_dedup_accepted(args.exact_dedup) ? 0. : DEFAULT_FPR;
        auto prepare = [&](Engine& eng, size_t j) {
            Prepared pr;
            try {
                const auto& files = read_files[j];
                if (genome_c < args.c) {
                    fprintf(stderr, "ERROR [sylph_hip] %s error: value of -c for contain = %llu -- greater than the smallest value of -c for a genome sketch = %llu. Continuing without sketching.\n", files[0].c_str(), (unsigned long long)args.c, (unsigned long long)genome_c);
                } else if (genome_k != args.k) {
                    fprintf(stderr, "ERROR [sylph_hip] %s -k %llu is not equal to -k %llu found in sketches. Continuing without sketching.\n", files[0].c_str(), (unsigned long long)args.k, (unsigned long long)genome_k);
                } else {
                    const bool gz = device_feed_enabled() && is_gzip_file(files[0]);   // (see sketch(): the device route)
                    const bool gz_dev = gz && device_inflate_enabled();
                    const bool dev = device_feed_enabled() && (gz_dev || (eng.ready() && !gz));
                    std::optional<IndexedInput> pre;
                    if (gz && !gz_dev) pre = index_inputs(files[0], files.size() > 1 ? &files[1] : nullptr, false);
                    else if (!dev) pre = ahead.get(j);
                    if (++indexes_obtained == n_raw) set_no_more_inflates(true);
                    struct Later { std::optional<IndexedInput>& p; ~Later() { if (p) background([x = std::make_shared<std::optional<IndexedInput>>(std::move(p))]() mutable { x.reset(); }); } } later{pre};
                    if (!device_feed_enabled()) ahead.start(j + n_workers);
                    if (files.size() == 1) pr.meta = sketch_sequences_needle_impl(eng, files[0], args.c, args.k, std::nullopt, false, dev ? nullptr : &pre, &pr.session, dev);
                    else pr.meta = sketch_pair_sequences_impl(eng, files[0], files[1], args.c, args.k, std::nullopt, false, raw_pair_fpr, dev ? nullptr : &pre, &pr.session, dev);
                }
            } catch (...) { pr.error = std::current_exception(); }
            promises[j].set_value(std::move(pr));
        };
        auto worker = [&] {
            // every sample thread brings its own context: the database's context (the caller's engine) stays free for the
            // profile stage and the reassignment probes
            std::unique_ptr<Engine> own;
            Engine* eng = nullptr;
            try { own.reset(new Engine(replica_device[worker_no++ % replica_device.size()])); eng = own.get(); }
            catch (...) { eng = nullptr; }
            for (;;) {
                const size_t j = next_job++;
                if (j >= n_raw) return;
                {   // do not run further ahead of the consumer than ahead_limit samples (their sessions hold HBM)
                    std::unique_lock<std::mutex> lk(gate_mu);
                    gate_cv.wait(lk, [&] { return cancel.load() || j < released.load() + ahead_limit; });
                }
                if (cancel) { promises[j].set_value(Prepared{}); continue; }
                if (!eng) { Prepared pr; pr.error = std::make_exception_ptr(Error{1, "could not create a GPU context for a sample thread"}); promises[j].set_value(std::move(pr)); continue; }
                prepare(*eng, j);
            }
        };
        sylph_pipeline* pipe = nullptr;
        sylph_pipeline_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.struct_size = sizeof(cfg);
        cfg.n_workers = 2;                       // (they only finish the sessions the sample threads pushed)
        cfg.depth = (uint32_t)ahead_limit + 1;
        cfg.max_batch = 4;
        cfg.c = (uint32_t)args.c; cfg.k = (uint32_t)args.k;
        cfg.reads_mode = SYLPH_READS_PAIRED; cfg.seed_mode = SYLPH_SEED_AVX2_COMPAT;
        cfg.want_table = args.estimate_unknown ? 1 : 0;                 // -u walks the counts on the host
        cfg.min_number_kmers = args.min_number_kmers;
        if (n_gpus > 1) hip_check(sylph_pipeline_create_multi(dbs.data(), (uint32_t)dbs.size(), &cfg, &pipe), "sylph_pipeline_create_multi");
        else hip_check(sylph_pipeline_create(db, &cfg, &pipe), "sylph_pipeline_create");
        struct PipeGuard { sylph_pipeline* p; ~PipeGuard() { sylph_pipeline_destroy(p); } } pipe_guard{pipe};
        std::vector<std::thread> pool;
        size_t submitted = 0;
        struct Unconsumed {   // (declared before PoolJoin: runs after the sample threads are joined) sessions nobody took over
            std::vector<std::future<Prepared>>& futures; size_t& submitted;
            ~Unconsumed() {
                for (size_t j = submitted; j < futures.size(); j++) {
                    if (!futures[j].valid() || futures[j].wait_for(std::chrono::seconds(0)) != std::future_status::ready) continue;
                    try { Prepared pr = futures[j].get(); if (pr.session) sylph_sketch_destroy(pr.session); } catch (...) {}
                }
            }
        } unconsumed{futures, submitted};
        struct PoolJoin {   // on every way out: the sample threads stop taking work and are joined
            std::vector<std::thread>& pool; std::atomic<bool>& cancel; std::mutex& mu; std::condition_variable& cv;
            ~PoolJoin() {
                { std::lock_guard<std::mutex> lk(mu); cancel = true; }
                cv.notify_all();
                for (auto& t : pool) if (t.joinable()) t.join();
            }
        } pool_join{pool, cancel, gate_mu, gate_cv};
        for (size_t w = 0; w < n_workers; w++) pool.emplace_back(worker);
        std::vector<std::optional<SequencesSketch>> metas(n_raw);
        std::vector<char> submitted_ok(n_raw, 0);
        size_t done = 0;
        auto release_one = [&] { { std::lock_guard<std::mutex> lk(gate_mu); released++; } gate_cv.notify_all(); };
        while (done < n_raw) {
            // hand over every prepared sample that is ready, in input order (wait for one only when nothing is outstanding)
            while (submitted < n_raw && sylph_pipeline_outstanding(pipe) < cfg.depth) {
                const bool must_wait = sylph_pipeline_outstanding(pipe) == 0 && submitted == done;
                if (!must_wait && futures[submitted].wait_for(std::chrono::seconds(0)) != std::future_status::ready) break;
                Prepared pr = futures[submitted].get();
                if (pr.error) { if (pr.session) sylph_sketch_destroy(pr.session); submitted++; std::rethrow_exception(pr.error); }
                metas[submitted] = std::move(pr.meta);
                if (metas[submitted] && pr.session) {
                    hip_check(sylph_pipeline_submit_session(pipe, pr.session, submitted), "sylph_pipeline_submit_session");
                    submitted_ok[submitted] = 1;
                } else if (pr.session) sylph_sketch_destroy(pr.session);
                submitted++;
            }
            const size_t j = done;
            if (j >= submitted) continue;        // (the wait above guarantees progress)
            if (submitted_ok[j]) {
                sylph_pipeline_result r;
                hip_check(sylph_pipeline_next(pipe, &r), "sylph_pipeline_next");
                if (r.status != SYLPH_OK) throw Error{1, std::string("sample ") + read_files[j][0] + ": " + (r.error ? r.error : "GPU stage failed")};
                SequencesSketch& S = *metas[j];
                if (args.estimate_unknown && r.counts) S.counts.assign(r.counts, r.counts + r.n_table);
                report(S, read_files[j][0], r.contain_count, r.cov_off, r.covs, r.cov_width, r.dev_kmers, r.dev_counts, r.n_table, SYLPH_MEM_DEVICE,
                       dbs[(size_t)std::max(0, sylph_pipeline_replica_of_last(pipe))]);
            }
            finished(read_files[j]);
            metas[j].reset();
            done++;
            release_one();
        }
    }

    // ---- samples given as sketches (*.sylsp): the table is on the host
    for (const auto& sf : read_sketch_files) {
        const std::vector<std::string> files{sf};
        // get_seq_sketch, contain.rs:544-599
        std::optional<SequencesSketch> seq;
        {
            SequencesSketch s = read_sylsp(files[0]);
            if (s.c > genome_c) { fprintf(stderr, "ERROR [sylph_hip] %s value of -c is %llu; this is greater than the smallest value of -c = %llu for a genome sketch. Exiting.\n", files[0].c_str(), (unsigned long long)s.c, (unsigned long long)genome_c); }
            else seq = std::move(s);
        }
        if (seq) {
            const SequencesSketch& S = *seq;
            // first pass: GPU probe of every genome, then host statistics
            const uint32_t* cc = nullptr; const uint64_t* coff = nullptr; const uint32_t* covs = nullptr; uint64_t ncov = 0;
            hip_check(sylph_db_contain_view(db, S.kmers.data(), S.counts.data(), S.kmers.size(), SYLPH_MEM_HOST,
                                            args.min_number_kmers, &cc, &coff, &covs, &ncov), "sylph_db_contain_view");
            report(S, files[0], cc, coff, covs, 4, S.kmers.data(), S.counts.data(), S.kmers.size(), SYLPH_MEM_HOST, db);
        }
        finished(files);
    }
    fflush(out);
    join_background();
    info("sylph finished.");
    return 0;
}

}  // namespace sylph_host
