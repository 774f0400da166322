// feed.cpp — host feed.  Every input file is parsed (and, for .gz, inflated) by its own reader thread
// into chunks of records; the driver thread walks the chunks in file order, copies the sequences into ONE page-locked batch
// (interleaving the mates of a pair) and pushes it with SYLPH_MEM_HOST_PINNED.  Parsing therefore overlaps with the H2D copy
// and the GPU work of the previous batch, the two mate files are read concurrently, and the library needs no staging memcpy.
// Record semantics are those of FastxReader (needletail 0.5.1: seq() without newlines, errors per record).
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include <zlib.h>

#include "sylph_host.hpp"

namespace sylph_host {

struct ChunkStream::Impl {
    std::unique_ptr<FastxReader> reader;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_put, cv_get;
    std::deque<RecordChunk> q;
    bool done = false, stop = false;
    RecordChunk cur;
    size_t rec_i = 0, base_i = 0;
    bool have_cur = false;
    static constexpr size_t CHUNK_BASES = 8u << 20, QUEUE_DEPTH = 4;

    void run() {
        FastxRecord rec;
        RecordChunk c;
        auto put = [&](RecordChunk&& x) {
            std::unique_lock<std::mutex> lk(mu);
            cv_put.wait(lk, [&] { return q.size() < QUEUE_DEPTH || stop; });
            if (stop) return false;
            q.push_back(std::move(x));
            cv_get.notify_one();
            return true;
        };
        for (;;) {
            bool ok = false, err = false;
            try { ok = reader->next(rec); } catch (const Error&) { err = true; }
            if (err) {
                c.len.push_back(RecordChunk::ERR);
            } else if (!ok) {
                break;
            } else {
                if (c.len.empty()) c.first_id = rec.id;
                c.bases.insert(c.bases.end(), rec.seq.begin(), rec.seq.end());
                c.len.push_back((uint32_t)std::min<size_t>(rec.seq.size(), 0xFFFFFFFEu));
                if (rec.seq.size() > 0xFFFFFFFEu) { c.len.back() = RecordChunk::ERR; c.bases.resize(c.bases.size() - rec.seq.size()); }
            }
            if (c.bases.size() >= CHUNK_BASES || c.len.size() >= (1u << 20)) {
                if (!put(std::move(c))) return;
                c = RecordChunk();
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (stop) return;
            }
        }
        if (!c.len.empty()) put(std::move(c));
        std::lock_guard<std::mutex> lk(mu);
        done = true;
        cv_get.notify_all();
    }
};

ChunkStream::ChunkStream(const std::string& path) : p_(new Impl) {
    p_->reader.reset(new FastxReader(path));   // throws Error exactly where FastxReader does
    p_->th = std::thread([this] { p_->run(); });
}

ChunkStream::~ChunkStream() {
    {
        std::lock_guard<std::mutex> lk(p_->mu);
        p_->stop = true;
        p_->cv_put.notify_all();
    }
    if (p_->th.joinable()) p_->th.join();
}

ChunkStream::Kind ChunkStream::next(const uint8_t*& seq, uint32_t& len) {
    Impl& s = *p_;
    while (!s.have_cur || s.rec_i >= s.cur.len.size()) {
        std::unique_lock<std::mutex> lk(s.mu);
        s.cv_get.wait(lk, [&] { return !s.q.empty() || s.done; });
        if (s.q.empty()) return END;
        s.cur = std::move(s.q.front());
        s.q.pop_front();
        s.cv_put.notify_one();
        s.have_cur = true;
        s.rec_i = 0;
        s.base_i = 0;
    }
    const uint32_t l = s.cur.len[s.rec_i++];
    if (l == RecordChunk::ERR) return ERR;
    seq = s.cur.bases.data() + s.base_i;
    len = l;
    s.base_i += l;
    return REC;
}

// ---- page-locked batch ---------------------------------------------------------------------------------------------
PinnedBatch::PinnedBatch() {}
void PinnedBatch::free_packed(Packed& p) {
    if (p.pageable) { free(p.bytes); free(p.off); }
    else { sylph_pinned_free(p.bytes); sylph_pinned_free(p.off); }
    p = Packed{};
}
PinnedBatch::~PinnedBatch() {
    sylph_pinned_free(bases_);
    sylph_pinned_free(off_);
    for (auto& p : pk_) free_packed(p);
    for (auto& p : early_) free_packed(p);
}

// page-locked (the double-buffered slots: needs the GPU runtime) or pageable (the early batches of a process's first sample)
void PinnedBatch::reserve_packed(Packed& p, size_t bytes, size_t recs) {
    auto take = [&](size_t n) -> void* {
        void* q = nullptr;
        if (p.pageable) { if (posix_memalign(&q, 4096, (n + 4095) & ~(size_t)4095) != 0) throw Error{1, "out of host memory for a read batch"}; }
        else if (sylph_pinned_alloc(n, &q) != SYLPH_OK) throw Error{1, std::string("sylph_pinned_alloc: ") + sylph_last_error()};
        return q;
    };
    auto drop = [&](void* q) { if (p.pageable) free(q); else sylph_pinned_free(q); };
    if (bytes > p.cap_bytes) {
        drop(p.bytes);
        p.bytes = nullptr;
        p.cap_bytes = 0;
        p.bytes = (uint8_t*)take(bytes);
        p.cap_bytes = bytes;
    }
    if (recs > p.cap_recs) {
        drop(p.off);
        p.off = nullptr;
        p.cap_recs = 0;
        p.off = (uint64_t*)take((recs + 1) * 8);
        p.cap_recs = recs;
    }
}
void PinnedBatch::prealloc_packed() {
    for (auto& p : pk_) reserve_packed(p, BATCH_BASES / 4 + 64, BATCH_RECS);
}

void PinnedBatch::reserve(size_t bases_cap, size_t recs_cap) {
    if (bases_cap > cap_bases_) {
        uint8_t* nb = nullptr;
        if (sylph_pinned_alloc(bases_cap, (void**)&nb) != SYLPH_OK) throw Error{1, std::string("sylph_pinned_alloc: ") + sylph_last_error()};
        if (n_bases_) memcpy(nb, bases_, n_bases_);
        sylph_pinned_free(bases_);
        bases_ = nb;
        cap_bases_ = bases_cap;
    }
    if (recs_cap > cap_recs_) {
        uint64_t* no = nullptr;
        if (sylph_pinned_alloc((recs_cap + 1) * 8, (void**)&no) != SYLPH_OK) throw Error{1, std::string("sylph_pinned_alloc: ") + sylph_last_error()};
        no[0] = 0;
        if (n_recs_) memcpy(no, off_, (n_recs_ + 1) * 8);
        sylph_pinned_free(off_);
        off_ = no;
        cap_recs_ = recs_cap;
    }
}

void PinnedBatch::flush(sylph_sketch* sk) {
    if (!n_recs_) return;
    if (sylph_sketch_push_n(sk, bases_, off_, n_recs_, n_bases_, SYLPH_MEM_HOST_PINNED) != SYLPH_OK)
        throw Error{1, std::string("sylph_sketch_push: ") + sylph_last_error()};
    n_recs_ = 0;
    n_bases_ = 0;
}

void PinnedBatch::add(sylph_sketch* sk, const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, bool pair) {
    if (!bases_) reserve(BATCH_BASES, BATCH_RECS);
    const size_t need = (size_t)la + (pair ? lb : 0), nrec = pair ? 2 : 1;
    if (n_recs_ && (n_bases_ + need > cap_bases_ || n_recs_ + nrec > cap_recs_)) flush(sk);
    if (need > cap_bases_) reserve(need, cap_recs_);   // one record longer than a whole batch
    memcpy(bases_ + n_bases_, a, la);
    n_bases_ += la;
    off_[++n_recs_] = n_bases_;
    if (pair) {
        memcpy(bases_ + n_bases_, b, lb);
        n_bases_ += lb;
        off_[++n_recs_] = n_bases_;
    }
}

// ---- block-parallel FASTQ indexing -------------------------------------------------------------------
namespace { std::atomic<unsigned> g_parse_share{1}; }
void set_parse_share(unsigned sample_threads) { g_parse_share = std::max(1u, sample_threads); }
unsigned effective_cpus() {
    static const unsigned n = [] {
        unsigned cpus = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) cpus = std::min<unsigned>(cpus, (unsigned)CPU_COUNT(&set));
        auto quota_of = [](const std::string& dir, double& q) {            // cgroup v2: "cpu.max" = "<quota|max> <period>"; v1: two files
            if (FILE* f = fopen((dir + "/cpu.max").c_str(), "r")) {
                char a[64] = {0};
                long long period = 0;
                const int got = fscanf(f, "%63s %lld", a, &period);
                fclose(f);
                if (got == 2 && strcmp(a, "max") != 0 && period > 0) { q = std::min(q, atof(a) / (double)period); return; }
            }
            long long quota = -1, period = 0;
            if (FILE* f = fopen((dir + "/cpu.cfs_quota_us").c_str(), "r")) { if (fscanf(f, "%lld", &quota) != 1) quota = -1; fclose(f); }
            if (FILE* f = fopen((dir + "/cpu.cfs_period_us").c_str(), "r")) { if (fscanf(f, "%lld", &period) != 1) period = 0; fclose(f); }
            if (quota > 0 && period > 0) q = std::min(q, (double)quota / (double)period);
        };
        double q = 1e9;
        quota_of("/sys/fs/cgroup", q);
        quota_of("/sys/fs/cgroup/cpu", q);
        if (FILE* f = fopen("/proc/self/cgroup", "r")) {                    // a nested group: "0::/path" (v2) or "N:cpu,cpuacct:/path" (v1)
            char line[512];
            while (fgets(line, sizeof(line), f)) {
                std::string l(line);
                while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
                const size_t c2 = l.rfind(':');
                if (c2 == std::string::npos || c2 + 1 >= l.size() || l[c2 + 1] != '/') continue;
                const std::string path = l.substr(c2 + 1);
                if (path == "/") continue;
                quota_of("/sys/fs/cgroup" + path, q);
                quota_of("/sys/fs/cgroup/cpu" + path, q);
            }
            fclose(f);
        }
        if (q < 1e8) cpus = std::min<unsigned>(cpus, (unsigned)std::max(1.0, std::ceil(q)));
        if (const char* e = getenv("SYLPH_HIP_CPUS")) cpus = (unsigned)std::max(1, atoi(e));
        return cpus;
    }();
    return n;
}
unsigned parse_threads() {
    static const unsigned n = [] {
        if (const char* e = getenv("SYLPH_HIP_PARSE_THREADS")) return (unsigned)std::max(1, atoi(e));
        // a quarter of the hardware threads (8..64) when the machine is the process's own; under a CPU quota as many as the quota allows
        // (the feed's phases follow each other: each may use all of it)
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency()), cpus = effective_cpus();
        // (one and a half threads per usable CPU: the phases wait on page-cache faults and on each other — measured on the GPU box's 16 CPUs,
        //  four plain 1 Gbp pairs in one command: 0.92 s with 16 threads, 0.73 with 24, 0.78 with 32, 0.81 with 64; tools/feed_threads_sweep.sh)
        return cpus < hw ? std::min(64u, std::max(2u, cpus + cpus / 2)) : std::min(64u, std::max(std::min(hw, 8u), hw / 4));
    }();
    // `-t` sample threads index / gather / index-ahead at the same time, each with this many workers: they share the budget
    // (never below 2 per sample thread), so a large -t on a many-core box does not start thousands of transient threads
    return std::max(std::min(n, 2u), n / g_parse_share.load());
}

namespace { std::atomic<size_t> g_index_budget{0}; }
void set_index_memory_budget(size_t bytes) { g_index_budget = bytes; }
size_t index_memory_budget() { return g_index_budget; }

namespace {
template <class F>
void run_workers(unsigned n, F&& f) {
    std::vector<std::thread> th;
    for (unsigned w = 1; w < n; w++) th.emplace_back([&f, w] { f(w); });
    f(0u);
    for (auto& t : th) t.join();
}
inline size_t next_line(const uint8_t* d, size_t n, size_t p) {   // start of the line after the one containing p (n if none)
    const void* nl = p < n ? memchr(d + p, '\n', n - p) : nullptr;
    return nl ? (size_t)((const uint8_t*)nl - d) + 1 : n;
}
}  // namespace

FastqIndex::~FastqIndex() {
    for (auto& t : releasers_) if (t.joinable()) t.join();
    if (!data) return;
    if (anonymous && map_bytes) inflated_release((void*)data, map_bytes);
    else munmap((void*)data, size);
}

namespace {
struct InflatedPool {
    struct Map { void* p; size_t bytes; };
    std::mutex mu;
    std::vector<Map> idle;
    size_t idle_bytes = 0;
    size_t limit() const {
        const size_t b = index_memory_budget();
        return std::min<size_t>(b ? b / 4 : (size_t)8 << 30, (size_t)16 << 30);
    }
};
InflatedPool& inflated_pool() { static InflatedPool p; return p; }
}  // namespace
void* inflated_acquire(size_t bytes, size_t* map_bytes) {
    InflatedPool& pool = inflated_pool();
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        int best = -1;
        for (size_t i = 0; i < pool.idle.size(); i++)
            if (pool.idle[i].bytes >= bytes && pool.idle[i].bytes <= bytes + bytes / 2 + ((size_t)64 << 20) &&
                (best < 0 || pool.idle[i].bytes < pool.idle[(size_t)best].bytes)) best = (int)i;
        if (best >= 0) {
            const InflatedPool::Map m = pool.idle[(size_t)best];
            pool.idle.erase(pool.idle.begin() + best);
            pool.idle_bytes -= m.bytes;
            *map_bytes = m.bytes;
            return m.p;
        }
    }
    const size_t want = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* buf = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (buf == MAP_FAILED) return nullptr;
    (void)madvise(buf, want, MADV_HUGEPAGE);
    *map_bytes = want;
    return buf;
}
void inflated_release(void* p, size_t map_bytes) {
    InflatedPool& pool = inflated_pool();
    std::vector<InflatedPool::Map> drop;
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        pool.idle.push_back({p, map_bytes});
        pool.idle_bytes += map_bytes;
        while (pool.idle_bytes > pool.limit() && !pool.idle.empty()) {     // the oldest go first
            drop.push_back(pool.idle.front());
            pool.idle_bytes -= pool.idle.front().bytes;
            pool.idle.erase(pool.idle.begin());
        }
    }
    for (const auto& m : drop) munmap(m.p, m.bytes);
}

static std::atomic<bool> g_no_more_inflates{false};
void set_no_more_inflates(bool v) { g_no_more_inflates = v; }
void FastqIndex::release_behind(size_t byte_offset) const {
    // only the inflated copy of a blocked-gzip file is real memory of this process (a mapped plain file is page cache the kernel
    // reclaims by itself): give the pages the feed has gathered from back, so that a large file never stays resident as a whole
    if (!anonymous || !data) return;
    // (a buffer small enough to be recycled keeps its pages: the next file's inflate writes into them instead of faulting new ones in
    //  — unless no further file of the command will be inflated (round 5): then nobody will reuse them, and a GB of 4 KiB pages left to
    //  the kernel at process exit took 0.2 s of a 0.75 s one-sample command; they go back NOW, on a thread of their own)
    const bool recycle = map_bytes && map_bytes <= inflated_pool().limit() / 2;
    if (recycle && !g_no_more_inflates) return;
    const size_t page = 4096, upto = byte_offset / page * page;
    if (upto <= released_) return;
    void* from = (void*)(data + released_);
    const size_t len = upto - released_;
    released_ = upto;
    if (recycle) releasers_.emplace_back([from, len] { (void)madvise(from, len, MADV_DONTNEED); });   // (joined before the mapping goes anywhere: ~FastqIndex)
    else (void)madvise(from, len, MADV_DONTNEED);
}

namespace {
// ---- BGZF (blocked gzip, what `bgzip` writes): every member is a complete deflate stream of <= 64 KiB whose compressed size
// stands in its header (extra subfield 'B','C') and whose inflated size stands in its trailer, so the members can be found
// without inflating and inflated independently — by all parse threads at once, straight to their final offsets.  An ordinary
// .gz is ONE deflate stream: nothing to split, it stays with the sequential reader (needletail does the same, one thread).
struct BgzfBlock { size_t in, in_len, out; uint32_t out_len, crc; };

// libdeflate (2-3x zlib's inflate rate) when the system has it — bound with dlopen, there is no header to build against —
// else zlib.  Both inflate one raw deflate stream into a buffer of known size.
struct Deflate {
    void* lib = nullptr;
    void* (*alloc)() = nullptr;
    int (*run)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    Deflate() {
        if (getenv("SYLPH_HIP_NO_LIBDEFLATE")) return;
        lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return;
        alloc = (decltype(alloc))dlsym(lib, "libdeflate_alloc_decompressor");
        run = (decltype(run))dlsym(lib, "libdeflate_deflate_decompress");
        release = (decltype(release))dlsym(lib, "libdeflate_free_decompressor");
        crc = (decltype(crc))dlsym(lib, "libdeflate_crc32");
        if (!alloc || !run || !release || !crc) lib = nullptr;
    }
};
const Deflate& deflate_lib() { static const Deflate d; return d; }

bool bgzf_blocks(const uint8_t* d, size_t n, std::vector<BgzfBlock>& blocks, size_t& total) {
    size_t p = 0;
    total = 0;
    while (p < n) {
        if (n - p < 28 || d[p] != 0x1f || d[p + 1] != 0x8b || d[p + 2] != 8 || !(d[p + 3] & 4)) return false;
        const size_t xlen = d[p + 10] | (size_t)d[p + 11] << 8;
        if (p + 12 + xlen > n) return false;
        size_t bsize = 0;
        for (size_t q = p + 12; q + 4 <= p + 12 + xlen;) {             // the extra field's subfields
            const size_t slen = d[q + 2] | (size_t)d[q + 3] << 8;
            if (d[q] == 'B' && d[q + 1] == 'C' && slen == 2 && q + 6 <= p + 12 + xlen) bsize = (d[q + 4] | (size_t)d[q + 5] << 8) + 1;
            q += 4 + slen;
        }
        if (d[p + 3] != 4 || bsize < 12 + xlen + 8 || p + bsize > n) return false;   // (bgzip sets no other flag)
        const uint8_t* t = d + p + bsize - 8;
        BgzfBlock b;
        b.in = p + 12 + xlen;
        b.in_len = bsize - 12 - xlen - 8;
        b.crc = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        b.out_len = t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
        b.out = total;
        if (b.out_len > (1u << 16)) return false;
        total += b.out_len;
        blocks.push_back(b);
        p += bsize;
    }
    return !blocks.empty();
}

bool bgzf_inflate(const uint8_t* d, const std::vector<BgzfBlock>& blocks, uint8_t* out, unsigned threads) {
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, blocks.size() / 64 + 1));
    std::vector<char> good(T, 1);
    const Deflate& L = deflate_lib();
    run_workers(T, [&](unsigned w) {
        const size_t b0 = blocks.size() * w / T, b1 = blocks.size() * (w + 1) / T;
        void* dec = L.lib ? L.alloc() : nullptr;
        z_stream z;
        memset(&z, 0, sizeof(z));
        if (!dec && inflateInit2(&z, -15) != Z_OK) { good[w] = 0; return; }
        for (size_t i = b0; i < b1 && good[w]; i++) {
            const BgzfBlock& b = blocks[i];
            if (dec) {
                size_t got = 0;
                if (L.run(dec, d + b.in, b.in_len, out + b.out, b.out_len, &got) != 0 || got != b.out_len) good[w] = 0;
                else if (L.crc(0, out + b.out, b.out_len) != b.crc) good[w] = 0;
            } else {
                z.next_in = const_cast<Bytef*>(d + b.in);
                z.avail_in = (uInt)b.in_len;
                z.next_out = out + b.out;
                z.avail_out = b.out_len;
                const int r = inflate(&z, Z_FINISH);
                if (r != Z_STREAM_END || z.avail_out != 0) good[w] = 0;
                else if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), out + b.out, b.out_len) != b.crc) good[w] = 0;
                inflateReset(&z);
            }
        }
        if (dec) L.release(dec); else inflateEnd(&z);
    });
    for (unsigned w = 0; w < T; w++) if (!good[w]) return false;
    return true;
}
}  // namespace

FastqIndex::FastqIndex(const std::string& path, unsigned threads, bool build) : path_(path) {
    struct stat st;
    // a pipe (mkfifo, process substitution) is opened ONCE, by the sequential reader: opening and closing it here would end its writer
    if (stat(path.c_str(), &st) != 0 || st.st_size < 4 || !S_ISREG(st.st_mode)) return;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return;
    if (fstat(fd, &st) != 0 || st.st_size < 4 || !S_ISREG(st.st_mode)) { close(fd); return; }
    size = (size_t)st.st_size;
    void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); data = nullptr; return; }
    data = (const uint8_t*)m;
    // An uncompressed file is COPIED into anonymous memory (2 MiB pages where the system gives them) by all parse threads with
    // pread: walking a file mapping costs one minor fault per 64 KiB, all of them under the process's one address-space lock —
    // with 2 x 64 threads indexing the two mate files that lock, not the memory, set the pace (70 ms per GB; the copy takes
    // ~15).  Within the memory budget only (beyond it the file stays a mapping: page cache, reclaimable), and the copy's pages
    // are given back behind the gather cursor like those of an inflated file.
    // (opt-in, SYLPH_HIP_FEED_COPY=1: measured on the GPU box, the copy halves the index time but its page zeroing and the extra
    //  pass over the data slow the gather of the sample being pushed meanwhile by more than that)
    if (!(data[0] == 0x1f && data[1] == 0x8b) && getenv("SYLPH_HIP_FEED_COPY")) {
        const size_t budget = index_memory_budget();
        if (!budget || size <= budget) {
            void* buf = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (buf != MAP_FAILED) {
                (void)madvise(buf, size, MADV_HUGEPAGE);
                const unsigned Tc = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, size / (8u << 20) + 1));
                std::vector<char> good(Tc, 1);
                run_workers(Tc, [&](unsigned w) {
                    size_t p = size / Tc * w;
                    const size_t end = w + 1 == Tc ? size : size / Tc * (w + 1);
                    while (p < end) {
                        const ssize_t r = pread(fd, (uint8_t*)buf + p, std::min<size_t>(end - p, 8u << 20), (off_t)p);
                        if (r <= 0) { good[w] = 0; return; }
                        p += (size_t)r;
                    }
                });
                bool all = true;
                for (char g : good) all = all && g;
                if (all) {
                    munmap(m, size);
                    data = (const uint8_t*)buf;
                    anonymous = true;
                } else
                    munmap(buf, size);
            }
        }
    }
    close(fd);
    if (data[0] == 0x1f && data[1] == 0x8b) {
        // blocked gzip: inflate all members in parallel into an anonymous mapping that takes the file mapping's place
        std::vector<BgzfBlock> blocks;
        size_t total = 0;
        if (!bgzf_blocks(data, size, blocks, total) || total < 4) {
            // an ordinary gzip file is ONE deflate stream: pgunzip.cpp inflates it on all parse threads where it can prove the
            // result (member CRC); otherwise the sequential reader (needletail does the same, one thread)
            uint8_t* buf = nullptr;
            size_t n_out = 0;
            const unsigned hw = effective_cpus();
            const unsigned tz = getenv("SYLPH_HIP_PARSE_THREADS") ? threads : std::max(threads, std::min(64u, std::max(1u, hw / 2)));
            size_t n_map = 0;
            if (!parallel_gunzip(data, size, tz, &buf, &n_out, &n_map, index_memory_budget())) return;
            munmap((void*)data, size);
            data = buf;
            size = n_out;
            map_bytes = n_map;
            anonymous = true;
            blocks.clear();
        }
        if (!blocks.empty()) {
        // The inflated copy is anonymous memory: with overcommit the mapping always succeeds and a file (times the files indexed
        // at once) beyond what the machine has ends in the OOM killer instead of in the sequential reader, which runs in
        // constant memory.  The drivers set the budget from MemAvailable and their concurrency.
        if (const size_t budget = index_memory_budget(); budget && total > budget) return;
        size_t n_map = 0;
        void* buf = inflated_acquire(total, &n_map);                               // (a recycled buffer where there is one: no first touch)
        if (!buf) return;
        // (inflating is pure compute per member: it takes more threads than the memory-bound index does)
        const unsigned hw = effective_cpus();
        const bool inflated = bgzf_inflate(data, blocks, (uint8_t*)buf, getenv("SYLPH_HIP_PARSE_THREADS") ? threads : std::max(threads, std::min(64u, std::max(1u, hw / 2))));
        munmap((void*)data, size);
        data = (const uint8_t*)buf;
        size = total;
        map_bytes = n_map;
        anonymous = true;
        if (!inflated) return;                                                     // (the sequential reader reports the damage)
        }
    }
    if (data[0] != '@') return;                                   // not FASTQ: sequential reader
    text_ready = true;
    if (build) build_index(threads);
}

void FastqIndex::build_index(unsigned threads) {
    if (!text_ready || ok) return;
    const std::string& path = path_;
    const uint8_t* d = data;
    const size_t n = size;
    static const bool trace = getenv("SYLPH_HIP_FEED_TRACE") != nullptr;
    const auto t_ix0 = std::chrono::steady_clock::now();
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / (1u << 20)));
    // a record starts at a line that begins with '@' and whose line after next begins with '+' (a QUALITY line may begin
    // with '@' too, but then the line after next is a sequence line)
    std::vector<size_t> starts(T + 1, n);
    starts[0] = 0;
    run_workers(T, [&](unsigned w) {
        {   // read-ahead for this worker's share of a file mapping, asked for by the worker itself (one call over the whole
            // file walks a quarter of a million page-cache entries on the caller's thread before anything else happens)
            const size_t a = n / T * w / 4096 * 4096, b = w + 1 == T ? n : n / T * (w + 1);
            if (!anonymous && b > a) (void)madvise((void*)(d + a), b - a, MADV_WILLNEED);
        }
        if (w == 0) return;
        size_t p = next_line(d, n, n / T * w);
        while (p < n) {
            if (d[p] == '@') {
                const size_t l2 = next_line(d, n, next_line(d, n, p));
                if (l2 < n && d[l2] == '+') break;
            }
            p = next_line(d, n, p);
        }
        starts[w] = p;
    });
    for (unsigned w = 0; w < T; w++) if (starts[w] > starts[w + 1]) return;   // (cannot happen for a regular file)
    std::vector<std::vector<uint64_t>> offs(T);
    std::vector<std::vector<uint32_t>> lens(T);
    std::vector<char> good(T, 1);
    std::vector<uint64_t> range_bases(T, 0);
    run_workers(T, [&](unsigned w) {
        uint64_t sum = 0;
        size_t p = starts[w];
        const size_t end = starts[w + 1];
        auto& o = offs[w];
        auto& l = lens[w];
        o.reserve((end - p) / 300 + 16);
        l.reserve((end - p) / 300 + 16);
        while (p < end) {
            if (d[p] != '@') {                                     // only blank space may follow the last record
                for (size_t q = p; q < n; q++) if (d[q] != '\n' && d[q] != '\r') { good[w] = 0; return; }
                if (end != n) { good[w] = 0; return; }
                p = end;
                break;
            }
            const size_t s = next_line(d, n, p);
            if (s >= n) { good[w] = 0; return; }
            const size_t pl = next_line(d, n, s);
            if (pl >= n || d[pl] != '+') { good[w] = 0; return; }
            if (d[pl - 1] != '\n') { good[w] = 0; return; }        // (pl < n: the sequence line was terminated)
            const size_t q = next_line(d, n, pl);                  // quality line [q, nx): may be the unterminated last line
            if (q == n && d[n - 1] != '\n') { good[w] = 0; return; }   // the '+' line is the last, unterminated line: no quality
            const size_t nx = next_line(d, n, q);
            size_t sl = pl - 1 - s, ql = nx > q ? ((d[nx - 1] == '\n') ? nx - 1 - q : nx - q) : 0;
            if (sl && d[s + sl - 1] == '\r') sl--;
            if (ql && d[q + ql - 1] == '\r') ql--;
            if (sl != ql || sl > 0xFFFFFFFEu) { good[w] = 0; return; }
            o.push_back(s);
            l.push_back((uint32_t)sl);
            sum += sl;
            p = nx;
        }
        range_bases[w] = sum;
        if (p != end && !(end == n && p >= n)) good[w] = 0;        // the next range must begin exactly where this one ended
    });
    for (unsigned w = 0; w < T; w++) if (!good[w]) return;
    std::vector<size_t> pre(T + 1, 0);
    std::vector<uint64_t> bases_before(T + 1, 0);
    for (unsigned w = 0; w < T; w++) {
        pre[w + 1] = pre[w] + lens[w].size();
        bases_before[w + 1] = bases_before[w] + range_bases[w];
    }
    seq_off.resize(pre[T]);
    seq_len.resize(pre[T]);
    cum.resize(pre[T] + 1);                                       // cum[i] = bases of records [0, i): what the batches are cut by
    cum[0] = 0;
    run_workers(T, [&](unsigned w) {
        if (!lens[w].empty()) {
            memcpy(seq_off.data() + pre[w], offs[w].data(), offs[w].size() * 8);
            memcpy(seq_len.data() + pre[w], lens[w].data(), lens[w].size() * 4);
            uint64_t run = bases_before[w];
            uint64_t* c = cum.data() + pre[w] + 1;
            for (size_t i = 0; i < lens[w].size(); i++) { run += lens[w][i]; c[i] = run; }
        }
    });
    ok = true;
    if (trace)
        fprintf(stderr, "[sylph_hip feed] index of %-40s %8.3f ms (%zu records, %u threads)\n", path.c_str(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ix0).count() * 1e3, seq_len.size(), T);
}

void PinnedBatch::gather_packed(int slot, const FastqIndex& a, const FastqIndex* b, const std::vector<uint64_t>& cum_a,
                                const std::vector<uint64_t>* cum_b, size_t i0, size_t i1, unsigned threads) {
    gather_into(pk_[slot & 1], a, b, cum_a, cum_b, i0, i1, threads);
}
// The same into PAGEABLE memory, a batch of its own each time: nothing here needs the GPU runtime, so the first sample of a process is
// gathered and packed while its context is still coming up (~0.2 s of runtime initialisation, during which the parse threads idled).
void PinnedBatch::gather_packed_early(const FastqIndex& a, const FastqIndex* b, const std::vector<uint64_t>& cum_a,
                                      const std::vector<uint64_t>* cum_b, size_t i0, size_t i1, unsigned threads) {
    early_.emplace_back();
    early_.back().pageable = true;
    gather_into(early_.back(), a, b, cum_a, cum_b, i0, i1, threads);
}
// the early batches go through the library's own staging buffers (SYLPH_MEM_HOST), in order, and are freed
void PinnedBatch::push_early(sylph_sketch* sk) {
    for (auto& p : early_) {
        if (p.n_recs && sylph_sketch_push_enc(sk, p.bytes, p.off, p.n_recs, p.n_bases, SYLPH_MEM_HOST, SYLPH_ENC_2BIT) != SYLPH_OK)
            throw Error{1, std::string("sylph_sketch_push: ") + sylph_last_error()};
        free_packed(p);
    }
    early_.clear();
}
void PinnedBatch::gather_into(Packed& p, const FastqIndex& a, const FastqIndex* b, const std::vector<uint64_t>& cum_a,
                              const std::vector<uint64_t>* cum_b, size_t i0, size_t i1, unsigned threads) {
    p.n_recs = p.n_bases = 0;
    if (i1 <= i0) return;
    const size_t n_items = i1 - i0, nrec = b ? 2 * n_items : n_items;
    const uint64_t base0 = cum_a[i0] + (b ? (*cum_b)[i0] : 0);
    const uint64_t total = cum_a[i1] + (b ? (*cum_b)[i1] : 0) - base0;
    reserve_packed(p, std::max<size_t>((total + 3) / 4 + 64, p.cap_bytes), std::max<size_t>(nrec, p.cap_recs));
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n_items / 4096 + 1));
    p.off[0] = 0;
    std::vector<Pack2Bit> writers;
    writers.reserve(T);
    for (unsigned w = 0; w < T; w++) {
        const size_t j0 = i0 + n_items * w / T;
        writers.emplace_back(p.bytes, cum_a[j0] + (b ? (*cum_b)[j0] : 0) - base0);
    }
    run_workers(T, [&](unsigned w) {
        const size_t j0 = i0 + n_items * w / T, j1 = i0 + n_items * (w + 1) / T;
        Pack2Bit& wr = writers[w];
        for (size_t i = j0; i < j1; i++) {
            uint64_t o = cum_a[i] + (b ? (*cum_b)[i] : 0) - base0;
            wr.append(a.data + a.seq_off[i], a.seq_len[i], a.size - a.seq_off[i]);
            o += a.seq_len[i];
            const size_t r = (b ? 2 * (i - i0) : (i - i0)) + 1;
            p.off[r] = o;
            if (b) {
                wr.append(b->data + b->seq_off[i], b->seq_len[i], b->size - b->seq_off[i]);
                p.off[r + 1] = o + b->seq_len[i];
            }
        }
        wr.finish();
    });
    merge_pack_edges(p.bytes, writers.data(), writers.size());
    memset(p.bytes + (total + 3) / 4, 0, 64);            // (slack the device side may read)
    p.n_recs = nrec;
    p.n_bases = total;
}

bool device_feed_enabled() {
    static const bool on = [] {
        if (getenv("SYLPH_HIP_SEQUENTIAL_FEED")) return false;
        if (const char* e = getenv("SYLPH_HIP_FEED_DEVICE")) return atoi(e) != 0;
        return true;
    }();
    return on;
}
TextUploader::~TextUploader() { sylph_upload_destroy(up_); }
void TextUploader::prepare(sylph_ctx* ctx) {
    std::lock_guard<std::mutex> lk(mu_);
    if (!up_ && sylph_upload_begin(ctx, 1u << 20, 64ull << 20, &up_) != SYLPH_OK) up_ = nullptr;   // (send tries again and reports)
}
bool TextUploader::send(sylph_ctx* ctx, const std::vector<std::string>& files, unsigned threads, std::vector<Text>& out) {
    struct Fd { int fd = -1; ~Fd() { if (fd >= 0) close(fd); } };
    std::vector<Fd> fds(files.size());
    std::vector<Src> src;
    for (size_t i = 0; i < files.size(); i++) {
        struct stat st;
        if (stat(files[i].c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 4) return false;   // (a pipe is never opened here)
        fds[i].fd = open(files[i].c_str(), O_RDONLY);
        if (fds[i].fd < 0 || fstat(fds[i].fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 4) return false;
        uint8_t head[2];
        if (pread(fds[i].fd, head, 2, 0) != 2 || head[0] != '@') return false;      // (gzip: 0x1f 0x8b; FASTA: '>')
        src.push_back(Src{fds[i].fd, nullptr, (uint64_t)st.st_size});
    }
    return send_sources(ctx, src, threads, out);
}
bool TextUploader::send(sylph_ctx* ctx, const std::vector<Mem>& texts, unsigned threads, std::vector<Text>& out) {
    std::vector<Src> src;
    for (const Mem& m : texts) {
        if (!m.p || m.bytes < 4 || m.p[0] != '@') return false;
        src.push_back(Src{-1, m.p, m.bytes});
    }
    return send_sources(ctx, src, threads, out);
}
bool TextUploader::send_sources(sylph_ctx* ctx, const std::vector<Src>& src, unsigned threads, std::vector<Text>& out) {
    std::lock_guard<std::mutex> lk(mu_);
    std::vector<uint64_t> at(src.size() + 1, 0);
    for (size_t i = 0; i < src.size(); i++) at[i + 1] = (at[i] + src[i].size + 15) & ~15ull;
    const uint64_t total = at[src.size()];
    // the whole text of the sample lies in device memory at once (plus its line index and the gathered bases): beyond this the host feed,
    // which works through a sample in batches, takes it
    static const uint64_t max_text = (uint64_t)((getenv("SYLPH_HIP_FEED_DEVICE_MAX_GB") ? atof(getenv("SYLPH_HIP_FEED_DEVICE_MAX_GB")) : 16.) * (1ull << 30));
    if (total > max_text) return false;
    auto check = [](int rc, const char* what) { if (rc != SYLPH_OK) throw Error{1, std::string(what) + ": " + sylph_last_error()}; };
    // (no room for the text on the device: not an error — the host feed takes the sample)
    if (!up_) { const int rc = sylph_upload_begin(ctx, total, 64ull << 20, &up_); if (rc == SYLPH_ERR_NOMEM) { up_ = nullptr; return false; } check(rc, "sylph_upload_begin"); }
    else { const int rc = sylph_upload_restart(up_, total); if (rc == SYLPH_ERR_NOMEM) return false; check(rc, "sylph_upload_restart"); }
    uint64_t g = 0;                                                                  // bytes of the side-by-side layout sent so far
    while (g < total) {
        void* chunk = nullptr;
        uint64_t cap = 0;
        check(sylph_upload_chunk(up_, &chunk, &cap), "sylph_upload_chunk");
        const uint64_t n = std::min<uint64_t>(cap, total - g);
        const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads, n / (4u << 20) + 1));
        std::atomic<bool> good{true};
        run_workers(T, [&](unsigned w) {
            uint64_t p = g + n * w / T;
            const uint64_t end = g + n * (w + 1) / T;
            while (p < end) {
                size_t i = 0;
                while (at[i + 1] <= p) i++;                                           // the text (or the padding behind it) p lies in
                uint8_t* dst = (uint8_t*)chunk + (p - g);
                if (p >= at[i] + src[i].size) {                                       // padding up to the next 16-byte boundary
                    const uint64_t m = std::min(end, at[i + 1]) - p;
                    memset(dst, '\n', m);
                    p += m;
                    continue;
                }
                const uint64_t m = std::min(end, at[i] + src[i].size) - p;
                if (src[i].mem) { memcpy(dst, src[i].mem + (p - at[i]), m); p += m; continue; }
                const ssize_t r = pread(src[i].fd, dst, std::min<uint64_t>(m, 8u << 20), (off_t)(p - at[i]));
                if (r <= 0) { good = false; return; }
                p += (uint64_t)r;
            }
        });
        if (!good) { check(sylph_upload_commit(up_, 0), "sylph_upload_commit"); throw Error{1, "could not read the sample's text"}; }
        check(sylph_upload_commit(up_, n), "sylph_upload_commit");
        g += n;
    }
    const void* dev = nullptr;
    check(sylph_upload_finish(up_, &dev), "sylph_upload_finish");
    out.clear();
    for (size_t i = 0; i < src.size(); i++) out.push_back(Text{(const uint8_t*)dev + at[i], src[i].size});
    sent_ = true;
    return true;
}

void PinnedBatch::push_packed(sylph_sketch* sk, int slot) {
    Packed& p = pk_[slot & 1];
    if (!p.n_recs) return;
    if (sylph_sketch_push_enc(sk, p.bytes, p.off, p.n_recs, p.n_bases, SYLPH_MEM_HOST_PINNED, SYLPH_ENC_2BIT) != SYLPH_OK)
        throw Error{1, std::string("sylph_sketch_push: ") + sylph_last_error()};
    p.n_recs = p.n_bases = 0;
}

}  // namespace sylph_host
