// feed.cpp — host feed.  Every input file is parsed (and, for .gz, inflated) by its own reader thread
// into chunks of records; the driver thread walks the chunks in file order, copies the sequences into ONE page-locked batch
// (interleaving the mates of a pair) and pushes it with SYLPH_MEM_HOST_PINNED.  Parsing therefore overlaps with the H2D copy
// and the GPU work of the previous batch, the two mate files are read concurrently, and the library needs no staging memcpy.
// Record semantics are those of FastxReader (needletail 0.5.1: seq() without newlines, errors per record).
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

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
PinnedBatch::~PinnedBatch() {
    sylph_pinned_free(bases_);
    sylph_pinned_free(off_);
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

}  // namespace sylph_host
