// formats.cpp — .sylsp / .syldb (bincode 1.3.3 default layout) and FASTA/FASTQ(+gzip) records.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <fstream>

#include "sylph_host.hpp"

namespace sylph_host {

namespace {

// Streams to `path + ".tmp"` through a 4 MiB buffer, checks every write and the close, and renames into place: a full disk
// or an I/O error can never leave a truncated database behind a success message (a GTDB-scale .syldb is 13 GB).
struct Writer {
    std::string path, tmp;
    FILE* f = nullptr;
    std::vector<char> b;
    explicit Writer(const std::string& p) : path(p), tmp(p + ".tmp") {
        f = fopen(tmp.c_str(), "wb");
        if (!f) throw Error{1, path + " path not valid; exiting."};
        b.reserve(4u << 20);
    }
    ~Writer() { if (f) { fclose(f); (void)remove(tmp.c_str()); } }
    Writer(const Writer&) = delete;
    Writer& operator=(const Writer&) = delete;
    void spill() {
        if (!b.empty() && fwrite(b.data(), 1, b.size(), f) != b.size()) throw Error{1, "write error on " + path + " (disk full?)"};
        b.clear();
    }
    void raw(const void* p, size_t n) {
        if (n >= (1u << 20)) {                       // large arrays go straight out
            spill();
            if (fwrite(p, 1, n, f) != n) throw Error{1, "write error on " + path + " (disk full?)"};
            return;
        }
        const char* c = (const char*)p;
        b.insert(b.end(), c, c + n);
        if (b.size() >= (4u << 20)) spill();
    }
    void u8(uint8_t v) { raw(&v, 1); }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void f64(double v) { raw(&v, 8); }
    void str(const std::string& s) { u64(s.size()); raw(s.data(), s.size()); }
    void finish() {
        spill();
        const bool bad = fflush(f) != 0 || ferror(f);
        const bool bad_close = fclose(f) != 0;
        f = nullptr;
        if (bad || bad_close) { (void)remove(tmp.c_str()); throw Error{1, "write error on " + path + " (disk full?)"}; }
        if (rename(tmp.c_str(), path.c_str()) != 0) { (void)remove(tmp.c_str()); throw Error{1, "could not move " + tmp + " to " + path}; }
    }
};

// The file is mapped, not read into a buffer: a GTDB-scale .syldb is 13 GB and its k-mer vectors are copied out exactly once.
struct Mapped {
    const char* base = nullptr;
    size_t len = 0;
    size_t size() const { return len; }
    const char& operator[](size_t i) const { return base[i]; }
};
struct Reader {
    Mapped b;
    size_t p = 0;
    std::string path;
    explicit Reader(const std::string& pth) : path(pth) {
        const int fd = open(pth.c_str(), O_RDONLY);
        if (fd < 0) throw Error{1, "The sketch `" + pth + "` could not be opened"};
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); throw Error{1, "The sketch `" + pth + "` could not be opened"}; }
        b.len = (size_t)st.st_size;
        if (b.len) {
            void* m = mmap(nullptr, b.len, PROT_READ, MAP_PRIVATE, fd, 0);
            close(fd);
            if (m == MAP_FAILED) throw Error{1, "The sketch `" + pth + "` could not be mapped"};
            (void)madvise(m, b.len, MADV_SEQUENTIAL);
            b.base = (const char*)m;
        } else {
            close(fd);
        }
    }
    ~Reader() { if (b.base) munmap((void*)b.base, b.len); }
    Reader(const Reader&) = delete;
    Reader& operator=(const Reader&) = delete;
    void need(size_t n) {
        if (n > b.size() || p > b.size() - n) throw Error{1, "The sketch `" + path + "` is not a valid sketch. Perhaps it is an older incompatible version"};
    }
    uint8_t u8() { need(1); return (uint8_t)b[p++]; }
    uint32_t u32() { need(4); uint32_t v; memcpy(&v, &b[p], 4); p += 4; return v; }
    uint64_t u64() { need(8); uint64_t v; memcpy(&v, &b[p], 8); p += 8; return v; }
    double f64() { need(8); double v; memcpy(&v, &b[p], 8); p += 8; return v; }
    std::string str() { const uint64_t n = u64(); need(n); std::string s(&b[p], n); p += n; return s; }
    std::vector<uint64_t> vec_u64() {
        const uint64_t n = u64();
        if (n > b.size() / 8) need(b.size() + 1);   // (n * 8 must not wrap)
        need(n * 8);
        std::vector<uint64_t> v(n);
        if (n) memcpy(v.data(), &b[p], n * 8);
        p += n * 8;
        return v;
    }
};

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

}  // namespace

// types.rs:145-155: kmer_counts as a *sequence* of (u64, u32) (types.rs:132-138), c, k, file_name, sample_name, paired,
// mean_read_length.  The reference writes the map in hashbrown iteration order; readers rebuild a map, so any order is
// equivalent — we write ascending k-mer order.
void write_sylsp(const std::string& path, const SequencesSketch& s) {
    Writer w(path);
    w.u64(s.kmers.size());
    {   // the table as one block of 12-byte entries (a 1 Gbp sample: 1.9 M entries; per-field appends cost 26-35 ms of a 250 ms command)
        const size_t n = s.kmers.size();
        std::vector<char> packed(n * 12);
        char* out = packed.data();
        for (size_t i = 0; i < n; i++, out += 12) { memcpy(out, &s.kmers[i], 8); memcpy(out + 8, &s.counts[i], 4); }
        w.raw(packed.data(), packed.size());
    }
    w.u64(s.c);
    w.u64(s.k);
    w.str(s.file_name);
    if (s.sample_name) { w.u8(1); w.str(*s.sample_name); } else w.u8(0);
    w.u8(s.paired ? 1 : 0);
    w.f64(s.mean_read_length);
    w.finish();
}

SequencesSketch read_sylsp(const std::string& path) {
    Reader r(path);
    SequencesSketch s;
    const uint64_t n = r.u64();
    if (n > r.b.size() / 12) r.need(r.b.size() + 1);
    r.need(n * 12);
    std::vector<std::pair<uint64_t, uint32_t>> kv(n);
    for (uint64_t i = 0; i < n; i++) { kv[i].first = r.u64(); kv[i].second = r.u32(); }
    // the reference inserts into a map (types.rs:117-129): a repeated key keeps its last value
    std::stable_sort(kv.begin(), kv.end(), [](auto& a, auto& b) { return a.first < b.first; });
    for (uint64_t i = 0; i < n; i++) {
        if (i + 1 < n && kv[i + 1].first == kv[i].first) continue;
        s.kmers.push_back(kv[i].first);
        s.counts.push_back(kv[i].second);
    }
    s.c = r.u64();
    s.k = r.u64();
    s.file_name = r.str();
    if (r.u8()) s.sample_name = r.str();
    s.paired = r.u8() != 0;
    s.mean_read_length = r.f64();
    return s;
}

// types.rs:163-173 as Vec<GenomeSketch>
void write_syldb(const std::string& path, const std::vector<GenomeSketch>& gs) {
    Writer w(path);
    w.u64(gs.size());
    for (const GenomeSketch& g : gs) {
        w.u64(g.genome_kmers.size());
        w.raw(g.genome_kmers.data(), g.genome_kmers.size() * 8);
        if (g.pseudotax_tracked_nonused_kmers) {
            w.u8(1);
            w.u64(g.pseudotax_tracked_nonused_kmers->size());
            w.raw(g.pseudotax_tracked_nonused_kmers->data(), g.pseudotax_tracked_nonused_kmers->size() * 8);
        } else {
            w.u8(0);
        }
        w.str(g.file_name);
        w.str(g.first_contig_name);
        w.u64(g.c); w.u64(g.k); w.u64(g.gn_size); w.u64(g.min_spacing);
    }
    w.finish();
}

std::vector<GenomeSketch> read_syldb(const std::string& path) {
    Reader r(path);
    const uint64_t n = r.u64();
    std::vector<GenomeSketch> gs;
    gs.reserve(std::min<uint64_t>(n, 1u << 20));
    for (uint64_t i = 0; i < n; i++) {
        GenomeSketch g;
        g.genome_kmers = r.vec_u64();
        if (r.u8()) g.pseudotax_tracked_nonused_kmers = r.vec_u64();
        g.file_name = r.str();
        g.first_contig_name = r.str();
        g.c = r.u64(); g.k = r.u64(); g.gn_size = r.u64(); g.min_spacing = r.u64();
        gs.push_back(std::move(g));
    }
    return gs;
}

std::vector<GenomeSketch> read_syldb_views(const std::string& path) {
    auto rd = std::make_shared<Reader>(path);
    Reader& r = *rd;
    if (r.b.base) (void)madvise((void*)r.b.base, r.b.len, MADV_RANDOM);   // only the record headers are touched here: no read-ahead of the vectors
    const uint64_t n = r.u64();
    std::vector<GenomeSketch> gs;
    gs.reserve(std::min<uint64_t>(n, 1u << 20));
    auto skip_vec = [&](const uint8_t*& at, uint64_t& len) {
        len = r.u64();
        if (len > r.b.size() / 8) r.need(r.b.size() + 1);
        r.need(len * 8);
        at = (const uint8_t*)&r.b[r.p];
        r.p += len * 8;
    };
    for (uint64_t i = 0; i < n; i++) {
        GenomeSketch g;
        g.view = true;
        g.mapping = rd;
        skip_vec(g.view_kmers, g.view_n);
        if (r.u8()) { g.view_has_tracked = true; skip_vec(g.view_tracked, g.view_tn); }
        g.file_name = r.str();
        g.first_contig_name = r.str();
        g.c = r.u64(); g.k = r.u64(); g.gn_size = r.u64(); g.min_spacing = r.u64();
        gs.push_back(std::move(g));
    }
    if (r.b.base) (void)madvise((void*)r.b.base, r.b.len, MADV_NORMAL);   // (the gather that follows walks the vectors front to back)
    return gs;
}

bool is_fastq(const std::string& f) {
    for (const char* s : {".fq", ".fnq", ".fastq", ".fq.gz", ".fnq.gz", ".fastq.gz"}) if (ends_with(f, s)) return true;
    return false;
}
bool is_fasta(const std::string& f) {
    for (const char* s : {".fa", ".fna", ".fasta", ".fa.gz", ".fna.gz", ".fasta.gz"}) if (ends_with(f, s)) return true;
    return false;
}

// ---- FASTX ---------------------------------------------------------------------------------------------------------
FastxReader::FastxReader(const std::string& path) {
    gz_ = gzopen(path.c_str(), "rb");   // transparently reads plain files too
    if (!gz_) throw Error{1, path + " is not a valid fasta/fastq file; skipping."};
    gzbuffer((gzFile)gz_, 1 << 20);
    buf_.resize(1 << 20);
    buf_.clear();
    // parse_fastx_file (needletail 0.5.1; call sites sketch.rs:488,557,780-781,906) fails up front on an empty file or on a
    // first byte that is neither '>' nor '@': peek, so that callers take their "not a valid fasta/fastq file; skipping" branch
    // instead of writing an empty sketch
    buf_.resize(1 << 20);
    const int n = gzread((gzFile)gz_, &buf_[0], 1 << 20);
    if (n < 0) { gzclose((gzFile)gz_); gz_ = nullptr; throw Error{1, path + " is not a valid fasta/fastq file; skipping."}; }
    buf_.resize((size_t)n);
    if (n == 0) eof_ = true;
    // (the very first byte decides, as in needletail's parse_fastx_reader: a file that begins with a blank line is rejected too)
    if (buf_.empty() || (buf_[0] != '>' && buf_[0] != '@')) {
        gzclose((gzFile)gz_);
        gz_ = nullptr;
        throw Error{1, path + " is not a valid fasta/fastq file; skipping."};
    }
}
FastxReader::~FastxReader() { if (gz_) gzclose((gzFile)gz_); }

bool FastxReader::getline(std::string& line) {
    line.clear();
    if (has_pending_) { line.swap(pending_); has_pending_ = false; return true; }
    for (;;) {
        if (pos_ < buf_.size()) {
            const char* b = buf_.data() + pos_;
            const char* nl = (const char*)memchr(b, '\n', buf_.size() - pos_);
            if (nl) {
                line.append(b, nl - b);
                pos_ += (nl - b) + 1;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(b, buf_.size() - pos_);
            pos_ = buf_.size();
        }
        if (eof_) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            return !line.empty();
        }
        buf_.resize(1 << 20);
        const int n = gzread((gzFile)gz_, &buf_[0], 1 << 20);
        if (n < 0) throw Error{1, "read error"};
        buf_.resize((size_t)n);
        pos_ = 0;
        if (n == 0) eof_ = true;
    }
}

// Fast path for 4-line FASTQ records that lie completely inside the current buffer (all but one record per MiB): four memchr
// calls and two assigns instead of four line copies.  Returns false (nothing consumed) when the slow path has to take over.
bool FastxReader::next_fastq_in_buffer(FastxRecord& rec) {
    if (!started_ || !fastq_ || has_pending_) return false;
    const char* b = buf_.data();
    const size_t n = buf_.size();
    size_t p = pos_;
    const char* e[4];
    size_t start[4];
    for (int l = 0; l < 4; l++) {
        if (p >= n) return false;
        start[l] = p;
        const char* nl = (const char*)memchr(b + p, '\n', n - p);
        if (!nl) return false;
        e[l] = nl;
        p = (size_t)(nl - b) + 1;
    }
    if (b[start[0]] != '@' || e[0] == b + start[0] || b[start[2]] != '+') return false;   // blank lines, malformed: slow path
    auto trim = [&](int l) { const char* z = e[l]; if (z > b + start[l] && z[-1] == '\r') z--; return z; };
    const char* h_end = trim(0);
    const char* s_end = trim(1);
    if (trim(3) - (b + start[3]) != s_end - (b + start[1])) return false;   // quality length != sequence length: slow path reports it
    rec.id.assign(b + start[0] + 1, h_end);
    rec.seq.assign(b + start[1], s_end);
    pos_ = p;
    return true;
}

bool FastxReader::next(FastxRecord& rec) {
    if (next_fastq_in_buffer(rec)) return true;
    std::string line;
    // skip blank lines between records
    do {
        if (!getline(line)) return false;
    } while (line.empty());
    if (!started_) {
        started_ = true;
        if (line[0] == '@') fastq_ = true;
        else if (line[0] == '>') fastq_ = false;
        else throw Error{1, "not a fasta/fastq file"};
    }
    rec.seq.clear();
    if (fastq_) {
        if (line[0] != '@') throw Error{1, "malformed fastq record"};
        rec.id.assign(line, 1, std::string::npos);
        if (!getline(rec.seq)) throw Error{1, "truncated fastq record"};
        std::string plus, qual;
        if (!getline(plus) || plus.empty() || plus[0] != '+') throw Error{1, "malformed fastq record"};
        if (!getline(qual) && !rec.seq.empty()) throw Error{1, "truncated fastq record"};
        if (qual.size() != rec.seq.size()) throw Error{1, "fastq record with quality and sequence of different lengths"};   // needletail rejects it
        return true;
    }
    if (line[0] != '>') throw Error{1, "malformed fasta record"};
    rec.id.assign(line, 1, std::string::npos);
    while (getline(line)) {
        if (!line.empty() && line[0] == '>') { pending_.swap(line); has_pending_ = true; break; }
        rec.seq += line;
    }
    return true;
}

}  // namespace sylph_host
