// capi.cpp — flat C entry points over the host library for ctypes tests (statistics + on-disk formats).
#include <cstring>

#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "sylph_host.hpp"

using namespace sylph_host;

extern "C" {

struct SylphHostStats {
    double naive_ani, final_est_ani, final_est_cov, mean_cov, median_cov, lambda;
    double ani_ci_lo, ani_ci_hi, lambda_ci_lo, lambda_ci_hi;
    int32_t lambda_status;   // 0 Low, 1 High, 2 Lambda
    int32_t passed, has_ci, pad;
    uint64_t contain_count, n_kmers;
};

// statistics half of get_stats (contain.rs:657-813)
int sylph_host_stats(const uint32_t* covs, uint64_t n, uint64_t n_genome_kmers, uint64_t k, double min_count_correct,
                     double minimum_ani_or_neg, int pseudotax, int no_ci, int no_adj, int mean_coverage, SylphHostStats* out) {
    memset(out, 0, sizeof(*out));
    ContainArgs a;
    a.min_count_correct = min_count_correct;
    if (minimum_ani_or_neg >= 0) a.minimum_ani = minimum_ani_or_neg;
    a.pseudotax = pseudotax; a.no_ci = no_ci; a.no_adj = no_adj; a.mean_coverage = mean_coverage;
    auto r = stats_from_covs(a, std::vector<uint32_t>(covs, covs + n), n_genome_kmers, k, std::nullopt);
    if (!r) return 0;
    out->passed = 1;
    out->naive_ani = r->naive_ani; out->final_est_ani = r->final_est_ani; out->final_est_cov = r->final_est_cov;
    out->mean_cov = r->mean_cov; out->median_cov = r->median_cov; out->lambda = r->lambda;
    out->lambda_status = r->lambda_status == AdjustStatus::Low ? 0 : (r->lambda_status == AdjustStatus::High ? 1 : 2);
    if (r->ani_ci_lo) { out->has_ci = 1; out->ani_ci_lo = *r->ani_ci_lo; out->ani_ci_hi = *r->ani_ci_hi; out->lambda_ci_lo = *r->lambda_ci_lo; out->lambda_ci_hi = *r->lambda_ci_hi; }
    out->contain_count = r->contain_count; out->n_kmers = r->n_kmers;
    return 1;
}

// -u: read k-mer identity of a sample table given in the order it is walked (contain.rs:901-951), and the share of the sample's
// bases that genomes of sizes gn_size[i] at coverages cov[i] explain (contain.rs:392-408)
double sylph_host_kmer_identity(const uint32_t* counts, uint64_t n, uint64_t k, double mean_read_length) {
    SequencesSketch S;
    S.counts.assign(counts, counts + n);
    S.k = k;
    S.mean_read_length = mean_read_length;
    return *get_kmer_identity(S, true);
}
double sylph_host_covered_bases(const uint64_t* gn_size, const double* cov, uint64_t n_genomes, const uint32_t* counts, uint64_t n,
                                uint64_t c, uint64_t k, double mean_read_length) {
    SequencesSketch S;
    S.counts.assign(counts, counts + n);
    S.k = k; S.c = c;
    std::vector<GenomeSketch> gs(n_genomes);
    std::vector<AniResult> rs(n_genomes);
    for (uint64_t i = 0; i < n_genomes; i++) { gs[i].gn_size = gn_size[i]; rs[i].final_est_cov = cov[i]; rs[i].genome_index = i; }
    return estimate_covered_bases(rs, gs, S, mean_read_length, k);
}

// `inspect` scalars (tests): returns the length written (without the terminator), 0 if the buffer is too small
static uint64_t copy_out(const std::string& s, char* buf, uint64_t cap) {
    if (s.size() + 1 > cap) return 0;
    memcpy(buf, s.c_str(), s.size() + 1);
    return s.size();
}
uint64_t sylph_host_inspect_f32(float v, char* buf, uint64_t cap) { return copy_out(inspect_f32(v), buf, cap); }
uint64_t sylph_host_inspect_f64(double v, char* buf, uint64_t cap) { return copy_out(inspect_f64(v), buf, cap); }
uint64_t sylph_host_inspect_str(const char* s, char* buf, uint64_t cap) { return copy_out(inspect_str(s), buf, cap); }

double sylph_host_poisson_cdf(double lambda, uint64_t x) { return poisson_cdf(lambda, x); }

// round trip helpers: write a .sylsp from arrays, read it back into caller buffers (sizes via the first call)
int sylph_host_write_sylsp(const char* path, const uint64_t* kmers, const uint32_t* counts, uint64_t n, uint64_t c, uint64_t k,
                           const char* file_name, const char* sample_name_or_null, int paired, double mean_read_length) {
    try {
        SequencesSketch s;
        s.kmers.assign(kmers, kmers + n); s.counts.assign(counts, counts + n);
        s.c = c; s.k = k; s.file_name = file_name; s.paired = paired; s.mean_read_length = mean_read_length;
        if (sample_name_or_null) s.sample_name = sample_name_or_null;
        write_sylsp(path, s);
        return 0;
    } catch (const Error&) { return -1; }
}

int sylph_host_read_sylsp(const char* path, uint64_t* kmers, uint32_t* counts, uint64_t cap, uint64_t* n, uint64_t* c, uint64_t* k,
                          int* paired, double* mean_read_length, char* file_name, char* sample_name, uint64_t name_cap,
                          int* has_sample_name) {
    try {
        SequencesSketch s = read_sylsp(path);
        *n = s.kmers.size(); *c = s.c; *k = s.k; *paired = s.paired; *mean_read_length = s.mean_read_length;
        *has_sample_name = s.sample_name ? 1 : 0;
        snprintf(file_name, name_cap, "%s", s.file_name.c_str());
        snprintf(sample_name, name_cap, "%s", s.sample_name ? s.sample_name->c_str() : "");
        for (uint64_t i = 0; i < s.kmers.size() && i < cap; i++) { kmers[i] = s.kmers[i]; counts[i] = s.counts[i]; }
        return 0;
    } catch (const Error&) { return -1; }
}

// .syldb: number of genomes, then per-genome accessors
void* sylph_host_read_syldb(const char* path) {
    try { return new std::vector<GenomeSketch>(read_syldb(path)); } catch (const Error&) { return nullptr; }
}
uint64_t sylph_host_syldb_size(void* h) { return ((std::vector<GenomeSketch>*)h)->size(); }
void sylph_host_syldb_genome(void* h, uint64_t i, uint64_t* n_kmers, uint64_t* n_tracked, int* has_tracked, uint64_t* c, uint64_t* k,
                             uint64_t* gn_size, uint64_t* min_spacing, char* file_name, char* contig_name, uint64_t name_cap) {
    const GenomeSketch& g = (*(std::vector<GenomeSketch>*)h)[i];
    *n_kmers = g.genome_kmers.size();
    *has_tracked = g.pseudotax_tracked_nonused_kmers ? 1 : 0;
    *n_tracked = g.pseudotax_tracked_nonused_kmers ? g.pseudotax_tracked_nonused_kmers->size() : 0;
    *c = g.c; *k = g.k; *gn_size = g.gn_size; *min_spacing = g.min_spacing;
    snprintf(file_name, name_cap, "%s", g.file_name.c_str());
    snprintf(contig_name, name_cap, "%s", g.first_contig_name.c_str());
}
void sylph_host_syldb_copy(void* h, uint64_t i, uint64_t* kmers, uint64_t* tracked) {
    const GenomeSketch& g = (*(std::vector<GenomeSketch>*)h)[i];
    memcpy(kmers, g.genome_kmers.data(), g.genome_kmers.size() * 8);
    if (g.pseudotax_tracked_nonused_kmers && tracked) memcpy(tracked, g.pseudotax_tracked_nonused_kmers->data(), g.pseudotax_tracked_nonused_kmers->size() * 8);
}
void sylph_host_syldb_free(void* h) { delete (std::vector<GenomeSketch>*)h; }

// FASTA/FASTQ(+gzip) record stream digest, through the plain reader (threaded = 0) or through the reader thread + chunk
// queue of the host feed (threaded = 1): number of records, parse errors, total bases, FNV-1a over every sequence byte and
// every record length, length of the first header.  Returns -1 if the file cannot be opened.
int sylph_host_fastx_digest(const char* path, int threaded, uint64_t* n_records, uint64_t* n_errors, uint64_t* n_bases,
                            uint64_t* digest) {
    uint64_t h = 1469598103934665603ull, nr = 0, ne = 0, nb = 0;
    auto mix = [&](const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; } };
    auto mix_len = [&](uint64_t l) { mix((const uint8_t*)&l, 8); };
    try {
        if (threaded) {
            ChunkStream cs(path);
            const uint8_t* seq = nullptr;
            uint32_t len = 0;
            for (;;) {
                const ChunkStream::Kind k = cs.next(seq, len);
                if (k == ChunkStream::END) break;
                if (k == ChunkStream::ERR) { ne++; if (ne > 1000) break; continue; }
                mix(seq, len); mix_len(len); nr++; nb += len;
            }
        } else {
            FastxReader r(path);
            FastxRecord rec;
            for (;;) {
                bool ok = false;
                try { ok = r.next(rec); } catch (const Error&) { ne++; if (ne > 1000) break; continue; }
                if (!ok) break;
                mix((const uint8_t*)rec.seq.data(), rec.seq.size()); mix_len(rec.seq.size()); nr++; nb += rec.seq.size();
            }
        }
    } catch (const Error&) { return -1; }
    *n_records = nr; *n_errors = ne; *n_bases = nb; *digest = h;
    return 0;
}

// the same digest through the block-parallel index of an uncompressed 4-line FASTQ (feed.cpp); *ok = 0 when the file is not
// eligible (the drivers then use the sequential reader)
int sylph_host_fastq_index_digest(const char* path, unsigned threads, int* ok, uint64_t* n_records, uint64_t* n_bases, uint64_t* digest) {
    uint64_t h = 1469598103934665603ull, nb = 0;
    auto mix = [&](const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; } };
    // threads with bit 31 set: the two steps of the device-side route's gzip handling — the text first (mapping / inflate, no index:
    // text_ready), the index afterwards (build_index) — which must end where the one-step constructor ends
    const bool two_steps = (threads & 0x80000000u) != 0;
    threads &= 0x7FFFFFFFu;
    FastqIndex ix(path, threads, !two_steps);
    if (two_steps) {
        if (ix.ok) return 1;                           // nothing may have been indexed yet
        if (ix.text_ready) ix.build_index(threads);
    }
    *ok = ix.ok ? 1 : 0;
    if (!ix.ok) return 0;
    for (size_t i = 0; i < ix.n_records(); i++) {
        mix(ix.data + ix.seq_off[i], ix.seq_len[i]);
        const uint64_t l = ix.seq_len[i];
        mix((const uint8_t*)&l, 8);
        nb += l;
    }
    *n_records = ix.n_records(); *n_bases = nb; *digest = h;
    return 0;
}

// index only (tools/feed_bench.py times the parser alone with it)
int sylph_host_fastq_index_count(const char* path, unsigned threads, uint64_t* n_records) {
    FastqIndex ix(path, threads);
    *n_records = ix.ok ? ix.n_records() : 0;
    return ix.ok ? 1 : 0;
}

// parallel_gunzip (pgunzip.cpp) on a file, for the tests: -> 1 and the inflated bytes' length + CRC-32 (and, if out != null with
// room for them, the bytes) when the parallel path took the file; 0 when it declined (the feed then reads sequentially); -1: no file
int sylph_host_pgunzip(const char* path, unsigned threads, uint64_t* out_len, uint32_t* out_crc, uint8_t* out, uint64_t out_cap) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    std::vector<uint8_t> gz;
    uint8_t buf[1 << 16];
    for (size_t r; (r = fread(buf, 1, sizeof(buf), f)) > 0;) gz.insert(gz.end(), buf, buf + r);
    fclose(f);
    uint8_t* o = nullptr;
    size_t n = 0, n_map = 0;
    if (!sylph_host::parallel_gunzip(gz.data(), gz.size(), threads, &o, &n, &n_map, 0)) return 0;
    if (out_len) *out_len = n;
    if (out_crc) {
        uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
        for (size_t q = 0; q < n; q += 1u << 30) c = (uint32_t)crc32(c, o + q, (uInt)std::min<size_t>(n - q, 1u << 30));
        *out_crc = c;
    }
    if (out && out_cap >= n) memcpy(out, o, n);
    sylph_host::inflated_release(o, n_map);
    return 1;
}

// the CPUs the feed believes it may use (hardware threads cut down to the affinity mask and the cgroup CPU quota) and the parse
// threads it derives from them: for the tests
unsigned sylph_host_effective_cpus(void) { return sylph_host::effective_cpus(); }
unsigned sylph_host_parse_threads(void) { return sylph_host::parse_threads(); }
}  // extern "C"
