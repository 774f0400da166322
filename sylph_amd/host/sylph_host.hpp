// sylph_host.hpp — host side above the C ABI (include/sylph_hip.h), mirroring the reference's operator interface
// for the hot path: same names, argument meaning and error behaviour as the Rust functions it stands in for
// (cited as file:line under /root/reference/src).  The reference is compiled Rust and no Rust toolchain exists in this
// image, so the host is C++17.  Everything heavy is delegated to libsylph_hip.so; what stays here is what the
// north_star leaves on the host: record parsing, sequential f64 bookkeeping, on-disk formats, coverage/ANI statistics
// (contain.rs:657-813, inference.rs:207-242), profile reassignment and TSV formatting.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <atomic>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/sylph_hip.h"

namespace sylph_host {

// constants.rs:1-17
constexpr double CUTOFF_PVALUE = 0.9999999999;
constexpr size_t SAMPLE_SIZE_CUTOFF = 25;
constexpr double MEDIAN_ANI_THRESHOLD = 2.;
constexpr double MIN_ANI_DEF = 0.9, MIN_ANI_P_DEF = 0.95;
constexpr double MAX_MEDIAN_FOR_MEAN_FINAL_EST = 15.;
constexpr double DEFAULT_FPR = 0.0001;
constexpr double MED_KMER_FOR_ID_EST = 3.;   // constants.rs:17
constexpr const char* QUERY_FILE_SUFFIX = ".syldb";
constexpr const char* SAMPLE_FILE_SUFFIX = ".sylsp";

struct Error { int code; std::string msg; };   // thrown; the CLI maps it to log::error! + exit(code)

// types.rs:145-155.  kmer_counts kept as two parallel arrays in ascending k-mer order (a keyed multiset is all any
// consumer of the reference's FxHashMap can observe).
struct SequencesSketch {
    std::vector<uint64_t> kmers;
    std::vector<uint32_t> counts;
    uint64_t c = 0, k = 0;
    std::string file_name;
    std::optional<std::string> sample_name;
    bool paired = false;
    double mean_read_length = 0.;
};

// types.rs:163-173
struct GenomeSketch {
    std::vector<uint64_t> genome_kmers;
    std::optional<std::vector<uint64_t>> pseudotax_tracked_nonused_kmers;
    std::string file_name, first_contig_name;
    uint64_t c = 0, k = 0, gn_size = 0, min_spacing = 0;
    // A sketch read as a VIEW of a mapped .syldb (read_syldb_views): the two vectors above stay empty, the k-mers lie in the file
    // mapping (unaligned little-endian u64s) that `mapping` keeps alive.  `profile` / `query` never look at a genome's k-mers on the
    // host — they go to the GPU, the statistics need their number — so a 13 GB database is not copied into 113,104 vectors first.
    const uint8_t* view_kmers = nullptr;
    const uint8_t* view_tracked = nullptr;
    uint64_t view_n = 0, view_tn = 0;
    bool view = false, view_has_tracked = false;
    std::shared_ptr<void> mapping;
    size_t n_kmers() const { return view ? view_n : genome_kmers.size(); }
    size_t n_tracked() const { return view ? view_tn : (pseudotax_tracked_nonused_kmers ? pseudotax_tracked_nonused_kmers->size() : 0); }
    bool has_tracked() const { return view ? view_has_tracked : (bool)pseudotax_tracked_nonused_kmers; }
    const uint8_t* kmers_bytes() const { return view ? view_kmers : (const uint8_t*)genome_kmers.data(); }
    const uint8_t* tracked_bytes() const { return view ? view_tracked : (pseudotax_tracked_nonused_kmers ? (const uint8_t*)pseudotax_tracked_nonused_kmers->data() : nullptr); }
};

// ---- on-disk formats: bincode 1.3.3 default options (little-endian, fixed-width ints, u64 lengths) ----
void write_sylsp(const std::string& path, const SequencesSketch& s);          // sketch.rs:360,411
SequencesSketch read_sylsp(const std::string& path);                          // contain.rs:559
void write_syldb(const std::string& path, const std::vector<GenomeSketch>& g); // sketch.rs:474
std::vector<GenomeSketch> read_syldb(const std::string& path);                // contain.rs:495
// the same without copying a k-mer: one pass over the record headers of the mapped file (contain.rs:492-500 is the reference's
// single-threaded 13 GB bincode read, the dominant cost of a one-sample profile at GTDB scale)
std::vector<GenomeSketch> read_syldb_views(const std::string& path);

// ---- FASTA/FASTQ (+gzip) records with needletail 0.5.1 semantics: seq() without newlines, id() = whole header ----
struct FastxRecord { std::string id; std::string seq; };
class FastxReader {
   public:
    explicit FastxReader(const std::string& path);   // throws Error if the file cannot be opened / is not fasta/fastq
    ~FastxReader();
    bool next(FastxRecord& rec);                      // false at EOF; throws Error on a malformed record
   private:
    void* gz_ = nullptr;
    std::string buf_;
    size_t pos_ = 0;
    bool eof_ = false, fastq_ = false, started_ = false;
    bool getline(std::string& line);
    bool next_fastq_in_buffer(FastxRecord& rec);
    std::string pending_;
    bool has_pending_ = false;
};
// ---- host feed (feed.cpp): one reader thread per file -> chunks of records; page-locked batches for the GPU ----
struct RecordChunk {
    static constexpr uint32_t ERR = 0xFFFFFFFFu;   // len entry of a record that failed to parse
    std::vector<uint8_t> bases;                    // sequences of the chunk's records, concatenated
    std::vector<uint32_t> len;
    std::string first_id;
};
class ChunkStream {
   public:
    explicit ChunkStream(const std::string& path);   // throws Error like FastxReader; starts the reader thread
    ~ChunkStream();
    enum Kind { REC, ERR, END };
    Kind next(const uint8_t*& seq, uint32_t& len);    // records in file order; seq stays valid until the next call
   private:
    struct Impl;
    std::unique_ptr<Impl> p_;
};
// Block-parallel index of an UNCOMPRESSED, strictly 4-line FASTQ file (feed.cpp): the file is mapped, cut into byte ranges, and
// one thread per range finds its first record boundary and records where every sequence lies.  `ok` is false when the file is
// anything else (gzip, FASTA, blank lines, multi-line records, a quality line of the wrong length, ...): the caller then falls
// back to the sequential reader, which reproduces needletail's record/error semantics exactly.
struct FastqIndex {
    bool ok = false;
    const uint8_t* data = nullptr;   // the mapping (owned)
    size_t size = 0;
    size_t map_bytes = 0;            // anonymous: what the mapping really spans (a recycled buffer may be larger than the file)
    bool anonymous = false;          // `data` is the inflated copy of a gzip file: it goes back to the buffer pool (inflated_release)
    void release_behind(size_t byte_offset) const;   // the feed is done with everything before byte_offset
    mutable size_t released_ = 0;                    // bytes of an inflated copy already given back
    mutable std::vector<std::thread> releasers_;     // ... by these threads (page freeing off the feed's critical path)
    std::vector<uint64_t> seq_off;   // byte offset of every record's sequence line
    std::vector<uint32_t> seq_len;
    std::vector<uint64_t> cum;       // cum[i] = sequence bases of records [0, i) (n_records + 1 entries)
    bool text_ready = false;         // data[0, size) is the file's complete plain text (the mapping, or the inflated copy of a gzip file) and begins with '@'
    std::string path_;
    FastqIndex() = default;
    // build = false: stop behind the mapping / the inflate (text_ready says whether there is text) — the device-side route sends the text
    // as it is; build_index() finishes the job on the host when that route is not taken
    FastqIndex(const std::string& path, unsigned threads, bool build = true);
    void build_index(unsigned threads);
    ~FastqIndex();
    FastqIndex(const FastqIndex&) = delete;
    FastqIndex& operator=(const FastqIndex&) = delete;
    size_t n_records() const { return seq_len.size(); }
};
// pgunzip.cpp: parallel inflate of a single-member gzip file (block-start search + window-free decoding + CRC check); false = not
// done (any doubt at all): the caller reads the file sequentially.  *out: *out_size bytes in an anonymous mapping of *out_map bytes
// from inflated_acquire (give it back with inflated_release).
bool parallel_gunzip(const uint8_t* gz, size_t n, unsigned threads, uint8_t** out, size_t* out_size, size_t* out_map, size_t memory_budget);
// Inflated copies of gzip files are GBs of anonymous memory each: a fresh mapping costs a page fault and a page of zeroes per 4 KiB
// (90 ms per GB on the GPU box, as much as inflating it).  The buffers of files the feed is done with are kept (up to a quarter of
// the index memory budget, at most 16 GB) and handed to the next file: its inflate writes into pages that are already there.
void* inflated_acquire(size_t bytes, size_t* map_bytes);
void inflated_release(void* p, size_t map_bytes);
// CPUs this process may really use: the hardware threads, cut down to the cgroup's CPU quota (cpu.max / cfs_quota_us) and to the
// affinity mask.  A container that shows 256 hardware threads under a quota of 16 CPUs runs 128 decoder threads for 12 ms of every
// 100 ms period and is frozen for the other 88 (seen on the GPU box: per-stretch inflate times of 10 ms and 100 ms side by side):
// every thread count of the feed starts from this number, not from hardware_concurrency().
unsigned effective_cpus();
unsigned parse_threads();   // worker threads of the parallel feed PER sample thread: SYLPH_HIP_PARSE_THREADS, else a quarter of the hardware threads (8..64), divided by set_parse_share
void set_parse_share(unsigned sample_threads);   // the `-t` sample threads that run a feed each share the parse-thread budget
constexpr size_t MAX_SAMPLE_THREADS = 16;        // each sample thread owns a GPU context + ~0.5 GB of page-locked batch buffers
// Bytes a FastqIndex may hold in ANONYMOUS memory (the inflated copy of a blocked-gzip file); 0 = no limit.  The drivers set it
// from MemAvailable and the number of files they index at the same time: beyond it the file goes to the sequential reader,
// which runs in constant memory.
void set_index_memory_budget(size_t bytes);
void set_no_more_inflates(bool v);               // no further input file of the command will be inflated: inflated copies give their pages back as they are consumed
size_t index_memory_budget();

// BYTE_TO_SEQ + 2-bit packing into one part of a shared stream (pack2bit.cpp)
class Pack2Bit {
   public:
    Pack2Bit(uint8_t* out, uint64_t base_offset);   // `out` = start of the whole stream; this writer begins at base `base_offset`
    // `readable`: bytes that may be read from seq on (>= len; more lets the last bases of the record take the vector path too)
    void append(const uint8_t* seq, size_t len, size_t readable = 0);
    void finish();
    // the bytes this part shares with its neighbours (not stored by the writer): merge_pack_edges puts them together
    uint64_t first_byte = 0, last_byte = 0;
    uint8_t first_val = 0, last_val = 0;
    bool first_partial = false, last_partial = false;
   private:
    void emit(uint64_t word, unsigned n_bytes);
    void push64(uint64_t v);
    void push_bits(uint64_t v, unsigned nbits);
    uint8_t* outp_;
    uint8_t* base_ = nullptr;
    uint64_t acc_ = 0;
    unsigned nacc_ = 0;      // pending bits in acc_ (its most significant ones)
    bool skip_first_ = false;
};
void merge_pack_edges(uint8_t* out, const Pack2Bit* writers, size_t n);

class PinnedBatch {   // flat bases + offsets in page-locked memory (sylph_pinned_alloc), pushed with SYLPH_MEM_HOST_PINNED
   public:
    static constexpr size_t BATCH_BASES = 256u << 20, BATCH_RECS = 4u << 20;
    PinnedBatch();
    ~PinnedBatch();
    // appends one record (pair = false) or the two mates of a pair, flushing to the session first when the batch is full
    void add(sylph_sketch* sk, const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, bool pair);
    void flush(sylph_sketch* sk);
    void prealloc() { if (!bases_) reserve(BATCH_BASES, BATCH_RECS); }   // the page-locked allocation itself takes ~50 ms
    // records [i0, i1) of an indexed file (single-end), or pairs [i0, i1) of two indexed files interleaved mate 1, mate 2,
    // PACKED (2 bits per base) into page-locked slot 0 or 1 by `threads` workers — no GPU call: the other slot may be on its way
    // to the device meanwhile; push_packed hands the slot over (sylph_sketch_push_enc, SYLPH_ENC_2BIT, SYLPH_MEM_HOST_PINNED)
    void gather_packed(int slot, const FastqIndex& a, const FastqIndex* b, const std::vector<uint64_t>& cum_a,
                       const std::vector<uint64_t>* cum_b, size_t i0, size_t i1, unsigned threads);
    void push_packed(sylph_sketch* sk, int slot);
    void prealloc_packed();
    // round 5: batches gathered + packed into PAGEABLE memory (no GPU runtime involved) while the context of a process's first sample
    // is still coming up; push_early hands them over in order (SYLPH_MEM_HOST: the library's staging buffers) once there is a session
    void gather_packed_early(const FastqIndex& a, const FastqIndex* b, const std::vector<uint64_t>& cum_a, const std::vector<uint64_t>* cum_b,
                             size_t i0, size_t i1, unsigned threads);
    void push_early(sylph_sketch* sk);
   private:
    void reserve(size_t bases_cap, size_t recs_cap);
    uint8_t* bases_ = nullptr;
    uint64_t* off_ = nullptr;
    size_t cap_bases_ = 0, cap_recs_ = 0, n_bases_ = 0, n_recs_ = 0;
    struct Packed { uint8_t* bytes = nullptr; uint64_t* off = nullptr; size_t cap_bytes = 0, cap_recs = 0, n_bases = 0, n_recs = 0; bool pageable = false; };
    Packed pk_[2];
    std::vector<Packed> early_;
    void reserve_packed(Packed& p, size_t bytes, size_t recs);
    void free_packed(Packed& p);
    void gather_into(Packed& p, const FastqIndex& a, const FastqIndex* b, const std::vector<uint64_t>& cum_a, const std::vector<uint64_t>* cum_b,
                     size_t i0, size_t i1, unsigned threads);
};
// Round 5: the TEXT of plain FASTQ files sent to the device, where the library finds the records (sylph_fastq_index): one reusable
// uploader per engine (two page-locked chunks the parse threads fill with pread while the other travels), the files of a sample laid
// side by side at 16-byte boundaries in one device buffer.
class TextUploader {
   public:
    struct Text { const uint8_t* dev = nullptr; uint64_t bytes = 0; };
    ~TextUploader();
    void prepare(sylph_ctx* ctx);   // page-locks the two chunks ahead of the first send (the engine's background bring-up)
    bool warm() const { return sent_.load(); }   // a text has travelled before: the device-side buffers of the route exist
    // false: some file is not a candidate (not a regular file, empty, gzip magic, does not begin with '@') — nothing was sent
    bool send(sylph_ctx* ctx, const std::vector<std::string>& files, unsigned threads, std::vector<Text>& out);
    // the same for texts that lie in host memory already (the inflated copies of gzip files)
    struct Mem { const uint8_t* p; uint64_t bytes; };
    bool send(sylph_ctx* ctx, const std::vector<Mem>& texts, unsigned threads, std::vector<Text>& out);
   private:
    struct Src { int fd; const uint8_t* mem; uint64_t size; };
    bool send_sources(sylph_ctx* ctx, const std::vector<Src>& src, unsigned threads, std::vector<Text>& out);
    std::mutex mu_;
    sylph_upload* up_ = nullptr;
    std::atomic<bool> sent_{false};
};
// The device-side route for plain FASTQ (TextUploader + sylph_fastq_*), taken by every sample whose engine is already up unless
// SYLPH_HIP_FEED_DEVICE=0: the host feed's index + gather + pack scale with the CPUs the process may use (73-125 ms per warm 1 Gbp pair
// on 16, 104-206 on 4), the text's trip over PCIe does not (58-65 ms per pair all in; profiles/r05_feed_device_route.txt).  Samples
// whose text exceeds SYLPH_HIP_FEED_DEVICE_MAX_GB (default 16) stay with the host feed, which works through a sample in batches.
bool device_feed_enabled();
bool is_fastq(const std::string& f);   // sketch.rs:95
bool is_fasta(const std::string& f);   // sketch.rs:109

// ---- sketching (GPU through the C ABI) ----
bool fast_exit();                    // `sylph-hip sketch` leaves through _exit once its outputs are written (not with SYLPH_HIP_CLEAN_EXIT=1)
void join_background();              // waits for the host side's fire-and-forget threads (unmapping behind a sample)
void trace_mark(const char* what);   // SYLPH_HIP_FEED_TRACE: a line with the milliseconds since the host library was loaded
struct Engine {   // one GPU context shared by the drivers
    int device = -1;
    PinnedBatch batch;   // reused by every sample sketched through this engine
    TextUploader text;   // ... and the uploader of the device-side FASTQ route (commands.cpp sketch_fastq_on_device)
    std::atomic<bool> warm_text_route{false};   // set before the bring-up's warm-up sample: it goes the device FASTQ route's way (the first sample will)
    std::atomic<bool> defer_pinned{false};   // set before the bring-up reaches them: `batch` and `text` page-lock their buffers at first use instead
    // GPU bring-up (runtime initialisation, context, page-locked batch, first-use loading of the sketch kernels: ~0.3 s) runs on a
    // background thread from the moment the engine exists, so that it overlaps with argument handling and the indexing of the
    // first input file; context() waits for it (and rethrows its error).
    explicit Engine(int device = -1);
    ~Engine();
    sylph_ctx* context();          // waits until the context exists and the sketch kernels are loaded
    bool ready() const;            // context() would not block
    void wait_pinned();            // ... and until the page-locked double buffers of `batch` are there (allocated behind `ready`)
   private:
    sylph_ctx* ctx_ = nullptr;
    std::thread init_;
    std::string init_error_;
    int init_code_ = 0;
    struct Gate;                   // ready flag + condition variable
    std::unique_ptr<Gate> gate_;
};
// sketch.rs:897 / :771 / :550 / :481 — return nullopt where the reference returns None (warn + skip).
std::optional<SequencesSketch> sketch_sequences_needle(Engine& e, const std::string& read_file, uint64_t c, uint64_t k,
                                                       std::optional<std::string> sample_name, bool no_dedup);
std::optional<SequencesSketch> sketch_pair_sequences(Engine& e, const std::string& read_file1, const std::string& read_file2,
                                                     uint64_t c, uint64_t k, std::optional<std::string> sample_name,
                                                     bool no_dedup, double dedup_fpr);
std::optional<GenomeSketch> sketch_genome(Engine& e, uint64_t c, uint64_t k, const std::string& ref_file, uint64_t min_spacing,
                                          bool pseudotax);
std::vector<GenomeSketch> sketch_genome_individual(Engine& e, uint64_t c, uint64_t k, const std::string& ref_file,
                                                   uint64_t min_spacing, bool pseudotax);

// ---- statistics (host, f64) ----
enum class AdjustStatus { Low, High, Lambda };
struct AniResult {   // types.rs:184-203
    double naive_ani = 0, final_est_ani = 0, final_est_cov = 0, mean_cov = 0, median_cov = 0;
    size_t contain_count = 0, n_kmers = 0;
    AdjustStatus lambda_status = AdjustStatus::Low;
    double lambda = 0;
    std::optional<double> ani_ci_lo, ani_ci_hi, lambda_ci_lo, lambda_ci_hi;
    size_t genome_index = 0;
    std::optional<double> rel_abund, seq_abund;
    std::optional<size_t> kmers_lost;
};
struct ContainArgs {   // the cmdline.rs:88-160 fields the statistics read
    double min_count_correct = 3., min_number_kmers = 50.;
    std::optional<double> minimum_ani;
    bool pseudotax = false, no_ci = false, no_adj = false, mean_coverage = false, estimate_unknown = false;
    bool debug_f64 = false;   // --debug-f64: float columns as %.17g (test aid, not in the reference)
    double redundant_ani = 99.0;
    std::optional<double> seq_id;   // -I/--read-seq-id (per cent): overrides the automatic read identity of -u (contain.rs:275)
    uint64_t threads = 3;   // -t: genomes whose statistics run concurrently (cmdline.rs default 3)
};
std::optional<double> ratio_lambda(const std::vector<uint32_t>& full_covs, double min_count_correct);   // inference.rs:207
std::optional<double> ani_from_lambda(std::optional<double> lambda, double k, const std::vector<uint32_t>& full_cov);  // contain.rs:817
double poisson_cdf(double lambda, uint64_t x);   // statrs Poisson::cdf = Q(x+1, lambda) (third party; parity unpinned)
// statistics half of get_stats (contain.rs:657-813): covs = the non-zero sample counts of the genome's k-mers found
// in the sample (any order), n_genome_kmers = genome_kmers.len(), kmers_lost = Some(..) in the winner pass.
std::optional<AniResult> stats_from_covs(const ContainArgs& args, std::vector<uint32_t> covs, size_t n_genome_kmers, uint64_t k,
                                         std::optional<size_t> kmers_lost);

// ---- commands ----
struct SketchArgs {   // cmdline.rs:28-86
    std::vector<std::string> files, reads, genomes, first_pair, second_pair;
    std::optional<std::vector<std::string>> sample_names;
    std::string db_out_name = "database", sample_output_dir = "./";
    bool individual = false, no_dedup = false, no_pseudotax = false;
    bool exact_dedup = false;   // --exact-dedup (not in the reference): the exact marker set where the reference would use its cuckoo filter
    uint64_t k = 31, c = 200, min_spacing_kmer = 30, threads = 3;   // -t: samples in flight (cmdline.rs: default 3)
    double fpr = DEFAULT_FPR;
    int gpus = 1;               // --gpus N|all (-1): the samples' workers are dealt to N GPUs (not in the reference: its rayon pool spans the machine, sketch.rs:313, :371)
    std::optional<std::string> list_sequence, list_reads, list_genomes, list_first_pair, list_second_pair, list_sample_names;
};
struct ContainCmdArgs : ContainArgs {   // cmdline.rs:88-160
    std::vector<std::string> files, reads, first_pair, second_pair;
    std::optional<std::string> file_list, out_file_name;
    uint64_t k = 31, c = 200, min_spacing_kmer = 30;
    bool individual = false;
    bool exact_dedup = false;   // --exact-dedup: raw pairs are deduplicated with the exact marker set (the reference forces its cuckoo filter, contain.rs:591)
    int gpus = 1;               // --gpus N|all (-1): raw samples are spread over N GPUs, each with a replica of the database (not in the reference: its rayon pool spans the machine by itself)
};
// a10: paired input is deduplicated as the reference does — behind its cuckoo filter for --fpr != 0
// (the default; raw pairs in profile / query always: contain.rs:591) — unless the caller asks for the exact set: --exact-dedup,
// SYLPH_HIP_EXACT_DEDUP=1 in the environment (neither is in the reference), or --fpr 0 where the command has it
bool exact_dedup_accepted(bool flag);
// --estimate-unknown (contain.rs:901-951, :377-408)
std::optional<double> get_kmer_identity(const SequencesSketch& S, bool estimate_unknown);
void estimate_true_cov(std::vector<AniResult>& results, std::optional<double> kmer_id_opt, bool estimate_unknown, double read_length, uint64_t k);
double estimate_covered_bases(const std::vector<AniResult>& results, const std::vector<GenomeSketch>& genomes, const SequencesSketch& S,
                              double read_length, uint64_t k);
// `inspect` (inspect.rs:117): YAML summary of *.syldb / *.sylsp files; host only
struct InspectArgs {   // cmdline.rs:166-173
    std::vector<std::string> files;
    std::optional<std::string> out_file_name;
};
int inspect(const InspectArgs& args, FILE* out);
std::string inspect_f32(float v);                 // the scalars as serde_yaml 0.9 / ryu would print them (tests)
std::string inspect_f64(double v);
std::string inspect_str(const std::string& s);
int sketch(Engine& e, const SketchArgs& args);                              // sketch.rs:276; returns the exit code
int contain(Engine& e, ContainCmdArgs args, bool pseudotax_in, FILE* out);  // contain.rs:115 (query: false, profile: true)

}  // namespace sylph_host
