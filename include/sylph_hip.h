/*
 * sylph_hip.h — C ABI of the MI355X (gfx950) sketch + profile engine.
 *
 * This is the drop-in boundary for the hot path of bluenote-1577/sylph v0.8.1.  The reference exposes no
 * plugin/FFI API; each entry point below replaces one in-crate Rust function (cited as file:line under
 * /root/reference/src) at the seam where a Rust `extern "C"` binding would be added (see INTEGRATION.md for
 * the `rust/ffi.rs` stub).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions (mirroring the reference's calling conventions):
 *  - inputs are borrowed for the duration of the call; outputs named `**out` are allocated by the library and
 *    released with sylph_free(); handles are released with their *_destroy();
 *  - every function returns SYLPH_OK (0) or a negative error code and never aborts/unwinds across the boundary
 *    (the reference warn+skips or exit(1)s; a release build is panic=abort, Cargo.toml:40); the message for the
 *    last error on the calling thread is sylph_last_error();
 *  - a sylph_ctx may be used from several host threads (rayon workers call concurrently, sketch.rs:313,371;
 *    contain.rs:267,284): calls on one ctx are serialised internally, use one ctx per worker for overlap;
 *  - results never depend on thread interleaving: tables are returned in ascending k-mer order.
 *  - `mem` says where the caller's input arrays live: SYLPH_MEM_HOST (ordinary host memory, copied H2D by the
 *    library through its own pinned staging buffers), SYLPH_MEM_HOST_PINNED (page-locked memory obtained from
 *    sylph_pinned_alloc(), copied H2D directly — the host feed of a parser that fills batches in place; accepted
 *    by sylph_sketch_push[_n]) or SYLPH_MEM_DEVICE (already resident in this GPU's HBM; zero-copy).
 */
#ifndef SYLPH_HIP_H
#define SYLPH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYLPH_OK 0
#define SYLPH_ERR_INVALID (-1)   /* bad argument (the reference would panic!/exit(1)) */
#define SYLPH_ERR_HIP (-2)       /* HIP runtime error */
#define SYLPH_ERR_NOMEM (-3)
#define SYLPH_ERR_STATE (-4)     /* call made in the wrong session state */
#define SYLPH_ERR_FORMAT (-5)    /* the input is not what this entry point parses (sylph_fastq_index): read it with the host reader */

/* Which k-mers of a sequence are hashed.
 * SYLPH_SEED_SCALAR      = seeding.rs:86-146 fmh_seeds (every k-mer).
 * SYLPH_SEED_AVX2_COMPAT = avx2_seeding.rs:33-148 extract_markers_avx2, the path the reference takes on every
 *                          AVX2 x86 host (sketch.rs:53-63): k-mers with start >= 4*((L-k+1)/4) are never hashed,
 *                          sequences shorter than k+1 (reads) / 2k (genome contigs) yield nothing. Default. */
#define SYLPH_SEED_SCALAR 0
#define SYLPH_SEED_AVX2_COMPAT 1

#define SYLPH_READS_SINGLE 0     /* sketch_sequences_needle, sketch.rs:897 */
#define SYLPH_READS_PAIRED 1     /* sketch_pair_sequences, sketch.rs:771: the exact pair set (--fpr 0) unless the session option "dedup_fpr" selects the filter */

#define SYLPH_MEM_HOST 0
#define SYLPH_MEM_DEVICE 1
#define SYLPH_MEM_HOST_PINNED 2   /* host memory from sylph_pinned_alloc(): copied H2D without the staging memcpy */

typedef struct sylph_ctx sylph_ctx;        /* one GPU + one HIP stream + scratch memory */
typedef struct sylph_sketch sylph_sketch;  /* a read-sketch session (one sample) */
typedef struct sylph_db sylph_db;          /* a genome database (shard) resident in HBM */

int sylph_version(void);
/* gfx950 devices this process sees (0: none, or the runtime failed): what `sylph-hip profile --gpus all` spreads its replicas over —
 * the reference sizes its rayon pool from the machine the same way (contain.rs:136-139). */
int sylph_device_count(void);
const char *sylph_last_error(void);
void sylph_free(void *p);
/* Page-locked host memory for SYLPH_MEM_HOST_PINNED inputs (host feed); release with sylph_pinned_free(). */
int sylph_pinned_alloc(uint64_t bytes, void **out);
void sylph_pinned_free(void *p);

/* device < 0: current device.  stream: a hipStream_t to launch on (e.g. torch's current stream), or NULL for a
 * private stream. */
int sylph_ctx_create(int device, void *stream, sylph_ctx **out);
void sylph_ctx_destroy(sylph_ctx *ctx);
int sylph_ctx_synchronize(sylph_ctx *ctx);

/* ---- staged host -> device upload ----------------------------------------------------------------------------------------
 * A device buffer of `bytes` bytes filled from the host through two page-locked chunks the LIBRARY owns: the caller fills one
 * (sylph_upload_chunk hands it out, waiting for the copy that last used it) while the other travels (sylph_upload_commit queues the
 * copy of the chunk's first n bytes to the next device offset).  For data that does not lie in one piece on the host — the
 * genome_kmers vectors of a 13 GB .syldb file (types.rs:163-173; contain.rs:492-500 deserialises them on one thread, the dominant
 * cost of a one-sample `sylph profile`) are interleaved with names and tracked k-mers: the host gathers them from the file MAPPING
 * straight into the chunks with all its threads, and hands the device pointer to sylph_db_upload / sylph_db_attach_tracked with
 * SYLPH_MEM_DEVICE.  sylph_upload_finish waits for the last copy and returns the device pointer, valid until sylph_upload_destroy. */
typedef struct sylph_upload sylph_upload;
int sylph_upload_begin(sylph_ctx *ctx, uint64_t bytes, uint64_t chunk_bytes, sylph_upload **out);
int sylph_upload_chunk(sylph_upload *u, void **chunk, uint64_t *cap);
int sylph_upload_commit(sylph_upload *u, uint64_t n);
int sylph_upload_finish(sylph_upload *u, const void **device_ptr);
/* The same uploader for another `bytes` bytes (a feed that sends one sample's text after the other): chunks, stream and — where it
 * is large enough — the device buffer are kept; the pointer sylph_upload_finish returned before is no longer valid. */
int sylph_upload_restart(sylph_upload *u, uint64_t bytes);
void sylph_upload_destroy(sylph_upload *u);

/* Tuning / test knobs (not needed for normal use).  "finish" = "auto" (default: bucket partition + in-LDS replay,
 * falling back to the device-wide sort path when a bucket does not fit), "generic" (always the device-wide path) or
 * "bucket" (error instead of falling back).  "seeds" = "auto" (default: the read-per-lane kernel for short-read batches, the
 * position kernel with per-tile ordered slots otherwise), "slots" (always the position kernel with ordered slots) or
 * "unordered" (position kernel with LDS-staged atomics + radix sort by position).  "bucket_target" = mean number of
 * occurrences per replay bucket aimed for, "16".."256".  "shard_reduce" = "alltoall" (default) or "allgather": how the hits of a
 * sharded batch (sylph_db_contain_batch_sharded) reach the rank that owns the sample — one all-to-all of exactly the hits each rank
 * needs, or ONE all-gather of every rank's whole hit buffer padded to the longest (W x the bytes, one collective; set on the database's
 * context, or through sylph_pipeline_set_option).  "index_lambda" = postings per 64-byte bucket line of a database
 * index aimed for ("1".."8", default 4 — 29 GB at GTDB-R220 scale, 3 was 38.5 GB for 3 % less probe time; applies to databases uploaded afterwards), "index_pass_max" = postings sorted per pass
 * of the index build (tests lower it to force several passes), "push_chunk_bytes" = bytes of bases per chunk of a host batch
 * (default 64 MiB), "reads_wg_per_cu" = workgroups of the read-per-lane kernel per CU, each looping over blocks of reads
 * ("0".."64"; default 0 = one workgroup per block, measured 10 % faster than 8 looping workgroups per CU).
 * "profile_only" = "seeds" | "seeds,probe" | ... | "all": the kernel families sylph_ctx_profile times from now on (default all).  A timed
 * family costs two event records per launch group on its stream — 1 % of a pipelined sample, 4 % of a sample run alone, with all of
 * them on (profiles/r05_ab_timers.txt): bench.py times only the dominant kernel inside its timed region.
 * "reads_tail_pct" = "0".."50": inside a pipeline's seeding turn (one seeding kernel at a time), the share of a sample's blocks of reads
 * that is launched separately BEHIND the turn's event, so that the next sample's seeding kernel starts while they run ("0" = one launch;
 * default "10" since round 6: profiles/r06_ab_latency.txt).
 * "reads_hash" = "0" | "1" | "2" | "-1": how the read-per-lane kernel spells the hash and the threshold test in its k-mer loop —
 * the compiler's own lowering, the hand-scheduled 64-bit one, or the last hash step and the test on the high word only (a superset of
 * the seeds; the kernel's second pass, which hashes every candidate exactly anyway, prunes it); same tables all three, "-1" = the
 * build's default.  "reads_slack" = n (tests only): widens that superset by n high-word values so that the pruning road is common.
 * "plain_records" = "1" (default) / "0": single-end batches whose records carry no dedup marker (reads above 400 bases, or a
 * --no-dedup session) keep only the hashes of their seed occurrences and are counted without occurrence records; "0" writes
 * the records for every batch (A/B and tests: the tables are identical).
 * "cu_mask" = "lo:hi" / "all": the context's OWN stream is recreated on the compute units [lo, hi) of the device's CU-mask numbering
 * (A/B knob; restricting the sketch workers' streams lost 25-40 % in r04: profiles/r04_ab_pipeline_sweep.txt).
 * "stream_priority" = "high" | "normal" | "low": the context's OWN stream is recreated at that priority (an error on a context that
 * runs on the caller's stream: create that stream at the priority wanted); no measurable effect on a pipeline's rate or its
 * completion intervals (profiles/r06_ab_stream_priority.txt).
 * "fail_next_shard_probe" = "1": fault injection for the tests — the next sylph_db_contain_batch_sharded on this context fails
 * in its probe, between the collectives (every rank of the batch must then return the same error, nobody may hang). */
int sylph_ctx_set_option(sylph_ctx *ctx, const char *key, const char *value);

/* Per-kernel timing (hipEvent pairs on the ctx stream around every launch of the named kernel family) for
 * bench.py's roofline object.  enable=1 starts collecting and clears the totals.  Families (time + launches): "seeds", "compact",
 * "annotate", "sort" (the bucket partition), "replay", "a10" (the filter dedup's passes), "probe", "assemble" (hits -> result rows),
 * "exchange", "db_index", "genome_filter", "seeds_spill", "replay_overflow"; counters only (launches = how often that road was
 * taken; the tests read them): "deferred", "deferred_redo", "a10_part", "a10_redo". */
int sylph_ctx_profile(sylph_ctx *ctx, int enable);
int sylph_ctx_kernel_stats(sylph_ctx *ctx, const char *family, double *total_ms, uint64_t *launches);

/* ---- seeding -------------------------------------------------------------------------------------------- */

/* extract_markers(string, kmer_vec, c, k)  — sketch.rs:53-69 (-> avx2_seeding.rs:33 | seeding.rs:86).
 * FracMinHash seeds of one sequence: mm_hash64 (seeding.rs:4-15) of the canonical k-mer, kept if
 * hash < u64::MAX / c.  Output order: ascending k-mer start position (the reference's lane-interleaved push
 * order is not observable through any caller).  k must be 21 or 31 (avx2_seeding.rs:46-52). */
int sylph_seeds(sylph_ctx *ctx, const uint8_t *bases, uint64_t len, uint32_t c, uint32_t k, int seed_mode,
                uint64_t **out_hashes, uint64_t *out_n);

/* extract_markers_positions(string, kmer_vec, c, k, contig_number) — sketch.rs:71-93, for all contigs of one
 * genome at once: contig i = bases[contig_off[i], contig_off[i+1]).  Returns (contig, pos, hash) triples in
 * ascending (contig, pos) order, i.e. after the reference's `vec.sort()` (sketch.rs:593); pos is the index of
 * the k-mer's LAST base inside its contig (seeding.rs:205, avx2_seeding.rs:253-264). */
int sylph_seeds_positions(sylph_ctx *ctx, const uint8_t *bases, const uint64_t *contig_off, uint64_t n_contigs,
                          uint32_t c, uint32_t k, int seed_mode, uint32_t **out_contig, uint64_t **out_pos,
                          uint64_t **out_hash, uint64_t *out_n);

/* sketch_genome(c, k, ref_file, min_spacing, pseudotax) — sketch.rs:550-622, after FASTA parsing: seeds with
 * positions, genome-wide removal of every k-mer seen >= 2 times (:594-605), greedy min-spacing filter (:602-614).
 * genome_kmers come back in (contig, pos) order; rejected-by-spacing k-mers in out_tracked if pseudotax != 0
 * (out_tracked may be NULL otherwise). */
int sylph_sketch_genome(sylph_ctx *ctx, const uint8_t *bases, const uint64_t *contig_off, uint64_t n_contigs,
                        uint32_t c, uint32_t k, int seed_mode, uint64_t min_spacing, int pseudotax,
                        uint64_t **out_genome_kmers, uint64_t *out_n, uint64_t **out_tracked,
                        uint64_t *out_n_tracked);

/* The same for a BATCH of genomes in one call (a database build: sketch.rs:422-476 loops sketch_genome or
 * sketch_genome_individual :481-548 over files; §8f-3).  The contigs of all genomes are concatenated; genome g owns
 * contigs [genome_contig_off[g], genome_contig_off[g+1]) (for --individual-records: one contig per genome).  Duplicate
 * removal and spacing are per genome, exactly as above, and run on the device.  `mem` says where `bases` lives; the
 * offset arrays are host memory.  Outputs: genome g's genome_kmers = (*out_kmers)[kmer_off[g] .. kmer_off[g+1]) and
 * its tracked k-mers likewise through out_tracked/tracked_off (both NULL, or both non-NULL; filled only when
 * pseudotax != 0).  kmer_off/tracked_off are caller arrays of n_genomes + 1 entries; *out_kmers and *out_tracked are
 * released with sylph_free().  One batch holds at most 2^32-1 bases. */
int sylph_sketch_genomes(sylph_ctx *ctx, const uint8_t *bases, const uint64_t *contig_off, uint64_t n_contigs,
                         const uint64_t *genome_contig_off, uint64_t n_genomes, uint32_t c, uint32_t k, int seed_mode,
                         uint64_t min_spacing, int pseudotax, int mem, uint64_t **out_kmers, uint64_t *kmer_off,
                         uint64_t **out_tracked, uint64_t *tracked_off);

/* ---- read sketching (one session = one sample) --------------------------------------------------------- */

/* Replaces sketch_sequences_needle (sketch.rs:897-959) / sketch_pair_sequences (sketch.rs:771-895) after record
 * parsing.  The host keeps what is sequential or textual: mean_read_length (sketch.rs:941-943, :825-826), file
 * and sample names.  Deduplication is the exact rule of dup_removal_lsh_full_exact (sketch.rs:690-731) with
 * MAX_DEDUP_COUNT = 4 for single-end (constants.rs:14, sketch.rs:937) and no cut-off for pairs (:837); for pairs the session
 * option "dedup_fpr" > 0 selects dup_removal_lsh_full (sketch.rs:733-769), the reference's default (see sylph_sketch_set_option). */
int sylph_sketch_begin(sylph_ctx *ctx, uint32_t c, uint32_t k, int reads_mode, int no_dedup, int seed_mode,
                       sylph_sketch **out);

/* Append a batch of records in file order: record i = bases[rec_off[i], rec_off[i+1]), rec_off[0] == 0.
 * SYLPH_READS_PAIRED: records are interleaved mate1, mate2, mate1, ... (n_records even).  A batch may hold up
 * to 2^32-17 bases.  With SYLPH_MEM_DEVICE the memory must be readable from `bases` rounded down to 16 bytes up to
 * `bases + rec_off[n_records]` rounded up to 16 bytes (true for any pointer into a hipMalloc'ed buffer with 16 B of
 * slack at its end), and complete: the library runs on its own stream unless the ctx was given the producer's. */
int sylph_sketch_push(sylph_sketch *sk, const uint8_t *bases, const uint64_t *rec_off, uint64_t n_records,
                      int mem);
/* Same with the batch's total number of bases (== rec_off[n_records]) supplied by the caller, which saves the library a
 * device->host read of that word when rec_off lives in HBM. */
int sylph_sketch_push_n(sylph_sketch *sk, const uint8_t *bases, const uint64_t *rec_off, uint64_t n_records,
                        uint64_t n_bases, int mem);

/* The same with the encoding of `bases` stated.  SYLPH_ENC_ASCII: one byte per base, any byte value (BYTE_TO_SEQ semantics,
 * types.rs:50-59).  SYLPH_ENC_2BIT: the flat stream packed 4 bases per byte, base i in bits 7-2(i%4)..6-2(i%4) of byte i/4, codes
 * A=0 C=1 G=2 T/U=3 and everything else 0 — exactly what BYTE_TO_SEQ would make of the ASCII, so results are identical; a
 * quarter of the PCIe bytes (the transfer is the long pole of a host-fed sample).  rec_off stays in BASES.  sylph_pack_2bit
 * packs on the host (the feed does it while parsing).  A packed SYLPH_MEM_DEVICE stream must be 4-byte aligned.
 * Host batches (SYLPH_MEM_HOST[_PINNED]) of any size are cut into chunks of whole records inside the call and copied on a
 * second stream while the previous chunk is being sketched. */
#define SYLPH_ENC_ASCII 0
#define SYLPH_ENC_2BIT 1
int sylph_sketch_push_enc(sylph_sketch *sk, const uint8_t *bases, const uint64_t *rec_off, uint64_t n_records, uint64_t n_bases,
                          int mem, int enc);
int sylph_pack_2bit(const uint8_t *ascii, uint64_t n, uint8_t *out);

/* ---- plain four-line FASTQ text, parsed on the device (csrc/fastq.hip) ------------------------------------------------------
 * Replaces the record loop of sketch.rs:775-815 / :897-921 (needletail parse_fastx_file + next()) for uncompressed FASTQ: the TEXT
 * goes to the device (`mem` says where it lies now; host text is copied by the call, SYLPH_MEM_DEVICE text is borrowed until
 * sylph_fastq_destroy and must be readable from 16 bytes below `text` rounded down to 16 up to 16 bytes past its end, like a device
 * batch of sylph_sketch_push), the library finds the records — lines 4r .. 4r+3 = '@' line, sequence, '+' line, quality of the
 * sequence's length; a '\r' in front of a newline is not part of its line; blank space behind the last record is ignored — and
 * sylph_sketch_push_fastq gathers the sequences of records [first, first + n_items) into a batch of the session and sketches it
 * (a paired session takes the two mates' texts: item i = record i of `a`, then record i of `b`; the caller pushes min(n_a, n_b)
 * items, sketch.rs:813-815).  At most 2^32 - 65 bases and 2^31 - 1 records per push.  Returns SYLPH_ERR_FORMAT — and nothing
 * else happens — when the text is not exactly that (multi-line FASTQ, FASTA, a damaged record, no text): the caller then reads the
 * file with its own reader, whose record and error semantics are the reference's.  sylph_fastq_lengths copies the sequence
 * lengths of records [first, first + n) to the host: the reference's running mean of the read lengths (sketch.rs:941-943,
 * :825-826) is sequential f64 arithmetic in file order and stays there.  An index belongs to its context (sessions of the same
 * context push from it; calls are serialised like every call on a context), holds at most 2^32 - 1 lines, and must be destroyed
 * before its context; the gathered batch is the session's own buffer, so "borrow_until_finish" may be set on a session that is
 * fed by ONE sylph_sketch_push_fastq (the seeding verdict then waits for finish). */
typedef struct sylph_fastq sylph_fastq;
int sylph_fastq_index(sylph_ctx *ctx, const void *text, uint64_t n_bytes, int mem, sylph_fastq **out);
int sylph_fastq_counts(const sylph_fastq *f, uint64_t *n_records, uint64_t *n_bases);
int sylph_fastq_lengths(sylph_fastq *f, uint64_t first, uint64_t n, uint32_t *out);
int sylph_sketch_push_fastq(sylph_sketch *sk, sylph_fastq *a, sylph_fastq *b, uint64_t first, uint64_t n_items);
void sylph_fastq_destroy(sylph_fastq *f);

/* ---- gzip inflated on the device (csrc/inflate.hip, round 6) ------------------------------------------------------------------
 * The reference's normal input is gzip (README.md:51): parse_fastx_file (sketch.rs:780-781, :906; :491, :562 for genomes) hands the
 * file to flate2 and inflates it on the calling thread.  sylph_inflate replaces that step: the COMPRESSED bytes of one file (host
 * memory, borrowed for the call; single-member gzip, concatenated members and BGZF alike) go to the device, which finds the deflate
 * blocks, decodes them side by side and checks every member's CRC-32 and ISIZE.  The text stays in HBM: sylph_inflated_text lends
 * the device pointer (readable 256 bytes either side, valid until sylph_inflated_destroy) — hand it to sylph_fastq_index with
 * SYLPH_MEM_DEVICE — and sylph_inflated_read copies a range to the host (FASTA genomes, tests).  Returns SYLPH_ERR_FORMAT — and
 * nothing else happens — when the bytes are not gzip, are damaged, or are a stream this road does not take (3 GiB or more of
 * compressed bytes, a member that begins with a large stored or fixed-Huffman block), SYLPH_ERR_NOMEM when the device has no room: the caller then
 * inflates with its own reader (flate2 / zlib), whose error semantics are the reference's.  The result is byte-identical to zlib's
 * or the call fails; it belongs to its context and must be destroyed before it.  sylph_inflated_info: members, deflate blocks of
 * the stream, header-like bit positions that were decoded to find them, members small enough to have gone through zlib, blocks that
 * were decoded a second time because their first region of memory was too small. */
typedef struct sylph_inflated sylph_inflated;
int sylph_inflate(sylph_ctx *ctx, const void *gz, uint64_t n_bytes, int mem, sylph_inflated **out);
/* Several files in one call — the two mates of a pair (sketch.rs:771-782 opens both readers side by side): their bytes are taken as one
 * stream of gzip members, decoded in one pass (the files share the latency of a deflate block's wavefront instead of paying it one after
 * the other), and every file's text is a piece of the one text: sylph_inflated_file lends file i's pointer and length (each usable with
 * sylph_fastq_index / SYLPH_MEM_DEVICE like sylph_inflated_text's).  All files or none: one that is declined declines the call. */
int sylph_inflate_files(sylph_ctx *ctx, const void *const *gz, const uint64_t *n_bytes, uint32_t n_files, int mem,
                        sylph_inflated **out);
int sylph_inflated_file(const sylph_inflated *t, uint32_t i, const void **dev_text, uint64_t *n_bytes);
int sylph_inflated_text(const sylph_inflated *t, const void **dev_text, uint64_t *n_bytes);
int sylph_inflated_info(const sylph_inflated *t, uint64_t *n_members, uint64_t *n_blocks, uint64_t *n_candidates,
                        uint64_t *n_host_members, uint64_t *n_decoded_again);
int sylph_inflated_read(sylph_inflated *t, uint64_t first, uint64_t n, void *host_out);
void sylph_inflated_destroy(sylph_inflated *t);

/* Finish the sample: (k-mer, count) table in ascending k-mer order == SequencesSketch.kmer_counts
 * (types.rs:145-155) as a keyed multiset, and the number of occurrences removed as duplicates
 * (sketch.rs:886-892).  The *_device variant leaves the table in HBM (owned by the session, valid until
 * sylph_sketch_destroy) so it can be handed to sylph_db_contain without a host round trip. */
int sylph_sketch_finish(sylph_sketch *sk, uint64_t **out_kmers, uint32_t **out_counts, uint64_t *out_n,
                        uint64_t *out_dup_removed);
int sylph_sketch_finish_device(sylph_sketch *sk, const uint64_t **dev_kmers, const uint32_t **dev_counts,
                               uint64_t *out_n, uint64_t *out_dup_removed);
void sylph_sketch_destroy(sylph_sketch *sk);
/* Session options.  "borrow_until_finish" = "1": the caller promises that the SYLPH_MEM_DEVICE batches it pushes stay valid and
 * unchanged until sylph_sketch_finish[_device] has returned (the reference's &[u8] borrow, stretched from the push to the finish).
 * The session's first short-read batch then leaves without a host round trip — the seeding kernel's verdict (a record too long
 * for it, a block of reads that overflowed its slots, the number of seed occurrences) is read together with the finish's own
 * tail block, and a bad verdict re-runs the batch the checked way from the borrowed memory.  Results are identical; one
 * synchronisation per sample instead of two.  sylph_pipeline_submit sets it for its device batches (their memory is borrowed
 * until sylph_pipeline_next has returned the sample anyway).
 * "dedup_fpr" = "<f>" (paired sessions, before the first push; 0 = off, the default): the pair set of sketch_pair_sequences is kept
 * behind a scalable cuckoo filter of false-positive probability f, as the reference does for every --fpr != 0 — its default is 1e-4
 * (cmdline.rs:77; sketch.rs:796-804 builds the filter with initial capacity 10^7, "dedup_capacity" = "<n>" overrides that: tests — a
 * capacity that would fill its buckets beyond 80 % is refused, cuckoo insertions fail there and the answers turn order-dependent).
 * Same walk as sketch.rs:733-769 — test the filter, insert when absent, drop the seed when `*c > 0` — evaluated without walking: a
 * cuckoo filter reports an item iff an item with the same fingerprint and bucket pair went in before it (csrc/a10.hip).  The
 * filter's crate (scalable_cuckoo_filter 0.2.4) is not part of the reference tree: geometry and growth follow its documentation,
 * the hash bits are this library's, so WHICH pairs collide differs from a run of the reference (how many do — about f of the
 * tests once a filter is full — does not).  Bit-exact against the model of the same filter in the tests' CPU checker.
 * Round 5: where ONE filter holds the sample (every sample up to ~1.2 Gbp of pairs at the reference's capacity) the answers come from a
 * sort of the operations by class (two-level partition + in-LDS resolution: no table, no device-wide atomics; 0.25 ms per Gbp
 * instead of 0.92); larger samples, and samples with hundreds of copies of one pair, take round 4's walk over a class table.
 * "a10" = "auto" | "walk" | "part" chooses the pass (A/B and tests; "part" still falls back where one filter does not suffice). */
int sylph_sketch_set_option(sylph_sketch *sk, const char *key, const char *value);

/* ---- containment (sample vs every genome of a resident DB shard) ----------------------------------------- */

/* Load the genome_kmers of a set of GenomeSketch (types.rs:163-173; contain.rs:495 deserialises them) into HBM:
 * genome g = kmers[genome_off[g], genome_off[g+1]).  Builds the k-mer -> genome postings index once (one 64-byte line of
 * 8 {k-mer remainder, genome} slots per bucket of the k-mer space).  At most 2^30-1 genomes; the number of k-mers is bounded
 * by HBM only (the index is built in passes). */
int sylph_db_upload(sylph_ctx *ctx, const uint64_t *kmers, const uint64_t *genome_off, uint64_t n_genomes, int mem,
                    sylph_db **out);
/* A copy of an UNSHARDED database (kept index, tracked index of sylph_db_attach_tracked, genome lengths) on another context —
 * normally another GPU of the node: the arrays travel device to device (hipMemcpyPeer, over xGMI), nothing is uploaded or indexed
 * again.  What N GPUs share where the reference's rayon workers share one Vec<GenomeSketch> (contain.rs:252-295, :482-517).
 * Attach the tracked k-mers to `src` BEFORE replicating if the replicas are to reassign.  Free with sylph_db_destroy. */
int sylph_db_replicate(sylph_db *src, sylph_ctx *dst_ctx, sylph_db **out);
uint64_t sylph_db_n_genomes(const sylph_db *db);
uint64_t sylph_db_n_kmers(const sylph_db *db);

/* Probe half of get_stats (contain.rs:601-656) for every genome of the shard against one sample table
 * (k-mers distinct; entries with count 0 are ignored, :634).  Genomes with fewer than min_number_kmers k-mers
 * report 0 (:627).  contain_count[g] and cov_off[g..g+1] are caller-allocated (n_genomes, n_genomes+1);
 * *out_covs holds, for genome g, covs[cov_off[g] .. cov_off[g+1]) sorted ascending (the reference sorts them
 * before use, contain.rs:661). */
int sylph_db_contain(sylph_db *db, const uint64_t *sample_kmers, const uint32_t *sample_counts, uint64_t n, int mem,
                     double min_number_kmers, uint32_t *contain_count, uint64_t *cov_off, uint32_t **out_covs);
/* Same computation, results left in pinned host memory owned by the database object: no allocation and no second copy.
 * The three arrays stay valid until the next sylph_db_contain* call on this db or sylph_db_destroy (borrow semantics,
 * like the `&GenomeSketch` / `&SequencesSketch` references get_stats works on). *out_n_covs == (*cov_off)[n_genomes]. */
int sylph_db_contain_view(sylph_db *db, const uint64_t *sample_kmers, const uint32_t *sample_counts, uint64_t n, int mem,
                          double min_number_kmers, const uint32_t **contain_count, const uint64_t **cov_off,
                          const uint32_t **covs, uint64_t *out_n_covs);
/* The same again with the coverage values stored as the narrowest of 1, 2 or 4 bytes per value that holds the sample's
 * largest count (*cov_width receives the element size): at GTDB scale the result block shrinks from 7.4 MB to 1.9 MB per
 * sample, and its PCIe transfer is a visible part of the profile stage.  `get_stats` only ever widens these values. */
int sylph_db_contain_view_packed(sylph_db *db, const uint64_t *sample_kmers, const uint32_t *sample_counts, uint64_t n,
                                 int mem, double min_number_kmers, const uint32_t **contain_count,
                                 const uint64_t **cov_off, const void **covs, uint32_t *cov_width,
                                 uint64_t *out_n_covs);
/* Profile reassignment (the `profile` subcommand only).
 * sylph_db_attach_tracked: add the pseudotax_tracked_nonused_kmers of every genome of the shard (types.rs:166; genome g =
 * tracked_kmers[tracked_off[g], tracked_off[g+1])), which take part in the winner table.
 * sylph_db_reassign_view: winner_table (contain.rs:410-430) over the `n_passing` genomes that survived the first pass
 * (passing_gids in the order of the reference's result vector, passing_ani = their first-pass final_est_ani; ties go to
 * the earlier entry, :417) followed by the second probe pass with the winner map (contain.rs:300-307, :637-646).  Same
 * borrowed result views as sylph_db_contain_view (non-passing genomes report 0), plus kmers_lost[g] (:641). */
int sylph_db_attach_tracked(sylph_db *db, const uint64_t *tracked_kmers, const uint64_t *tracked_off, int mem);
int sylph_db_reassign_view(sylph_db *db, const uint64_t *sample_kmers, const uint32_t *sample_counts, uint64_t n, int mem,
                           const uint32_t *passing_gids, const double *passing_ani, uint32_t n_passing,
                           const uint32_t **contain_count, const uint64_t **cov_off, const uint32_t **covs,
                           uint64_t *out_n_covs, const uint32_t **kmers_lost);
void sylph_db_destroy(sylph_db *db);
uint64_t sylph_db_index_bytes(const sylph_db *db);   /* HBM taken by the postings index (lines + overflow runs) */

/* ---- batched containment: S samples per call --------------------------------------------------------------------- */

/* One sample table: n (k-mer, count) entries, k-mers ascending and distinct (what sylph_sketch_finish[_device] returns). */
typedef struct sylph_sample_ref {
    const uint64_t *kmers;
    const uint32_t *counts;
    uint64_t n;
} sylph_sample_ref;

/* The sample-chunk x genome loop of contain.rs:267-289 for a batch of n_samples tables at once: ONE probe launch over the
 * concatenated tables, one sort of (sample, genome, count) hit keys, one device->host copy.  All tables live in `mem`
 * (SYLPH_MEM_HOST or SYLPH_MEM_DEVICE).  Results are borrowed views into pinned memory owned by the db (valid until the next
 * contain call): row r = s * n_genomes + g;  contain_count[r];  coverage values of (s, g) = covs[cov_off[r] .. cov_off[r+1]),
 * ascending, stored as *cov_width (1, 2 or 4) bytes each.  n_samples * n_genomes must stay below 2^32. */
int sylph_db_contain_batch(sylph_db *db, const sylph_sample_ref *samples, uint32_t n_samples, int mem, double min_number_kmers,
                           const uint32_t **contain_count, const uint64_t **cov_off, const void **covs, uint32_t *cov_width,
                           uint64_t *out_n_covs);

/* ---- one database over several GPUs: k-mer-range shards + RCCL ----------------------------------------------------- */

/* SURVEY 8e / north_star: "genome DB resident in HBM and sharded across the 8 GPUs of one node, per-shard containments combined
 * by a single RCCL all-gather".  One process per GPU.  The database is sharded by k-mer RANGE: rank r holds the postings
 * whose k-mer lies in [bounds[r], bounds[r+1]) for ALL genomes, so that — sample tables being sorted — the part of any sample a
 * rank has to probe is a contiguous slice holding 1/world of it, and per-rank work per sample does not grow with `world`.
 * sylph_shard_bounds: equal-width ranges over [0, max_kmer] (hashes are uniform below u64::MAX / c); bounds has world + 1
 * entries.  sylph_db_upload_shard: every rank passes the same genome-major arrays (or at least all k-mers of its range, with
 * the genomes' full offsets) and keeps its range; genome lengths stay those of the whole genomes (contain.rs:627). */
int sylph_shard_bounds(uint64_t max_kmer, uint32_t world, uint64_t *bounds);
int sylph_db_upload_shard(sylph_ctx *ctx, const uint64_t *kmers, const uint64_t *genome_off, uint64_t n_genomes, int mem,
                          const uint64_t *bounds, uint32_t world, uint32_t rank, sylph_db **out);

/* The other way to cut a database (round 5) — north_star's own wording: "sharded by genome", the reference's unit of parallelism
 * (contain.rs:284 runs get_stats per genome on the rayon pool).  Rank r indexes ALL k-mers of the genomes [g_bounds[r], g_bounds[r+1])
 * under their global ids (sylph_genome_shard_bounds: contiguous ranges holding ~1/world of the k-mers each; genome_off is a HOST array
 * there).  sylph_db_contain_batch_sharded and the pipeline take such a database exactly like a k-mer-range one; inside, the table
 * "slices" are whole tables (every rank probes every sample in full — the all-to-all of step 2 delivers what an all-gather of the
 * tables would), the hits travel to their samples' owners as before.  Per-rank probe work grows with `world` here, which is why the
 * k-mer-range cut is the default; both, and plain replicas (sylph_db_replicate), are there to be measured against each other. */
int sylph_genome_shard_bounds(const uint64_t *genome_off, uint64_t n_genomes, uint32_t world, uint64_t *g_bounds);
int sylph_db_upload_genome_shard(sylph_ctx *ctx, const uint64_t *kmers, const uint64_t *genome_off, uint64_t n_genomes, int mem,
                                 const uint64_t *g_bounds, uint32_t world, uint32_t rank, sylph_db **out);

/* Communicator = the collectives the exchange needs, on device buffers, enqueued on `stream` (a hipStream_t).
 * sylph_comm_create_rccl: RCCL (librccl.so resolved at run time: the copy already mapped into the process — e.g. PyTorch's
 * — or the ROCm one); `id` is the 128-byte ncclUniqueId made by sylph_comm_rccl_unique_id on rank 0 and handed to every rank
 * by the host program (MPI, torch.distributed, a file ...).
 * sylph_comm_create: collectives supplied by the caller (tests run the exchange over gloo this way).  Both callbacks return
 * 0 on success.  all_gather: `bytes` from every rank, rank-major, into recv.  all_to_all: send_off / recv_off hold
 * world + 1 byte offsets; [send_off[r], send_off[r+1]) of `send` goes to rank r and lands at recv_off[me] of ITS recv
 * layout, i.e. this rank receives rank r's block into [recv_off[r], recv_off[r+1]). */
typedef struct sylph_comm sylph_comm;
typedef struct sylph_comm_ops {
    int (*all_gather)(void *user, const void *send, void *recv, uint64_t bytes, void *stream);
    int (*all_to_all)(void *user, const void *send, const uint64_t *send_off, void *recv, const uint64_t *recv_off, void *stream);
} sylph_comm_ops;
int sylph_comm_rccl_unique_id(uint8_t id[128]);
int sylph_comm_create_rccl(sylph_ctx *ctx, uint32_t rank, uint32_t world, const uint8_t id[128], sylph_comm **out);
int sylph_comm_create(uint32_t rank, uint32_t world, const sylph_comm_ops *ops, void *user, sylph_comm **out);
void sylph_comm_destroy(sylph_comm *comm);

/* sylph_db_contain_batch for a sharded database: every rank calls it with ITS OWN n_local samples (the ones it sketched;
 * all ranks must call together, n_local may differ per rank, 0 allowed) and gets the complete results of those samples
 * against the WHOLE database, laid out exactly as sylph_db_contain_batch returns them.  Inside, per batch:
 *   1. all-gather of the slice boundaries (a few hundred bytes),
 *   2. all-to-all of the table slices (each rank receives 1/world of every sample of the step),
 *   3. one probe launch over all received slices against the resident shard,
 *   4. the hits are grouped by the rank that owns their sample; all-gather of the group sizes (a few hundred bytes),
 *   5. all-to-all of the hit groups: every rank receives exactly the hits of its own samples, from every shard,
 *   6. each rank sorts its hits and assembles counts + coverage vectors (partial counts of a genome from different shards add
 *      up because a k-mer lives on exactly one shard).
 * Two latency-bound all-gathers of sizes, two all-to-alls of payload: xGMI is point-to-point, an all-to-all puts 1/world of
 * the bytes on every link at once, an all-gather of whole tables or hit lists would deliver world x what anybody needs.
 * Everything stays on the device between the steps; nothing is translated from the reference (which has no such path). */
int sylph_db_contain_batch_sharded(sylph_db *db, sylph_comm *comm, const sylph_sample_ref *samples, uint32_t n_local, int mem,
                                   double min_number_kmers, const uint32_t **contain_count, const uint64_t **cov_off,
                                   const void **covs, uint32_t *cov_width, uint64_t *out_n_covs);


/* Exchange totals of a sharded database since the last reset: batches run, bytes of table slices and of hits this rank SENT to
 * other ranks (the payload of the two all-to-alls; the all-gathers of sizes carry a few KB).  For scaling reports. */
int sylph_db_exchange_stats(sylph_db *db, uint64_t *batches, uint64_t *table_bytes_sent, uint64_t *hit_bytes_sent, int reset);

/* ---- a stream of samples through both stages (sketch -> profile), overlapped inside the library -------------------------- */

/* The reference runs its samples on the rayon pool (sketch.rs:313,371 sketches files on parallel workers; contain.rs:267-289 walks
 * sample chunks x genomes on all threads).  The GPU counterpart: `n_workers` sketch threads, each with a context (stream +
 * memory pool) of its own, and one profile thread on the database's context, so that the small launch-bound kernels of one
 * sample's dedup/count and profile stages run beside the (VALU-bound) seeding kernel of another and the result copies hide
 * behind both.  The caller submits samples and takes their results in submission order; nothing observable depends on how the
 * threads interleave.  A sample is given either as batches of records (what sylph_sketch_push_enc takes; the memory is
 * borrowed until sylph_pipeline_next has returned the sample) or as a session the caller has been pushing into
 * (sylph_pipeline_submit_session: the pipeline finishes it and owns it from then on).
 * The profile thread probes the oldest sketched sample together with every consecutive sample that is ready by then, up to
 * `max_batch` tables per launch.  With a sharded database (`comm` set; every rank runs a pipeline with the same max_batch and
 * submits the same number of samples) a batch is always exactly the next max_batch samples — or, after sylph_pipeline_flush,
 * what is left up to the flush point — so that all ranks enter the exchange of sylph_db_contain_batch_sharded together. */
typedef struct sylph_pipeline sylph_pipeline;
typedef struct sylph_read_batch {   /* the arguments of sylph_sketch_push_enc */
    const uint8_t *bases;
    const uint64_t *rec_off;
    uint64_t n_records, n_bases;
} sylph_read_batch;
typedef struct sylph_pipeline_config {
    uint32_t struct_size;     /* sizeof(sylph_pipeline_config) */
    uint32_t n_workers;       /* sketch threads / contexts; 0 = default (3) */
    uint32_t depth;           /* samples that may be outstanding (submitted, not yet returned by sylph_pipeline_next); 0 = n_workers + 5 */
    uint32_t max_batch;       /* sample tables per probe launch at most (<= 64); 0 = default (8) */
    uint32_t c, k;            /* sylph_sketch_begin's arguments for the samples submitted as batches */
    int reads_mode, no_dedup, seed_mode;
    int want_table;           /* also copy every sample's (k-mer, count) table to host memory (result.kmers / .counts) */
    double min_number_kmers;  /* contain.rs:627 */
    sylph_comm *comm;         /* NULL, or the communicator of a sharded database */
} sylph_pipeline_config;
typedef struct sylph_pipeline_result {
    uint64_t tag;             /* the caller's tag of the sample */
    int status;               /* SYLPH_OK, or the error code of the stage that failed for THIS sample (message in `error`) */
    const char *error;
    uint64_t n_table, dup_removed;                  /* what sylph_sketch_finish reports */
    const uint64_t *dev_kmers; const uint32_t *dev_counts;   /* the table in HBM (e.g. for sylph_db_reassign_view) */
    const uint64_t *kmers; const uint32_t *counts;           /* host copy, when want_table */
    /* containment of the sample against every genome, as sylph_db_contain_view_packed returns it: contain_count[g];
     * coverage values of genome g = covs[cov_off[g] .. cov_off[g + 1]) (ascending, cov_width bytes each); n_covs of them for this
     * sample.  The samples of one probe batch share a result block: `covs` is the block's base, so cov_off[0] is 0 only for the
     * first sample of a batch — always index through cov_off. */
    const uint32_t *contain_count; const uint64_t *cov_off; const void *covs; uint32_t cov_width; uint64_t n_covs;
    uint32_t probe_batch;     /* tables in the probe launch this sample was part of */
    double t_submit, t_sketch_begin, t_sketch_end, t_profile_begin, t_done;   /* CLOCK_MONOTONIC seconds */
} sylph_pipeline_result;
int sylph_pipeline_create(sylph_db *db, const sylph_pipeline_config *cfg, sylph_pipeline **out);
/* ONE sample loop over all GPUs of a node from one process (round 5).  The reference's loop uses the whole machine through its rayon
 * pool (contain.rs:252-295 samples x genomes; sketch.rs:313,371 one file per worker); a database of 113k genomes is a 29 GB index,
 * so on N GPUs the natural split is N replicas: dbs[i] = the copy on GPU i, made by sylph_db_replicate
 * (device to device over xGMI, the database is uploaded and indexed ONCE).  Every replica gets a pipeline of its own with `cfg`
 * (workers, depth, max_batch: per replica; cfg.comm must be NULL); the returned object takes every sylph_pipeline_* call:
 * submit sends a sample to the replica on the device its memory is on (device batches: hipPointerGetAttributes; sessions: their
 * context's device) or, for host memory, to the replica with the fewest samples outstanding; next returns the samples in
 * SUBMISSION order whichever GPU had them (sylph_pipeline_replica_of_last: which one — the device its dev_kmers live on, for
 * sylph_db_reassign_view on dbs[replica]); a result's pointers stay valid until the next sylph_pipeline_next that lands on the
 * same replica.  Two replicas may share a device (that is how the one-GPU tests run it). */
int sylph_pipeline_create_multi(sylph_db *const *dbs, uint32_t n_dbs, const sylph_pipeline_config *cfg, sylph_pipeline **out);
/* Index into dbs[] of the sample the last successful sylph_pipeline_next returned (0 for a single-database pipeline; -1: null). */
int sylph_pipeline_replica_of_last(sylph_pipeline *p);
/* SYLPH_ERR_STATE when `depth` samples are outstanding already (take one with sylph_pipeline_next first). */
int sylph_pipeline_submit(sylph_pipeline *p, const sylph_read_batch *batches, uint32_t n_batches, int mem, int enc, uint64_t tag);
int sylph_pipeline_submit_session(sylph_pipeline *p, sylph_sketch *sk, uint64_t tag);
/* Sharded pipelines only need it: everything submitted so far may be probed in a batch smaller than max_batch. */
int sylph_pipeline_flush(sylph_pipeline *p);
/* Blocks until the OLDEST outstanding sample is done.  Every pointer of *out stays valid until the next sylph_pipeline_next /
 * sylph_pipeline_destroy on this pipeline (the sample's session and its share of the result block are released then). */
int sylph_pipeline_next(sylph_pipeline *p, sylph_pipeline_result *out);
uint32_t sylph_pipeline_outstanding(sylph_pipeline *p);
/* "dedup_fpr" / "dedup_capacity": handed to every session the pipeline opens from then on (sylph_sketch_set_option).
 * The pipeline's own knobs — "serialize_seeding" (default 1: one seeding kernel at a time on the GPU — a worker's stream waits
 * for the event behind the previous worker's seeding kernel, no host blocks — while the other samples are in their dedup/count tails;
 * two VALU-bound seeding kernels side by side only slow each other, +3 % in r04), "min_batch" +
 * "batch_wait_us" (default 1 / 400 — 2 / 400 until round 6, when tables probed one by one became faster and bound the completion intervals: the profile thread waits up to batch_wait_us for min_batch ready tables while more samples are
 * being sketched — never for samples nobody has submitted; unsharded pipelines only; +1.2 % in r04) — else sylph_ctx_set_option on every worker context.  sylph_ctx_profile /
 * sylph_ctx_kernel_stats summed over the workers' and the database's contexts.
 * sylph_pipeline_next on a sharded pipeline returns SYLPH_ERR_STATE instead of blocking when fewer than max_batch samples are
 * outstanding and sylph_pipeline_flush has not covered the oldest one (nobody would ever wake the call). */
int sylph_pipeline_set_option(sylph_pipeline *p, const char *key, const char *value);
int sylph_pipeline_profile(sylph_pipeline *p, int enable);
int sylph_pipeline_kernel_stats(sylph_pipeline *p, const char *family, double *total_ms, uint64_t *launches);
/* Finishes what is outstanding, stops the threads, releases the contexts. */
void sylph_pipeline_destroy(sylph_pipeline *p);

#ifdef __cplusplus
}
#endif
#endif /* SYLPH_HIP_H */
