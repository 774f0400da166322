// hip_ffi.rs — extern "C" declarations for libsylph_hip.so (include/sylph_hip.h).  Shipped uncompiled: there is no Rust
// toolchain in the build image.  See INTEGRATION.md for the call sites in sketch.rs / contain.rs.
use std::os::raw::{c_char, c_int, c_void};
#[repr(C)] pub struct SylphCtx { _p: [u8; 0] }
#[repr(C)] pub struct SylphSketch { _p: [u8; 0] }
#[repr(C)] pub struct SylphDb { _p: [u8; 0] }
#[repr(C)] pub struct SylphComm { _p: [u8; 0] }
#[repr(C)] pub struct SylphFastq { _p: [u8; 0] }
#[repr(C)] pub struct SylphInflated { _p: [u8; 0] }
#[repr(C)] pub struct SylphUpload { _p: [u8; 0] }
#[repr(C)] pub struct SylphSampleRef { pub kmers: *const u64, pub counts: *const u32, pub n: u64 }   // one sorted (k-mer, count) table
#[repr(C)] pub struct SylphCommOps {   // caller-supplied collectives on device buffers (stream = hipStream_t); 0 = success
    pub all_gather: extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes: u64, stream: *mut c_void) -> c_int,
    pub all_to_all: extern "C" fn(user: *mut c_void, send: *const c_void, send_off: *const u64, recv: *mut c_void,
                                  recv_off: *const u64, stream: *mut c_void) -> c_int,
}
#[repr(C)] pub struct SylphPipeline { _p: [u8; 0] }
#[repr(C)] pub struct SylphReadBatch { pub bases: *const u8, pub rec_off: *const u64, pub n_records: u64, pub n_bases: u64 }
#[repr(C)] pub struct SylphPipelineConfig {
    pub struct_size: u32, pub n_workers: u32, pub depth: u32, pub max_batch: u32, pub c: u32, pub k: u32,
    pub reads_mode: c_int, pub no_dedup: c_int, pub seed_mode: c_int, pub want_table: c_int,
    pub min_number_kmers: f64, pub comm: *mut SylphComm,
}
#[repr(C)] pub struct SylphPipelineResult {
    pub tag: u64, pub status: c_int, pub error: *const c_char, pub n_table: u64, pub dup_removed: u64,
    pub dev_kmers: *const u64, pub dev_counts: *const u32, pub kmers: *const u64, pub counts: *const u32,
    pub contain_count: *const u32, pub cov_off: *const u64, pub covs: *const c_void, pub cov_width: u32, pub n_covs: u64,
    pub probe_batch: u32, pub t_submit: f64, pub t_sketch_begin: f64, pub t_sketch_end: f64, pub t_profile_begin: f64, pub t_done: f64,
}
pub const ENC_ASCII: c_int = 0; pub const ENC_2BIT: c_int = 1;
pub const SEED_AVX2_COMPAT: c_int = 1;   // what extract_markers does on every AVX2 host (sketch.rs:53-63)
pub const READS_SINGLE: c_int = 0; pub const READS_PAIRED: c_int = 1;
pub const MEM_HOST: c_int = 0;

extern "C" {
    pub fn sylph_last_error() -> *const c_char;
    pub fn sylph_free(p: *mut c_void);
    pub fn sylph_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut SylphCtx) -> c_int;
    pub fn sylph_ctx_destroy(ctx: *mut SylphCtx);
    // replaces extract_markers (sketch.rs:53) — test seam only, too small to offload per read
    pub fn sylph_seeds(ctx: *mut SylphCtx, bases: *const u8, len: u64, c: u32, k: u32, seed_mode: c_int,
                       out_hashes: *mut *mut u64, out_n: *mut u64) -> c_int;
    // replaces the body of sketch_genome after parsing (sketch.rs:550-622)
    pub fn sylph_sketch_genome(ctx: *mut SylphCtx, bases: *const u8, contig_off: *const u64, n_contigs: u64,
                               c: u32, k: u32, seed_mode: c_int, min_spacing: u64, pseudotax: c_int,
                               out_kmers: *mut *mut u64, out_n: *mut u64,
                               out_tracked: *mut *mut u64, out_n_tracked: *mut u64) -> c_int;
    // replace the record loops of sketch_sequences_needle (sketch.rs:897) / sketch_pair_sequences (sketch.rs:771)
    pub fn sylph_sketch_begin(ctx: *mut SylphCtx, c: u32, k: u32, reads_mode: c_int, no_dedup: c_int,
                              seed_mode: c_int, out: *mut *mut SylphSketch) -> c_int;
    pub fn sylph_sketch_push(sk: *mut SylphSketch, bases: *const u8, rec_off: *const u64, n_records: u64, mem: c_int) -> c_int;
    pub fn sylph_sketch_finish(sk: *mut SylphSketch, out_kmers: *mut *mut u64, out_counts: *mut *mut u32,
                               out_n: *mut u64, out_dup_removed: *mut u64) -> c_int;
    pub fn sylph_sketch_destroy(sk: *mut SylphSketch);
    // staged upload of data that lies in pieces on the host (the genome_kmers of a mapped .syldb): two page-locked chunks owned by the library
    pub fn sylph_upload_begin(ctx: *mut SylphCtx, bytes: u64, chunk_bytes: u64, out: *mut *mut SylphUpload) -> c_int;
    pub fn sylph_upload_chunk(u: *mut SylphUpload, chunk: *mut *mut c_void, cap: *mut u64) -> c_int;
    pub fn sylph_upload_commit(u: *mut SylphUpload, n: u64) -> c_int;
    pub fn sylph_upload_finish(u: *mut SylphUpload, device_ptr: *mut *const c_void) -> c_int;
    pub fn sylph_upload_restart(u: *mut SylphUpload, bytes: u64) -> c_int;   // the same chunks / stream / buffer for the next text
    pub fn sylph_upload_destroy(u: *mut SylphUpload);
    // "borrow_until_finish" = "1": device batches stay valid until finish -> one host round trip per sample instead of two
    // "dedup_fpr" = "<f>" (pairs, before the first push): dup_removal_lsh_full (sketch.rs:733-769) over a cuckoo filter of that
    // false-positive probability instead of the exact set — what sketch_pair_sequences does for every dedup_fpr != 0.
    pub fn sylph_sketch_set_option(sk: *mut SylphSketch, key: *const c_char, value: *const c_char) -> c_int;
    // replace the probe half of get_stats (contain.rs:601-656) for all genomes of a loaded database
    pub fn sylph_db_upload(ctx: *mut SylphCtx, kmers: *const u64, genome_off: *const u64, n_genomes: u64, mem: c_int,
                           out: *mut *mut SylphDb) -> c_int;
    pub fn sylph_db_contain(db: *mut SylphDb, sample_kmers: *const u64, sample_counts: *const u32, n: u64, mem: c_int,
                            min_number_kmers: f64, contain_count: *mut u32, cov_off: *mut u64, out_covs: *mut *mut u32) -> c_int;
    // database build: sketch_genome / sketch_genome_individual for a batch of genomes (sketch.rs:422-476 loop)
    pub fn sylph_sketch_genomes(ctx: *mut SylphCtx, bases: *const u8, contig_off: *const u64, n_contigs: u64,
                                genome_contig_off: *const u64, n_genomes: u64, c: u32, k: u32, seed_mode: c_int,
                                min_spacing: u64, pseudotax: c_int, mem: c_int, out_kmers: *mut *mut u64, kmer_off: *mut u64,
                                out_tracked: *mut *mut u64, tracked_off: *mut u64) -> c_int;
    pub fn sylph_sketch_push_n(sk: *mut SylphSketch, bases: *const u8, rec_off: *const u64, n_records: u64, n_bases: u64,
                               mem: c_int) -> c_int;
    pub fn sylph_sketch_finish_device(sk: *mut SylphSketch, dev_kmers: *mut *const u64, dev_counts: *mut *const u32,
                                      out_n: *mut u64, out_dup_removed: *mut u64) -> c_int;
    // borrowed-result variant of sylph_db_contain (no allocation per sample)
    pub fn sylph_db_contain_view(db: *mut SylphDb, sample_kmers: *const u64, sample_counts: *const u32, n: u64, mem: c_int,
                                 min_number_kmers: f64, contain_count: *mut *const u32, cov_off: *mut *const u64,
                                 covs: *mut *const u32, out_n_covs: *mut u64) -> c_int;
    // the same with coverage values as u8/u16/u32 (*cov_width = element size): 4x less PCIe traffic for typical samples
    pub fn sylph_db_contain_view_packed(db: *mut SylphDb, sample_kmers: *const u64, sample_counts: *const u32, n: u64, mem: c_int,
                                        min_number_kmers: f64, contain_count: *mut *const u32, cov_off: *mut *const u64,
                                        covs: *mut *const c_void, cov_width: *mut u32, out_n_covs: *mut u64) -> c_int;
    // profile: winner_table + second get_stats pass (contain.rs:297-311, 410-430, 637-646)
    pub fn sylph_db_attach_tracked(db: *mut SylphDb, tracked_kmers: *const u64, tracked_off: *const u64, mem: c_int) -> c_int;
    pub fn sylph_db_reassign_view(db: *mut SylphDb, sample_kmers: *const u64, sample_counts: *const u32, n: u64, mem: c_int,
                                  passing_gids: *const u32, passing_ani: *const f64, n_passing: u32,
                                  contain_count: *mut *const u32, cov_off: *mut *const u64, covs: *mut *const u32,
                                  out_n_covs: *mut u64, kmers_lost: *mut *const u32) -> c_int;
    pub fn sylph_ctx_set_option(ctx: *mut SylphCtx, key: *const c_char, value: *const c_char) -> c_int;
    pub fn sylph_ctx_synchronize(ctx: *mut SylphCtx) -> c_int;
    // host feed: page-locked batch buffers for sylph_sketch_push(.., MEM_HOST_PINNED)
    pub fn sylph_pinned_alloc(bytes: u64, out: *mut *mut c_void) -> c_int;
    pub fn sylph_pinned_free(p: *mut c_void);
    // small helpers: library version, sizes of an uploaded database, (contig, end position, hash) triples of a genome
    // (extract_markers_positions, sketch.rs:71-93), per-kernel timing for profiling builds
    pub fn sylph_version() -> c_int;
    pub fn sylph_db_n_genomes(db: *const SylphDb) -> u64;
    pub fn sylph_db_n_kmers(db: *const SylphDb) -> u64;
    pub fn sylph_seeds_positions(ctx: *mut SylphCtx, bases: *const u8, contig_off: *const u64, n_contigs: u64, c: u32, k: u32,
                                 seed_mode: c_int, out_contig: *mut *mut u32, out_pos: *mut *mut u64, out_hash: *mut *mut u64,
                                 out_n: *mut u64) -> c_int;
    pub fn sylph_ctx_profile(ctx: *mut SylphCtx, enable: c_int) -> c_int;
    pub fn sylph_ctx_kernel_stats(ctx: *mut SylphCtx, family: *const c_char, total_ms: *mut f64, launches: *mut u64) -> c_int;
    // ---- round 2: batched containment, packed input, one database over several GPUs ----
    pub fn sylph_db_index_bytes(db: *const SylphDb) -> u64;
    // the sample-chunk x genome loop of contain.rs:267-289 for S samples in one call (row = s * n_genomes + g)
    pub fn sylph_db_contain_batch(db: *mut SylphDb, samples: *const SylphSampleRef, n_samples: u32, mem: c_int,
                                  min_number_kmers: f64, contain_count: *mut *const u32, cov_off: *mut *const u64,
                                  covs: *mut *const c_void, cov_width: *mut u32, out_n_covs: *mut u64) -> c_int;
    // sketch_sequences_needle / sketch_pair_sequences batches with the encoding stated (ENC_2BIT: 4 bases per byte, a quarter
    // of the PCIe bytes); sylph_pack_2bit = BYTE_TO_SEQ (types.rs:50-59) + packing on the host
    pub fn sylph_sketch_push_enc(sk: *mut SylphSketch, bases: *const u8, rec_off: *const u64, n_records: u64, n_bases: u64,
                                 mem: c_int, enc: c_int) -> c_int;
    pub fn sylph_pack_2bit(ascii: *const u8, n: u64, out: *mut u8) -> c_int;
    // plain four-line FASTQ text parsed on the device (needletail's record loop, sketch.rs:775-815 / :897-921, for uncompressed FASTQ):
    // index the text (ERR_FORMAT = -5: not that, keep the needletail loop), read the lengths back for the running mean, push records
    pub fn sylph_fastq_index(ctx: *mut SylphCtx, text: *const c_void, n_bytes: u64, mem: c_int, out: *mut *mut SylphFastq) -> c_int;
    pub fn sylph_fastq_counts(f: *const SylphFastq, n_records: *mut u64, n_bases: *mut u64) -> c_int;
    pub fn sylph_fastq_lengths(f: *mut SylphFastq, first: u64, n: u64, out: *mut u32) -> c_int;
    pub fn sylph_sketch_push_fastq(sk: *mut SylphSketch, a: *mut SylphFastq, b: *mut SylphFastq, first: u64, n_items: u64) -> c_int;
    pub fn sylph_fastq_destroy(f: *mut SylphFastq);
    // round 6: gzip inflated on the device (what flate2 does inside parse_fastx_file, sketch.rs:780-781 / :906): compressed bytes in,
    // text in HBM out (hand the pointer to sylph_fastq_index with MEM_DEVICE); ERR_FORMAT = -5: keep the flate2 reader for this file
    pub fn sylph_inflate(ctx: *mut SylphCtx, gz: *const c_void, n_bytes: u64, mem: c_int, out: *mut *mut SylphInflated) -> c_int;
    pub fn sylph_inflate_files(ctx: *mut SylphCtx, gz: *const *const c_void, n_bytes: *const u64, n_files: u32, mem: c_int,
                               out: *mut *mut SylphInflated) -> c_int;   // the two mates of a pair in one pass
    pub fn sylph_inflated_file(t: *const SylphInflated, i: u32, dev_text: *mut *const c_void, n_bytes: *mut u64) -> c_int;
    pub fn sylph_inflated_text(t: *const SylphInflated, dev_text: *mut *const c_void, n_bytes: *mut u64) -> c_int;
    pub fn sylph_inflated_info(t: *const SylphInflated, n_members: *mut u64, n_blocks: *mut u64, n_candidates: *mut u64,
                               n_host_members: *mut u64, n_decoded_again: *mut u64) -> c_int;
    pub fn sylph_inflated_read(t: *mut SylphInflated, first: u64, n: u64, host_out: *mut c_void) -> c_int;
    pub fn sylph_inflated_destroy(t: *mut SylphInflated);
    // k-mer-range shards: bounds for `world` GPUs, upload of this rank's range, communicator, the collective batch call
    pub fn sylph_shard_bounds(max_kmer: u64, world: u32, bounds: *mut u64) -> c_int;
    pub fn sylph_db_upload_shard(ctx: *mut SylphCtx, kmers: *const u64, genome_off: *const u64, n_genomes: u64, mem: c_int,
                                 bounds: *const u64, world: u32, rank: u32, out: *mut *mut SylphDb) -> c_int;
    // round 5: the cut north_star words — whole genomes per rank (contain.rs:284's unit); same collective call, same pipeline
    pub fn sylph_genome_shard_bounds(genome_off: *const u64, n_genomes: u64, world: u32, g_bounds: *mut u64) -> c_int;
    pub fn sylph_db_upload_genome_shard(ctx: *mut SylphCtx, kmers: *const u64, genome_off: *const u64, n_genomes: u64, mem: c_int,
                                        g_bounds: *const u64, world: u32, rank: u32, out: *mut *mut SylphDb) -> c_int;
    pub fn sylph_comm_rccl_unique_id(id: *mut u8) -> c_int;                       // 128 bytes, made on rank 0
    pub fn sylph_comm_create_rccl(ctx: *mut SylphCtx, rank: u32, world: u32, id: *const u8, out: *mut *mut SylphComm) -> c_int;
    pub fn sylph_comm_create(rank: u32, world: u32, ops: *const SylphCommOps, user: *mut c_void, out: *mut *mut SylphComm) -> c_int;
    pub fn sylph_comm_destroy(comm: *mut SylphComm);
    pub fn sylph_db_contain_batch_sharded(db: *mut SylphDb, comm: *mut SylphComm, samples: *const SylphSampleRef, n_local: u32,
                                          mem: c_int, min_number_kmers: f64, contain_count: *mut *const u32,
                                          cov_off: *mut *const u64, covs: *mut *const c_void, cov_width: *mut u32,
                                          out_n_covs: *mut u64) -> c_int;
    pub fn sylph_db_destroy(db: *mut SylphDb);
    pub fn sylph_db_exchange_stats(db: *mut SylphDb, batches: *mut u64, table_bytes_sent: *mut u64, hit_bytes_sent: *mut u64, reset: c_int) -> c_int;
    // ---- round 3: the sample loop of `sylph profile` (contain.rs:267-289 over sketch.rs:313,371) as a pipeline inside the
    // library: sketch workers + one profile thread; submit samples, take results in submission order
    pub fn sylph_pipeline_create(db: *mut SylphDb, cfg: *const SylphPipelineConfig, out: *mut *mut SylphPipeline) -> c_int;
    // round 5: one sample loop over all GPUs of the node — dbs[i] = the replica on GPU i (sylph_db_replicate: the index copied over
    // xGMI); the returned pipeline takes every sylph_pipeline_* call and hands the samples back in submission order
    pub fn sylph_device_count() -> c_int;
    pub fn sylph_db_replicate(src: *mut SylphDb, dst_ctx: *mut SylphCtx, out: *mut *mut SylphDb) -> c_int;
    pub fn sylph_pipeline_create_multi(dbs: *const *mut SylphDb, n_dbs: u32, cfg: *const SylphPipelineConfig, out: *mut *mut SylphPipeline) -> c_int;
    pub fn sylph_pipeline_replica_of_last(p: *mut SylphPipeline) -> c_int;
    pub fn sylph_pipeline_submit(p: *mut SylphPipeline, batches: *const SylphReadBatch, n_batches: u32, mem: c_int, enc: c_int,
                                 tag: u64) -> c_int;
    pub fn sylph_pipeline_submit_session(p: *mut SylphPipeline, sk: *mut SylphSketch, tag: u64) -> c_int;
    pub fn sylph_pipeline_flush(p: *mut SylphPipeline) -> c_int;
    pub fn sylph_pipeline_next(p: *mut SylphPipeline, out: *mut SylphPipelineResult) -> c_int;
    pub fn sylph_pipeline_outstanding(p: *mut SylphPipeline) -> u32;
    pub fn sylph_pipeline_set_option(p: *mut SylphPipeline, key: *const c_char, value: *const c_char) -> c_int;
    pub fn sylph_pipeline_profile(p: *mut SylphPipeline, enable: c_int) -> c_int;
    pub fn sylph_pipeline_kernel_stats(p: *mut SylphPipeline, family: *const c_char, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn sylph_pipeline_destroy(p: *mut SylphPipeline);
}
