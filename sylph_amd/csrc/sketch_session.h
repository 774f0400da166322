// sketch_session.h — state of one read-sketch session (shared by sketch.hip and replay_lds.hip).
#pragma once
#include <algorithm>

#include "common.h"

namespace sylph {
// rid: bit 63 = the record (pair) has dedup markers; bit 62 = approximate dedup only (a10.hip): the filter reported one of the
// occurrence's two (k-mer, markers) items as present; bits 42..61 = paired records with markers only: the occurrence's place in
// its record's EMISSION order (device_common.h emission_rank — the order extract_markers pushed the record's seeds in, which is the
// order sketch.rs:828-867 hands them to the dedup: it decides nothing for the exact set, and for the filter which of two items of
// one pair meets the other); bits 0..41 = global record index (file order; mate = bit 0).
constexpr uint64_t RID_MARKER_BIT = 1ull << 63;
constexpr uint64_t RID_A10_BIT = 1ull << 62;
constexpr int RID_RANK_SHIFT = 42;
constexpr uint64_t RID_RANK_MAX = (1ull << 20) - 1;
constexpr uint64_t RID_MASK = (1ull << RID_RANK_SHIFT) - 1;
constexpr uint64_t INVALID_HASH = ~0ull;
// One surviving seed occurrence.  Array-of-structs (32 B, one sector) because finish() gathers occurrences through a sort
// permutation: four scattered 8 B reads per occurrence cost 4x the HBM sectors of one 32 B read.
struct alignas(32) OccRec { uint64_t hash, rid, m0, m1; };
}  // namespace sylph

namespace sylph {
void flush_pending_slots(sylph_sketch* sk);   // reads.hip
// bits of a hash kept in the 32-bit bucket key of an occurrence: key = hash >> key_shift(c) (hashes are below u64::MAX / c)
inline int key_shift(uint32_t c) { return std::max(0, bit_length(UINT64_MAX / (uint64_t)std::max<uint32_t>(c, 1)) - 32); }
// dedup: DEDUP_EXACT (dup_removal_lsh_full_exact), DEDUP_NONE (--no-dedup), DEDUP_FILTER (dup_removal_lsh_full: the marker test is
// the RID_A10_BIT a10_mark left in the records)
constexpr int DEDUP_EXACT = 0, DEDUP_NONE = 1, DEDUP_FILTER = 2;
void generic_replay(sylph_ctx* ctx, const uint64_t* d_hash, const OccRec* d_recs, uint32_t n_all, bool paired, int dedup,
                    DevBuf& out_k, DevBuf& out_c, uint64_t& n_out, uint64_t& removed_out);   // sketch.hip
void a10_mark(sylph_sketch* sk);              // a10.hip: marks the records (partitioned pass where one filter suffices, else the phase walk)
void a10_settle(sylph_sketch* sk);            // a10.hip: marks if not marked, reads the partitioned pass's verdict (synchronises), redoes with the walk if it was bad
bool a10_verdict(sylph_sketch* sk, const uint32_t words[2]);   // a10.hip: the verdict words read by the caller (finish_bucketed's tail); false = redone with the walk: start over
// A single-end session whose records carry no dedup markers so far (long reads: sketch.rs:922-927 passes no marker above 400
// bases; --no-dedup) keeps only the hashes of its occurrences — nothing the replay looks at besides the hash exists for them
// (no marker, no mate) — and finish() counts them without occurrence records.  The first batch that may carry markers (or a
// finish that needs the records after all) writes the records of what is there: OccRec{hash, 0, 0, 0}.
void materialise_plain_records(sylph_sketch* sk);   // sketch.hip
}

namespace sylph {
// The occurrences of the last short-read batch, still in the per-block slots the seeding kernel wrote them to (reads.hip).
// A sample that arrives in ONE batch — the device-resident case — never has them compacted: finish() partitions straight from
// the slots (replay_lds.hip).  A further batch, a batch with overflowing blocks, or the device-wide finish move them to the
// dense file-order arrays first (flush_pending_slots).
struct PendingSlots {
    bool live = false;
    uint32_t n_blk = 0, slot_cap = 0, n = 0;   // blocks, slots per block, occurrences in the region (deferred: an upper bound)
    // Deferred (round 4): the seeding kernel's verdict — a record too long for it, a block that overflowed its slots, the number of
    // occurrences — has NOT been read back yet.  Only for a session whose caller keeps the batch's memory valid until finish
    // ("borrow_until_finish": the pipeline's device batches, bench.py): finish_bucketed sizes everything from `n_expect` / `n`
    // (estimate / upper bound), takes every count from device memory, and reads the verdict together with its own tail block —
    // one host round trip per sample instead of two.  A bad verdict redoes the batch the ordinary way (redo_deferred_batch).
    bool deferred = false;
    uint32_t n_expect = 0;
    const uint8_t* bases = nullptr;            // the borrowed batch, for the redo
    uint32_t phase = 0;
    const uint64_t* off = nullptr;
    uint64_t n_records = 0, n_bases = 0;
    int enc = 0;
};
void resolve_deferred_slots(sylph_sketch* sk);   // reads.hip: reads the verdict now (block total + flags); may redo the batch
void redo_deferred_batch(sylph_sketch* sk);      // sketch.hip: the deferred batch once more, through the ordinary (checked) push
}  // namespace sylph

struct sylph_sketch {
    sylph_ctx* ctx;
    uint32_t c, k;
    int paired, no_dedup, avx2_compat;
    bool finished = false;
    bool borrow_until_finish = false;   // the caller keeps device batches valid until finish (sylph_sketch_set_option; see PendingSlots)
    double dedup_fpr = 0.;              // > 0 (paired sessions): the reference's default dedup over a cuckoo filter of this false-positive probability (a10.hip)
    uint64_t dedup_capacity = 10000000; // its initial capacity (sketch.rs:800)
    // a10.hip: 0 = the records carry no marks (yet / any more: a batch was pushed or redone), 1 = marked by the partitioned pass, whose
    // verdict words (a10_tail: buckets that overflowed, operations found) nobody has read yet, 2 = marked for good
    int a10_state = 0;
    int a10_force = 0;                  // session/test knob ("a10": 0 auto, 1 always the walk, 2 the partitioned pass wherever one filter suffices)
    bool filter_dedup() const { return paired && !no_dedup && dedup_fpr > 0.; }
    int dedup_mode() const { return no_dedup ? sylph::DEDUP_NONE : (filter_dedup() ? sylph::DEDUP_FILTER : sylph::DEDUP_EXACT); }
    uint64_t rec_base = 0;         // records pushed so far
    uint64_t n_occ = 0;            // occurrences (valid + invalid) appended so far
    uint64_t n_plain = 0;          // the first n_plain of them have no OccRec (marker-less single-end batches, see materialise_plain_records)
    sylph::DevBuf hash;                   // hash of every occurrence, file order (sort key)
    sylph::DevBuf recs;                   // OccRec of every occurrence, file order
    sylph::DevBuf slot_bases[2], slot_off[2];   // device slots of the host-batch pipeline (copy stream fills one, kernels read the other)
    sylph::DevBuf slot_rec, slot_key, slot_meta;   // reads.hip: per-block occurrence slots (OccRec | 32-bit bucket key) and block tables
    sylph::PendingSlots pend;
    sylph::DevBuf batch_ascii;            // a packed batch expanded to ASCII for the position-kernel path
    sylph::DevBuf fq_bases, fq_off;       // fastq.hip: the batch sylph_sketch_push_fastq gathered from FASTQ text (valid until the next such push)
    sylph::DevBuf out_k, out_c;           // final table
    uint64_t n_out = 0, dup_removed = 0;
    sylph::DevBuf counters;               // [0] survivors (u32 @0), [1] n_valid (u64 @8), [2] removed (u64 @16)
    sylph::DevBuf a10_tail;               // a10.hip: verdict words of the partitioned pass (see a10_state)
    explicit sylph_sketch(sylph_ctx* cx)
        : ctx(cx), hash(cx), recs(cx), slot_bases{sylph::DevBuf(cx), sylph::DevBuf(cx)}, slot_off{sylph::DevBuf(cx), sylph::DevBuf(cx)},
          slot_rec(cx), slot_key(cx), slot_meta(cx), batch_ascii(cx), fq_bases(cx), fq_off(cx), out_k(cx), out_c(cx), counters(cx), a10_tail(cx) {}
};

