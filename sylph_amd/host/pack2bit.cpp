// pack2bit.cpp — BYTE_TO_SEQ (types.rs:50-59) + 2-bit packing on the host, as fast as the parsers deliver bases: what the feed
// does before a SYLPH_ENC_2BIT push (a quarter of the PCIe bytes, and less memory traffic than copying the ASCII: 1 B read +
// 1/4 B written per base instead of 1 + 1).  Layout = sylph_sketch_push_enc's: base i of the flat stream in bits
// 7-2(i%4) .. 6-2(i%4) of byte i/4; A/a = 0, C/c = 1, G/g = 2, T/t/U/u = 3, the raw bytes 1, 2, 3 = themselves, everything else 0.
// A Pack2Bit writer appends records to its own part of the stream, starting at ANY base offset: several writers fill one
// buffer in parallel; a byte two writers share (a part that does not begin / end on a multiple of four bases) is stored by
// neither — its two halves are handed back and merged by the caller after the join.
#include <cstring>

#include "sylph_host.hpp"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace sylph_host {

namespace {
struct Lut {
    uint8_t v[256];
    Lut() {
        memset(v, 0, sizeof(v));
        v[1] = 1; v[2] = 2; v[3] = 3;
        const char* s = "AaCcGgTtUu";
        const uint8_t c[10] = {0, 0, 1, 1, 2, 2, 3, 3, 3, 3};
        for (int i = 0; i < 10; i++) v[(uint8_t)s[i]] = c[i];
    }
};
const Lut& lut() { static const Lut l; return l; }

#if defined(__x86_64__)
// 32 bases -> 64 bits, first base most significant; false when some byte is not one of ACGTacgt (the caller then takes the
// table: N, U, raw codes ...)
__attribute__((target("avx2"))) bool pack32_avx2(const uint8_t* p, uint64_t* out, uint32_t valid_mask = 0xFFFFFFFFu) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
    const __m256i three = _mm256_set1_epi8(3);
    const __m256i c = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), three);   // ((b>>1)^(b>>2))&3
    const __m256i letters = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i expect = _mm256_shuffle_epi8(letters, c);
    const __m256i upper = _mm256_and_si256(v, _mm256_set1_epi8((char)0xDF));
    if (((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(upper, expect)) & valid_mask) != valid_mask) return false;
    const __m256i w = _mm256_setr_epi8(64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1);
    const __m256i t = _mm256_madd_epi16(_mm256_maddubs_epi16(c, w), _mm256_set1_epi16(1));   // 8 x i32: the packed byte of bases 4j .. 4j+3
    const __m256i y = _mm256_packus_epi32(t, t);
    const __m256i z = _mm256_packus_epi16(y, y);
    const uint64_t le = (uint32_t)_mm256_extract_epi32(z, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(z, 4) << 32);
    *out = __builtin_bswap64(le);
    return true;
}
bool have_avx2() { static const bool h = __builtin_cpu_supports("avx2"); return h; }
#endif
}  // namespace

Pack2Bit::Pack2Bit(uint8_t* out, uint64_t base_offset) : outp_(out + (base_offset >> 2)), nacc_((unsigned)(base_offset & 3) * 2) {
    skip_first_ = nacc_ != 0;
    first_byte = last_byte = base_offset >> 2;
    base_ = out;
}

void Pack2Bit::emit(uint64_t word, unsigned n_bytes) {   // the n_bytes most significant bytes of `word` go out
    uint8_t b[8];
    const uint64_t be = __builtin_bswap64(word);
    memcpy(b, &be, 8);
    unsigned from = 0;
    if (skip_first_) {                                   // the byte shared with the writer in front: kept aside
        first_partial = true;
        first_val = b[0];
        skip_first_ = false;
        from = 1;
    }
    if (n_bytes > from) memcpy(outp_ + from, b + from, n_bytes - from);
    outp_ += n_bytes;
}

void Pack2Bit::push64(uint64_t v) {
    if (nacc_ == 0) { emit(v, 8); return; }
    emit(acc_ | (v >> nacc_), 8);
    acc_ = v << (64 - nacc_);
}

void Pack2Bit::push_bits(uint64_t v, unsigned nbits) {   // the nbits (even, 2 .. 62) most significant bits of v
    v &= ~0ull << (64 - nbits);
    acc_ |= v >> nacc_;
    if (nacc_ + nbits >= 64) {
        emit(acc_, 8);
        acc_ = nacc_ ? v << (64 - nacc_) : 0;
        nacc_ = nacc_ + nbits - 64;
    } else
        nacc_ += nbits;
}

void Pack2Bit::append(const uint8_t* seq, size_t len, size_t readable) {
    const uint8_t* L = lut().v;
#if defined(__x86_64__)
    if (have_avx2()) {
        while (len >= 32) {
            uint64_t v;
            if (!pack32_avx2(seq, &v)) {
                v = 0;
                for (int i = 0; i < 32; i++) v = (v << 2) | L[seq[i]];
            }
            push64(v);
            seq += 32;
            len -= 32;
            readable = readable > 32 ? readable - 32 : 0;
        }
        // the last 1 .. 31 bases of a record: one more 32-byte load when the bytes behind them are readable (inside a FASTQ
        // file they are the '+' and quality lines), validated and kept only as far as the record goes
        if (len && readable >= 32) {
            uint64_t v;
            if (pack32_avx2(seq, &v, (1u << len) - 1u)) { push_bits(v, (unsigned)len * 2); return; }
        }
    }
#endif
    for (size_t i = 0; i < len; i++) {
        acc_ |= (uint64_t)L[seq[i]] << (62 - nacc_);
        nacc_ += 2;
        if (nacc_ == 64) { emit(acc_, 8); acc_ = 0; nacc_ = 0; }
    }
}

void Pack2Bit::finish() {
    const unsigned whole = nacc_ / 8, rest = nacc_ % 8;
    if (skip_first_ && whole == 0) {                     // the whole part lies inside ONE byte shared on both sides
        first_partial = true;
        first_val = (uint8_t)(acc_ >> 56);
        skip_first_ = false;
        last_byte = first_byte;
        return;
    }
    if (whole) emit(acc_, whole);
    if (rest) {                                          // the byte shared with the writer behind
        last_partial = true;
        last_val = (uint8_t)((acc_ << (8 * whole)) >> 56);
        last_byte = (uint64_t)(outp_ - base_);
    }
}

void merge_pack_edges(uint8_t* out, const Pack2Bit* w, size_t n) {
    for (size_t i = 0; i < n; i++) {                     // (the buffer is reused: whatever the shared bytes held goes first)
        if (w[i].first_partial) out[w[i].first_byte] = 0;
        if (w[i].last_partial) out[w[i].last_byte] = 0;
    }
    for (size_t i = 0; i < n; i++) {
        if (w[i].first_partial) out[w[i].first_byte] |= w[i].first_val;
        if (w[i].last_partial) out[w[i].last_byte] |= w[i].last_val;
    }
}

}  // namespace sylph_host

extern "C" {
// test entry: packs `n_rec` records (concatenated in `bases`, offsets `off`) with `parts` writers over equal shares of the
// records, merges the shared bytes, returns the packed stream in out ((total + 3) / 4 bytes); buf_len = readable bytes of `bases`
int sylph_host_pack_records(const uint8_t* bases, const uint64_t* off, uint64_t n_rec, uint32_t parts, uint8_t* out, uint64_t buf_len) {
    using sylph_host::Pack2Bit;
    if (parts == 0) parts = 1;
    std::vector<Pack2Bit> w;
    for (uint32_t p = 0; p < parts; p++) {
        const uint64_t r0 = n_rec * p / parts, r1 = n_rec * (p + 1) / parts;
        w.emplace_back(out, off[r0]);
        for (uint64_t r = r0; r < r1; r++) w.back().append(bases + off[r], (size_t)(off[r + 1] - off[r]), (size_t)(buf_len - off[r]));
        w.back().finish();
    }
    sylph_host::merge_pack_edges(out, w.data(), w.size());
    return 0;
}
}
