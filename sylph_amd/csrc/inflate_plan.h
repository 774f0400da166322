// inflate_plan.h — the bookkeeping of the device-side gzip inflate (csrc/inflate.hip), host-compilable: nothing here touches HIP,
// so tests/inflate_plan_capi.cpp builds it with g++ and the CPU suite checks it against zlib (tests/test_inflate_plan.py).
//
// What the device cannot do side by side is what makes DEFLATE a chain: a block may copy from the 32 KiB before it, and where a
// block ENDS is known only once it has been decoded.  csrc/inflate.hip turns both into table work:
//   * every bit position of the file is tested for "a dynamic-Huffman block header starts here" (complete code-length code,
//     complete literal/length and distance codes, an end-of-block code) — the CANDIDATES, a superset of the true block starts;
//   * one wavefront decodes each candidate's block without the window (16-bit cells: a byte, or "byte w of the 32 KiB in front
//     of this block") and reports where it ended and how long its output is;
//   * this file walks the chain over those reports (chain_walk): start of the stream -> where that block ended -> the candidate
//     that starts there -> ... -> the final block of the member -> its trailer, the next member's header -> ...; candidates that
//     are not on the chain were never block starts and are dropped with their output;
//   * the windows are resolved down the chain in groups (inflate.hip) and the cells become bytes;
//   * CRC-32 and ISIZE of every member are checked: the CRC of a member is put together from the CRCs of 1 KiB pieces computed
//     anywhere on the device (crc32 is linear over GF(2): crc_shift below).
// Anything that does not add up — a chain that breaks, a block whose output did not fit its region, a CRC that differs — makes
// the whole call decline (SYLPH_ERR_FORMAT) and the caller inflates the file with zlib as before: this road can make a file
// faster, never different.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#define SYLPH_HD __host__ __device__
#else
#define SYLPH_HD
#endif

namespace sylph {
namespace inflate_plan {

constexpr uint32_t WINDOW = 32768;

// ---- what a decoding wavefront reports per candidate ----------------------------------------------------------------------------
enum : uint32_t {
    ST_NONE = 0,
    ST_NEXT_DYNAMIC = 1,   // stopped in front of a dynamic block's header at end_bit (that block is another candidate's)
    ST_FINAL = 2,          // decoded a block with BFINAL set; end_bit = first bit behind its end-of-block code
    ST_ERR_CODE = 16,      // invalid Huffman code / header
    ST_ERR_DISTANCE = 17,  // a copy from further back than a window reaches (or from before the stream's start)
    ST_ERR_OVERRUN = 18,   // ran out of input
    ST_ERR_STORED = 19,    // stored block: LEN != ~NLEN
    ST_OVERFLOW = 20,      // more cells than a block may have here (2^32 - 256)
};
struct BlockResult {
    unsigned long long end_bit;
    uint32_t n_out;        // cells written
    uint32_t status;
    uint32_t flags;        // bit 0: the output holds window references; bit 1: the region was too small — end_bit and n_out are right, the
                           // cells are not all there: the block is decoded once more into a region of n_out cells
    uint32_t pad;
    unsigned long long region;   // where the block's cells begin in the cells buffer (its own region, or the spill region it moved to)
    uint32_t stats[4];     // diagnostics (SYLPH_HIP_INFLATE_STATS=1): symbols decoded, copies served from global memory, shader
                           // kilo-cycles spent on headers + tables, kilo-cycles in all
};

// ---- CRC-32 (reflected, polynomial 0xEDB88320) as polynomial arithmetic over GF(2) ----------------------------------------------
// Bit 31 of a word is the coefficient of x^0 (zlib's convention in crc32.c: multmodp / x2nmodp).
constexpr uint32_t CRC_POLY = 0xEDB88320u;

SYLPH_HD inline uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
// x2n[k] = x^(2^k) mod P, k = 0 .. 63 (filled by crc_x2n_table)
inline void crc_x2n_table(uint32_t t[64]) {
    uint32_t p = 1u << 30;                       // x^1
    t[0] = p;
    for (int k = 1; k < 64; k++) t[k] = p = crc_multmodp(p, p);
}
// x^(8 n) mod P
SYLPH_HD inline uint32_t crc_x8n(const uint32_t* x2n, unsigned long long n_bytes) {
    uint32_t p = 1u << 31;                       // x^0
    unsigned k = 3;
    while (n_bytes) {
        if (n_bytes & 1) p = crc_multmodp(x2n[k & 63], p);
        n_bytes >>= 1;
        k++;
    }
    return p;
}
// The register of the CRC loop (no initial or final inversion) after n more zero bytes
SYLPH_HD inline uint32_t crc_shift(const uint32_t* x2n, uint32_t crc, unsigned long long n_bytes) {
    return n_bytes ? crc_multmodp(crc_x8n(x2n, n_bytes), crc) : crc;
}
// zlib's crc32() of a member from `raw` = XOR over its pieces of crc_shift(raw register of the piece started at 0, bytes behind the
// piece in the member): the initial 0xFFFFFFFF travels through all n bytes, the final inversion is on top
inline uint32_t crc_finish(const uint32_t* x2n, uint32_t raw, unsigned long long n_bytes) {
    return raw ^ crc_shift(x2n, 0xFFFFFFFFu, n_bytes) ^ 0xFFFFFFFFu;
}
// byte-at-a-time table and the raw register update (host reference for the tests; the device builds its slicing tables from this)
inline void crc_byte_table(uint32_t t[256]) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1;
        t[i] = c;
    }
}
inline uint32_t crc_raw(const uint32_t* t, uint32_t reg, const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) reg = t[(reg ^ p[i]) & 0xFF] ^ (reg >> 8);
    return reg;
}

// ---- gzip member header (RFC 1952) ----------------------------------------------------------------------------------------------
// -> offset of the first deflate byte of the member at d[p..), or 0 when there is no well-formed header there
inline size_t member_body(const uint8_t* d, size_t n, size_t p) {
    if (p + 18 > n || d[p] != 0x1f || d[p + 1] != 0x8b || d[p + 2] != 8 || (d[p + 3] & 0xE0)) return 0;
    const unsigned flg = d[p + 3];
    size_t q = p + 10;
    if (flg & 4) { if (q + 2 > n) return 0; q += 2 + (d[q] | (size_t)d[q + 1] << 8); }
    if (flg & 8) { while (q < n && d[q]) q++; q++; }
    if (flg & 16) { while (q < n && d[q]) q++; q++; }
    if (flg & 2) q += 2;
    return q + 8 <= n ? q : 0;
}

// ---- the chain ------------------------------------------------------------------------------------------------------------------
struct ChainBlock {
    uint32_t cand;            // index into the sorted candidates
    uint32_t member;
    uint64_t out_off;         // where its bytes go in the text
    uint32_t n_out;
    uint32_t flags;           // BlockResult::flags
};
struct HostPiece {            // a member that was inflated on the host (its first block is no candidate and it is small)
    size_t gz_begin, gz_end;  // the member's bytes in the file
    uint64_t out_off;
    uint32_t member;
};
struct Member {
    uint64_t out_begin, out_end;
    uint32_t crc, isize;      // from the trailer
    bool on_host;             // inflated and checked by zlib already: no device CRC needed
    uint32_t file;            // which of the call's files it lies in
};
struct Chain {
    std::vector<ChainBlock> blocks;
    std::vector<Member> members;
    std::vector<HostPiece> host;
    uint64_t total = 0;
    std::string why;          // non-empty: the attempt is given up, and this is the reason
};

// The compressed bytes: one file, or several taken as ONE stream of gzip members (the two mates of a pair: a member never crosses from
// one file into the next, so every header, trailer and small member is read inside one segment).
struct Segments {
    std::vector<const uint8_t*> ptr;
    std::vector<size_t> base;                 // base[i] = offset of file i in the stream; base.back() = its total length
    size_t total() const { return base.back(); }
    size_t find(size_t pos) const { return (size_t)(std::upper_bound(base.begin(), base.end() - 1, pos) - base.begin()) - 1; }
};
inline Segments one_segment(const uint8_t* gz, size_t n) { return Segments{{gz}, {0, n}}; }

// cand_bits: sorted, unique candidate start bits (the first member's first block is among them whatever its type);
// res[i]: the report of candidate i.
// host_inflate(begin, &end, &n_out): inflate the member that starts at byte `begin` with the CPU (zlib), report where it ends
// and how many bytes it gave; false = it cannot or will not (too large).  The caller keeps the bytes.
template <class HostInflate>
Chain chain_walk(const Segments& S, const std::vector<uint64_t>& cand_bits, const BlockResult* res, HostInflate&& host_inflate) {
    Chain c;
    const size_t n = S.total();
    size_t p = 0;                                     // the current member starts at byte p
    while (true) {
        const size_t si = S.find(p);
        const uint8_t* gz = S.ptr[si];
        const size_t gb = S.base[si], gn = S.base[si + 1] - gb;
        const size_t body_local = member_body(gz, gn, p - gb);
        if (!body_local) { c.why = "no gzip member header at byte " + std::to_string(p); return c; }
        const size_t body = body_local + gb;
        const uint32_t m = (uint32_t)c.members.size();
        Member mem{c.total, c.total, 0, 0, false, (uint32_t)si};
        uint64_t bit = (uint64_t)body * 8;
        size_t trailer = 0;
        auto it = std::lower_bound(cand_bits.begin(), cand_bits.end(), bit);
        if (it == cand_bits.end() || *it != bit) {
            // the member's first block is not dynamic: a small member (an empty one ends every BGZF file) goes through zlib
            size_t end = 0;
            uint64_t n_out = 0;
            if (!host_inflate(p, &end, &n_out)) { c.why = "member at byte " + std::to_string(p) + " does not start with a dynamic block and is not small"; return c; }
            mem.on_host = true;
            mem.out_end = c.total + n_out;
            if (n_out) c.host.push_back(HostPiece{p, end, c.total, m});
            c.total += n_out;
            c.members.push_back(mem);
            p = end;
        } else {
            while (true) {
                const size_t i = (size_t)(it - cand_bits.begin());
                const BlockResult& r = res[i];
                if (r.status != ST_NEXT_DYNAMIC && r.status != ST_FINAL) {
                    c.why = "block at bit " + std::to_string(bit) + ": decoder status " + std::to_string(r.status);
                    return c;
                }
                c.blocks.push_back(ChainBlock{(uint32_t)i, m, c.total, r.n_out, r.flags});
                c.total += r.n_out;
                if (r.status == ST_FINAL) { trailer = (size_t)((r.end_bit + 7) / 8); break; }
                bit = r.end_bit;
                it = std::lower_bound(it, cand_bits.end(), bit);
                if (it == cand_bits.end() || *it != bit) { c.why = "chain breaks at bit " + std::to_string(bit) + " (no candidate starts there)"; return c; }
            }
            if (trailer + 8 > gb + gn) { c.why = "member trailer beyond the end of the file"; return c; }
            const uint8_t* t = gz + (trailer - gb);
            mem.crc = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
            mem.isize = t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
            mem.out_end = c.total;
            if ((uint32_t)(mem.out_end - mem.out_begin) != mem.isize) { c.why = "member at byte " + std::to_string(p) + ": ISIZE differs from the decoded length"; return c; }
            c.members.push_back(mem);
            p = trailer + 8;
        }
        if (p == n) return c;
        if (p > gb + gn) { c.why = "member runs past the end of the file"; return c; }
        // bytes behind a member that are no further member: zlib's gzread treats them as garbage to ignore only after at least one
        // member — needletail (flate2 MultiGzDecoder) errors; either way not this road's business
        if (p < gb + gn && !member_body(gz, gn, p - gb)) { c.why = "trailing bytes behind the last member"; return c; }
    }
}

}  // namespace inflate_plan
}  // namespace sylph
