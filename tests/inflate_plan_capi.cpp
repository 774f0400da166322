// C wrapper around sylph_amd/csrc/inflate_plan.h for tests/test_inflate_plan.py (g++ -lz, no HIP): the very header inflate.hip includes
// — CRC-32 assembled from shifted pieces, gzip member headers, the chain walk — driven by a plain CPU MODEL of what the device
// kernels report (every bit position tested for a dynamic block header with the same acceptance rule as scan1/scan2_kernel; each
// candidate decoded without its window into 16-bit cells the way decode_kernel does, puff-style, one bit at a time; windows
// resolved down the chain).  Test infrastructure: the product never runs this.
#include <zlib.h>

#include <cstring>

#include "../sylph_amd/csrc/inflate_plan.h"

using namespace sylph::inflate_plan;

namespace {

struct BitReader {
    const uint8_t* d;
    uint64_t n_bits, pos;
    bool over = false;
    uint32_t bits(unsigned k) {
        uint32_t v = 0;
        for (unsigned i = 0; i < k; i++) {
            if (pos >= n_bits) { over = true; return 0; }
            v |= (uint32_t)((d[pos >> 3] >> (pos & 7)) & 1) << i;
            pos++;
        }
        return v;
    }
};

struct Code {
    uint16_t count[16] = {0};
    uint16_t sym[320];
    // kind 0: code-length code (must be complete), 1: literal/length, 2: distance (may be empty) — zlib inflate_table's rules
    bool build(const uint8_t* lens, unsigned n, int kind) {
        memset(count, 0, sizeof(count));
        for (unsigned i = 0; i < n; i++) count[lens[i]]++;
        const unsigned codes = n - count[0];
        count[0] = 0;
        int left = 1;
        unsigned maxlen = 0;
        for (unsigned l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return false; if (count[l]) maxlen = l; }
        if (codes == 0) { if (kind != 2) return false; }
        else if (left > 0 && (kind == 0 || maxlen != 1)) return false;
        unsigned off[16];
        off[1] = 0;
        for (unsigned l = 1; l < 15; l++) off[l + 1] = off[l] + count[l];
        for (unsigned i = 0; i < n; i++) if (lens[i]) sym[off[lens[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& b) const {
        int code = 0, first = 0, index = 0;
        for (unsigned l = 1; l <= 15; l++) {
            code |= (int)b.bits(1);
            if (b.over) return -1;
            const int c = count[l];
            if (code - c < first) return sym[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// the header of a dynamic block at b (behind the three type bits) -> both codes; false: zlib would reject it
bool dynamic_header(BitReader& b, Code& lit, Code& dist) {
    const unsigned hlit = b.bits(5) + 257, hdist = b.bits(5) + 1, hclen = b.bits(4) + 4;
    if (b.over || hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (unsigned i = 0; i < hclen; i++) cl[ORDER[i]] = (uint8_t)b.bits(3);
    Code pre;
    if (b.over || !pre.build(cl, 19, 0)) return false;
    uint8_t lens[320];
    unsigned i = 0;
    while (i < hlit + hdist) {
        const int s = pre.decode(b);
        if (s < 0) return false;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        unsigned rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + b.bits(2); }
        else if (s == 17) rep = 3 + b.bits(3);
        else rep = 11 + b.bits(7);
        if (b.over || i + rep > hlit + hdist) return false;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) return false;
    return lit.build(lens, hlit, 1) && dist.build(lens + hlit, hdist, 2);
}

const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// what one wavefront of decode_kernel does for one candidate.  A block that outgrows `cap` is decoded to its end all the same (the
// cells past the region are not kept: flags bit 1) — the caller has it decoded again with room for n_out cells.
BlockResult model_decode(const uint8_t* gz, size_t n, uint64_t start, bool have_window, size_t cap, std::vector<uint16_t>& out) {
    BitReader b{gz, (uint64_t)n * 8, start};
    BlockResult r{0, 0, ST_NONE, 0, 0, 0, {0, 0, 0, 0}};
    bool first = true;
    out.clear();
    size_t count = 0;                                  // cells decoded (== out.size() until the region is full)
    auto put = [&](uint16_t v) { if (count < cap) out.push_back(v); else r.flags |= 2; count++; };
    auto get = [&](size_t i) { return i < out.size() ? out[i] : (uint16_t)0; };
    while (r.status == ST_NONE) {
        const uint64_t at = b.pos;
        const unsigned bfinal = b.bits(1), type = b.bits(2);
        if (b.over) { r.status = ST_ERR_OVERRUN; break; }
        if (!first && type == 2) { r.status = ST_NEXT_DYNAMIC; r.end_bit = at; break; }
        first = false;
        if (type == 3) { r.status = ST_ERR_CODE; break; }
        if (type == 0) {
            b.pos = (b.pos + 7) & ~7ull;
            const unsigned len = b.bits(16), nlen = b.bits(16);
            if (b.over || (len ^ 0xFFFF) != nlen) { r.status = ST_ERR_STORED; break; }
            if (b.pos / 8 + len > n) { r.status = ST_ERR_OVERRUN; break; }
            for (unsigned i = 0; i < len; i++) put(gz[b.pos / 8 + i]);
            b.pos += (uint64_t)len * 8;
        } else {
            Code lit, dist;
            if (type == 2) { if (!dynamic_header(b, lit, dist)) { r.status = b.over ? ST_ERR_OVERRUN : ST_ERR_CODE; break; } }
            else {
                uint8_t l[320];
                for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                for (int i = 0; i < 32; i++) l[288 + i] = 5;
                lit.build(l, 288, 1);
                dist.build(l + 288, 32, 2);
            }
            for (;;) {
                int s = lit.decode(b);
                if (s < 0) { r.status = b.over ? ST_ERR_OVERRUN : ST_ERR_CODE; break; }
                if (s < 256) { put((uint16_t)s); continue; }
                if (s == 256) break;
                s -= 257;
                if (s >= 29) { r.status = ST_ERR_CODE; break; }
                const unsigned len = LBASE[s] + b.bits(LEXT[s]);
                const int ds = dist.decode(b);
                if (ds < 0 || ds >= 30) { r.status = b.over ? ST_ERR_OVERRUN : ST_ERR_CODE; break; }
                const unsigned dd = DBASE[ds] + b.bits(DEXT[ds]);
                if (b.over) { r.status = ST_ERR_OVERRUN; break; }
                if (dd > count) {
                    if (!have_window || dd - count > WINDOW) { r.status = ST_ERR_DISTANCE; break; }
                    r.flags |= 1;
                }
                for (unsigned i = 0; i < len; i++) {
                    const long src = (long)count - (long)dd;
                    put(src >= 0 ? get((size_t)src) : (uint16_t)(256 + WINDOW + src));
                }
            }
            if (r.status != ST_NONE) break;
        }
        if (bfinal) { r.status = ST_FINAL; r.end_bit = b.pos; }
    }
    r.n_out = (uint32_t)count;
    return r;
}

// scan1 + scan2's acceptance rule at one bit position
bool header_like(const uint8_t* gz, size_t n, uint64_t p) {
    BitReader b{gz, (uint64_t)n * 8, p};
    b.bits(1);
    if (b.bits(2) != 2 || b.over) return false;
    Code lit, dist;
    return dynamic_header(b, lit, dist);
}

void put_err(const std::string& e, char* err, size_t errn) {
    if (err && errn) { strncpy(err, e.c_str(), errn - 1); err[errn - 1] = 0; }
}

}  // namespace

extern "C" {

uint64_t ip_member_body(const uint8_t* d, uint64_t n, uint64_t p) { return member_body(d, (size_t)n, (size_t)p); }

// crc_kernel's arithmetic: the text in pieces of `piece` bytes (cut at member ends), each piece's raw register shifted to its place
// in its member, XORed, finished -> out_crc[m]
void ip_crc_members(const uint8_t* text, uint64_t total, uint32_t piece, const uint64_t* member_end, uint32_t n_members, uint32_t* out_crc) {
    uint32_t x2n[64], tab[256];
    crc_x2n_table(x2n);
    crc_byte_table(tab);
    std::vector<uint32_t> raw(n_members, 0);
    uint32_t m = 0;
    for (uint64_t pos = 0; pos < total;) {
        while (member_end[m] <= pos) m++;
        const uint64_t stop = std::min<uint64_t>({total, member_end[m], (pos / piece + 1) * piece});
        raw[m] ^= crc_shift(x2n, crc_raw(tab, 0, text + pos, (size_t)(stop - pos)), member_end[m] - stop);
        pos = stop;
    }
    uint64_t begin = 0;
    for (uint32_t i = 0; i < n_members; i++) { out_crc[i] = crc_finish(x2n, raw[i], member_end[i] - begin); begin = member_end[i]; }
}

uint32_t ip_crc_shift(uint32_t reg, uint64_t n_bytes) {
    uint32_t x2n[64];
    crc_x2n_table(x2n);
    return crc_shift(x2n, reg, n_bytes);
}

// The whole road on the CPU.  -> bytes written to out (<= out_cap), or -1 with the reason in err.  info: members, chain blocks,
// candidates, members inflated by zlib.  ratio / slack: the region rule of decode_kernel (cells per compressed byte up to the next
// candidate + slack).
long long ip_model_inflate(const uint8_t* gz, uint64_t n, uint8_t* out, uint64_t out_cap, uint64_t* info, uint32_t ratio, uint32_t slack, char* err, size_t errn) {
    const size_t body0 = member_body(gz, (size_t)n, 0);
    if (!body0) { put_err("not a gzip file", err, errn); return -1; }
    std::vector<uint64_t> cand;
    cand.push_back((uint64_t)body0 * 8);
    for (uint64_t p = (uint64_t)body0 * 8; p < n * 8; p++)
        if (p != (uint64_t)body0 * 8 && header_like(gz, (size_t)n, p)) cand.push_back(p);
    std::vector<BlockResult> res(cand.size());
    std::vector<std::vector<uint16_t>> cells(cand.size());
    for (size_t k = 0; k < cand.size(); k++) {
        const uint64_t next_byte = k + 1 < cand.size() ? cand[k + 1] >> 3 : n;
        res[k] = model_decode(gz, (size_t)n, cand[k], k != 0, (size_t)((next_byte - (cand[k] >> 3)) * ratio + slack), cells[k]);
    }
    std::vector<std::vector<uint8_t>> host_bytes;
    Chain c = chain_walk(one_segment(gz, (size_t)n), cand, res.data(), [&](size_t p, size_t* end, uint64_t* n_out) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return false;
        std::vector<uint8_t> o(1u << 20);
        zs.next_in = const_cast<Bytef*>(gz + p);
        zs.avail_in = (uInt)std::min<uint64_t>(n - p, 1u << 18);
        zs.next_out = o.data();
        zs.avail_out = (uInt)o.size();
        const bool ok = inflate(&zs, Z_FINISH) == Z_STREAM_END;
        if (ok) { *end = p + zs.total_in; *n_out = zs.total_out; o.resize(zs.total_out); if (!o.empty()) host_bytes.push_back(o); }
        inflateEnd(&zs);
        return ok;
    });
    if (!c.why.empty()) { put_err(c.why, err, errn); return -1; }
    if (c.total > out_cap) { put_err("output buffer too small", err, errn); return -1; }
    uint64_t redone = 0;
    for (const ChainBlock& b : c.blocks) {                 // blocks of the chain that outgrew their region: again, with room for all of it
        if (!(b.flags & 2)) continue;
        const BlockResult again = model_decode(gz, (size_t)n, cand[b.cand], b.cand != 0, b.n_out, cells[b.cand]);
        if (again.status != res[b.cand].status || again.n_out != b.n_out || again.end_bit != res[b.cand].end_bit || (again.flags & 2)) {
            put_err("a block decoded a second time came out differently", err, errn);
            return -1;
        }
        redone++;
    }
    for (size_t i = 0; i < c.host.size(); i++) memcpy(out + c.host[i].out_off, host_bytes[i].data(), host_bytes[i].size());
    std::vector<uint8_t> win(WINDOW, 0);
    for (const ChainBlock& b : c.blocks) {
        const std::vector<uint16_t>& v = cells[b.cand];
        uint8_t* o = out + b.out_off;
        for (size_t i = 0; i < v.size(); i++) o[i] = v[i] < 256 ? (uint8_t)v[i] : win[v[i] - 256];
        // the window in front of the next block: the last WINDOW bytes of (window ++ this block's bytes)
        if (v.size() >= WINDOW) memcpy(win.data(), o + v.size() - WINDOW, WINDOW);
        else { memmove(win.data(), win.data() + v.size(), WINDOW - v.size()); memcpy(win.data() + WINDOW - v.size(), o, v.size()); }
    }
    std::vector<uint64_t> m_end(c.members.size());
    std::vector<uint32_t> crc(c.members.size());
    for (size_t i = 0; i < c.members.size(); i++) m_end[i] = c.members[i].out_end;
    if (!c.members.empty()) ip_crc_members(out, c.total, 1024, m_end.data(), (uint32_t)c.members.size(), crc.data());
    uint64_t on_host = 0;
    for (size_t i = 0; i < c.members.size(); i++) {
        if (c.members[i].on_host) { on_host++; continue; }
        if (crc[i] != c.members[i].crc) { put_err("member " + std::to_string(i) + ": CRC-32 differs", err, errn); return -1; }
    }
    if (info) { info[0] = c.members.size(); info[1] = c.blocks.size(); info[2] = cand.size(); info[3] = on_host; info[4] = redone; }
    return (long long)c.total;
}

}  // extern "C"
