// pgunzip.cpp — parallel inflate of an ORDINARY (single-member) gzip file, the common `reads.fastq.gz` (round 4).
//
// A deflate stream is one long chain: every block may copy from the 32 KiB before it, so the reference (needletail -> flate2, one
// thread per file, sketch.rs:906) and rounds 1-3 here (zlib `gzread`) inflate a file on one core: 0.4 Gbp/s, a thousand times
// below what the GPU takes.  Only BGZF members could be inflated side by side (feed.cpp).  This file does what pugz (Kerbiriou &
// Chikhi 2019) showed for FASTQ: cut the COMPRESSED stream into as many stretches as there are threads and
//   1. find, for every stretch but the first, the first bit position at which a deflate block starts: try positions one by one,
//      keep the first that parses as a non-final dynamic block — complete code-length, literal/length and distance codes, an
//      end-of-block symbol, every literal a byte a FASTQ file may hold — and is FOLLOWED by another well-formed block header;
//   2. inflate every stretch from its block start to the next stretch's block start with its own decoder, WITHOUT the 32 KiB that
//      precede it: a copy that reaches back beyond the stretch's own output yields a symbol "byte w of the unknown window"
//      (16-bit cells: 0..255 a byte, 256 + w a window reference), and symbols are copied around like bytes.  This pass keeps only
//      the last 32 KiB of cells, in a ring that lives in the core's cache, and the stretch's length: writing all cells out (the
//      first version: 2 B per inflated byte, 2 GB per file) made 94 decoders that take 14 ms each when alone take 110 ms each —
//      page faults and memory traffic, not decoding (profiles/r04_feed*.txt, tools/gz_trace.sh);
//   3. hand the windows down the chain — the last 32 KiB of stretch i, resolved with the window of stretch i-1, are the window of
//      stretch i+1: 32 KiB of table look-ups per stretch, sequential but tiny — and decode every stretch a SECOND time, now as
//      plain bytes with its window known, straight into its place in the output (the decoder is cheap next to the memory it saves);
//   4. check the member's CRC-32 and length (per-stretch CRCs combined).  ANY doubt on the way — no block start found, a decoder
//      that runs past the next stretch's start instead of landing on it, an invalid code, a CRC mismatch, a second member — makes
//      the whole attempt return false, and the caller reads the file with the sequential reader as before: this path can make a
//      file faster, never different.
// Own decoder (zlib cannot run without its window); zlib only supplies crc32 / crc32_combine.
#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <zlib.h>

#include "sylph_host.hpp"

namespace sylph_host {

namespace {

constexpr uint32_t WINDOW = 32768;
constexpr unsigned PRIMARY = 10;              // bits of the first-level Huffman table

struct Bits {
    const uint8_t* d = nullptr;
    size_t n = 0, bytepos = 0;
    uint64_t acc = 0;
    unsigned cnt = 0;                          // valid bits in acc
    bool overrun = false;
    void init(const uint8_t* data, size_t len, size_t bitpos) {
        d = data; n = len; bytepos = bitpos >> 3; acc = 0; cnt = 0; overrun = false;
        refill();
        drop((unsigned)(bitpos & 7));
    }
    inline void refill() {
        if (bytepos + 8 <= n) {
            uint64_t w;
            memcpy(&w, d + bytepos, 8);
            acc |= w << cnt;
            const unsigned adv = (63 - cnt) >> 3;
            bytepos += adv;
            cnt += adv * 8;
        } else {
            while (cnt <= 56 && bytepos < n) { acc |= (uint64_t)d[bytepos++] << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(unsigned k) const { return (uint32_t)(acc & ((1ull << k) - 1)); }
    inline void drop(unsigned k) {
        if (k > cnt) { overrun = true; acc = 0; cnt = 0; return; }
        acc >>= k;
        cnt -= k;
    }
    inline uint32_t take(unsigned k) { const uint32_t v = peek(k); drop(k); return v; }
    size_t bitpos() const { return bytepos * 8 - cnt; }
};

// two-level canonical Huffman decoder table.  entry = value << 8 | bits; bit 31: `value` is the offset of a second-level table
// and `bits` the number of index bits of it
struct Huff {
    std::vector<uint32_t> t;
    unsigned maxlen = 0;
    static constexpr uint32_t SUB = 1u << 31;
    // lens[0..n): code lengths (0 = unused).  complete: Kraft sum exactly 1 (or, allow_single, exactly one code of length 1).
    bool build(const uint8_t* lens, unsigned n, bool allow_single) {
        unsigned count[16] = {0};
        for (unsigned i = 0; i < n; i++) count[lens[i]]++;
        if (count[0] == n) {                       // no code at all: fine for the distance code of a block of literals only
            if (!allow_single) return false;
            t.assign((size_t)1 << PRIMARY, 0);     // (every pattern decodes to "no code": an error where it is used)
            maxlen = 0;
            return true;
        }
        maxlen = 15;
        while (maxlen > 1 && count[maxlen] == 0) maxlen--;
        long left = 1;
        for (unsigned l = 1; l <= 15; l++) { left <<= 1; left -= (long)count[l]; if (left < 0) return false; }
        if (left > 0 && !(allow_single && n - count[0] == 1 && count[1] == 1)) return false;   // incomplete code
        unsigned next[16];
        unsigned code = 0;
        count[0] = 0;
        for (unsigned l = 1; l <= 15; l++) { code = (code + count[l - 1]) << 1; next[l] = code; }
        const unsigned sub_bits = maxlen > PRIMARY ? maxlen - PRIMARY : 0;
        t.assign((size_t)1 << PRIMARY, 0);
        for (unsigned sym = 0; sym < n; sym++) {
            const unsigned l = lens[sym];
            if (!l) continue;
            unsigned c = next[l]++, rev = 0;
            for (unsigned b = 0; b < l; b++) { rev = (rev << 1) | (c & 1); c >>= 1; }
            if (l <= PRIMARY) {
                for (unsigned k = rev; k < (1u << PRIMARY); k += 1u << l) t[k] = (sym << 8) | l;
            } else {
                const unsigned pre = rev & ((1u << PRIMARY) - 1);
                if (!(t[pre] & SUB)) {
                    const size_t off = t.size();
                    t.resize(off + ((size_t)1 << sub_bits), 0);
                    t[pre] = SUB | ((uint32_t)off << 8) | sub_bits;
                }
                const size_t off = (t[pre] & ~SUB) >> 8;
                for (unsigned k = rev >> PRIMARY; k < (1u << sub_bits); k += 1u << (l - PRIMARY)) t[off + k] = (sym << 8) | l;
            }
        }
        return true;
    }
    // -> symbol, or -1 (a bit pattern no code has: only in an incomplete single-code table)
    inline int decode(Bits& b) const {
        uint32_t e = t[b.peek(PRIMARY)];
        if (e & SUB) e = t[((e & ~SUB) >> 8) + (b.peek(PRIMARY + (e & 0xFF)) >> PRIMARY)];
        const unsigned l = e & 0xFF;
        if (!l) return -1;
        b.drop(l);
        return (int)(e >> 8);
    }
};

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct BlockCodes { Huff lit, dist; };

// header of a dynamic block (after the 3 type bits): both codes, validated the way zlib's inflate validates them
bool read_dynamic(Bits& b, BlockCodes& c) {
    b.refill();
    const unsigned hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (unsigned i = 0; i < hclen; i++) { b.refill(); cl[CL_ORDER[i]] = (uint8_t)b.take(3); }
    if (b.overrun) return false;
    {   // the code-length code must be complete: sum 2^(7 - len) == 2^7 — a few adds before any table is built (the block-start
        // search comes through here for one bit position in nine)
        unsigned kraft = 0;
        for (unsigned i = 0; i < 19; i++) if (cl[i]) kraft += 128u >> cl[i];
        if (kraft != 128) return false;
    }
    Huff clh;
    if (!clh.build(cl, 19, false)) return false;
    uint8_t lens[286 + 30];
    unsigned i = 0;
    while (i < hlit + hdist) {
        b.refill();
        const int s = clh.decode(b);
        if (s < 0 || b.overrun) return false;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        unsigned rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + b.take(2); }
        else if (s == 17) rep = 3 + b.take(3);
        else rep = 11 + b.take(7);
        if (i + rep > hlit + hdist) return false;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (b.overrun || lens[256] == 0) return false;                     // no end-of-block code
    return c.lit.build(lens, hlit, false) && c.dist.build(lens + hlit, hdist, true);
}

const BlockCodes& fixed_codes() {
    static const BlockCodes fc = [] {
        BlockCodes c;
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        c.lit.build(l, 288, false);
        uint8_t dl[32];
        for (int i = 0; i < 32; i++) dl[i] = 5;
        c.dist.build(dl, 32, true);       // (32 codes of 5 bits; 30 and 31 never occur in valid data: rejected where they are decoded)
        return c;
    }();
    return fc;
}

inline bool fastq_byte(unsigned v) { return v == '\n' || v == '\r' || v == '\t' || (v >= 32 && v <= 126); }

// Output of a stretch: 16-bit cells (0..255 bytes, 256 + w references into the unknown window of WINDOW bytes in front of it).
struct Cells {   // 16-bit cells in anonymous memory with 2 MiB pages where the system gives them.  The stretches' cells are regions of
    uint16_t* v = nullptr;        // ONE mapping (no per-thread mmap / mremap / munmap: those take the process's address-space lock
    size_t cap = 0, n = 0;        // exclusively, and dozens of threads growing and freeing 40 MB buffers queued up on it)
    bool own = false;
    Cells() = default;
    Cells(const Cells&) = delete;
    Cells& operator=(const Cells&) = delete;
    ~Cells() { release(); }
    bool reserve(size_t want) {                       // own mapping (the block-start search's scratch)
        if (want <= cap) return true;
        if (v && !own) return false;                  // a carved region does not grow: the attempt is given up (sequential reader)
        want = (want + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        void* nv = v ? mremap(v, cap * 2, want * 2, MREMAP_MAYMOVE) : mmap(nullptr, want * 2, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (nv == MAP_FAILED) return false;
        (void)madvise(nv, want * 2, MADV_HUGEPAGE);
        v = (uint16_t*)nv;
        cap = want;
        own = true;
        return true;
    }
    void carve(uint16_t* base, size_t cells) { v = base; cap = cells; n = 0; own = false; }
    inline bool room(size_t extra) { return n + extra <= cap || reserve(std::max(cap * 3 / 2, n + extra + (1u << 20))); }
    void release() { if (v && own) munmap(v, cap * 2); v = nullptr; cap = 0; }
};

enum class Stop { Error, AtStopBit, FinalBlock, OutOfLimit };

// The three things a stretch is decoded INTO (the decoder below is a template over them):
//   Cells    everything, as 16-bit cells, in memory of its own — the block-start search's scratch (a few MB per attempt);
//   RingOut  pass 1 of a stretch: only the last WINDOW cells are kept, in a ring that stays in the core's cache — what pass 1 is
//            for is the stretch's LENGTH and the symbolic image of its last 32 KiB (the next stretch's window);
//   ByteOut  pass 2: plain bytes, straight into the stretch's place in the output, references in front of the stretch taken from
//            its (by then known) window.
// put(v): a literal;  copy(len, dist, have_window): a match, false = it reaches where nothing can be (error);  n: cells so far.
struct CellsOut {
    Cells& c;
    size_t& n;
    explicit CellsOut(Cells& cells) : c(cells), n(cells.n) {}
    inline bool room(size_t extra) { return c.room(extra); }
    inline void put(unsigned v) { c.v[c.n++] = (uint16_t)v; }
    inline bool copy(unsigned len, unsigned dist, bool have_window) {
        uint16_t* o = c.v + c.n;
        if (dist <= c.n) {
            const uint16_t* src = o - dist;
            if (dist >= len) memcpy(o, src, (size_t)len * 2);
            else for (unsigned i = 0; i < len; i++) o[i] = src[i];      // (overlapping copies run forward, byte order)
        } else {
            if (!have_window || dist - c.n > WINDOW) return false;        // bytes into the unknown window: 1 .. WINDOW
            for (unsigned i = 0; i < len; i++) {
                const long src = (long)c.n + (long)i - (long)dist;
                o[i] = src >= 0 ? c.v[(size_t)src] : (uint16_t)(256 + WINDOW + src);
            }
        }
        c.n += len;
        return true;
    }
};
struct RingOut {
    uint16_t ring[WINDOW];
    size_t n = 0;
    inline bool room(size_t) { return true; }
    inline void put(unsigned v) { ring[n++ & (WINDOW - 1)] = (uint16_t)v; }
    inline bool copy(unsigned len, unsigned dist, bool have_window) {
        if (dist > n && (!have_window || dist - n > WINDOW)) return false;
        const size_t at = n & (WINDOW - 1);
        if (dist <= n && dist >= len && at + len <= WINDOW) {             // the usual match: inside the stretch, no overlap, no wrap of
            const size_t from = (n - dist) & (WINDOW - 1);                // either end in the ring
            if (from + len <= WINDOW) { memmove(ring + at, ring + from, (size_t)len * 2); n += len; return true; }   // (dist near WINDOW: the two may overlap in the ring)
        }
        for (unsigned i = 0; i < len; i++, n++) {                         // (dist <= WINDOW: the source cell is still in the ring)
            const long src = (long)n - (long)dist;
            ring[n & (WINDOW - 1)] = src >= 0 ? ring[(size_t)src & (WINDOW - 1)] : (uint16_t)(256 + WINDOW + src);
        }
        return true;
    }
    // the last min(n, WINDOW) cells in stream order
    void tail(std::vector<uint16_t>& t) const {
        const size_t take = std::min<size_t>(n, WINDOW);
        t.resize(take);
        for (size_t i = 0; i < take; i++) t[i] = ring[(n - take + i) & (WINDOW - 1)];
    }
};
struct ByteOut {
    uint8_t* o;               // the stretch's place in the output
    size_t cap;               // its length (known from pass 1): one byte more is an error
    const uint8_t* win;       // the WINDOW bytes in front of it (nullptr: the stream's true start)
    size_t n = 0;
    inline bool room(size_t extra) { return n + extra <= cap + 258 + 8; }   // (checked exactly where bytes are written)
    inline void put(unsigned v) { if (n < cap) o[n] = (uint8_t)v; n++; }
    inline bool copy(unsigned len, unsigned dist, bool) {
        if (n + len > cap) { n += len; return true; }                     // (longer than pass 1 found it: the caller sees n != cap)
        uint8_t* d = o + n;
        if (dist <= n) {
            const uint8_t* src = d - dist;
            if (dist >= len) memcpy(d, src, len);
            else for (unsigned i = 0; i < len; i++) d[i] = src[i];
        } else {
            if (!win || dist - n > WINDOW) return false;
            for (unsigned i = 0; i < len; i++) {
                const long src = (long)n + (long)i - (long)dist;
                d[i] = src >= 0 ? o[(size_t)src] : win[WINDOW + src];
            }
        }
        n += len;
        return true;
    }
};

// Inflates blocks from b's position.  have_window = false: the stream's true start (a reference before the output is an error).
// stop_bit: stop when a block ends exactly there (SIZE_MAX: run to the final block).  strict: FASTQ bytes only, stop after
// `max_blocks` blocks (the block-start search); limit_cells bounds the output.
template <class Out>
Stop inflate_to(Bits& b, Out& out, bool have_window, size_t stop_bit, bool strict, unsigned max_blocks, size_t limit_cells) {
    BlockCodes dyn;
    for (unsigned blocks = 0;; blocks++) {
        if (b.bitpos() == stop_bit) return Stop::AtStopBit;
        if (b.bitpos() > stop_bit) return Stop::Error;
        if (strict && blocks >= max_blocks) return Stop::OutOfLimit;
        b.refill();
        const unsigned final_block = b.take(1), type = b.take(2);
        if (b.overrun || type == 3) return Stop::Error;
        if (type == 0) {
            b.drop(b.cnt & 7);                                          // to the byte boundary
            b.refill();
            const unsigned len = b.take(16), nlen = b.take(16);
            if (b.overrun || (len ^ 0xFFFF) != nlen) return Stop::Error;
            if (!out.room(len)) return Stop::Error;
            for (unsigned i = 0; i < len; i++) {
                b.refill();
                const unsigned v = b.take(8);
                if (b.overrun || (strict && !fastq_byte(v))) return Stop::Error;
                out.put(v);
            }
        } else {
            const BlockCodes* c = &fixed_codes();
            if (type == 2) { if (!read_dynamic(b, dyn)) return Stop::Error; c = &dyn; }
            for (;;) {
                if (!out.room(258 + 8)) return Stop::Error;
                b.refill();                                             // >= 56 bits: up to three codes of <= 15 bits
                int s = c->lit.decode(b);
                if (s >= 0 && s < 256) {
                    if (strict && !fastq_byte((unsigned)s)) return Stop::Error;
                    out.put((unsigned)s);
                    s = c->lit.decode(b);
                    if (s >= 0 && s < 256) {
                        if (strict && !fastq_byte((unsigned)s)) return Stop::Error;
                        out.put((unsigned)s);
                        s = c->lit.decode(b);
                        if (s >= 0 && s < 256) {
                            if (strict && !fastq_byte((unsigned)s)) return Stop::Error;
                            out.put((unsigned)s);
                            continue;
                        }
                    }
                }
                if (s < 0) return Stop::Error;
                if (s == 256) break;
                s -= 257;
                if (s >= 29) return Stop::Error;
                b.refill();                                             // length extra (<= 5) + distance code (<= 15) + extra (<= 13)
                const unsigned len = LEN_BASE[s] + b.take(LEN_EXTRA[s]);
                const int ds = c->dist.decode(b);
                if (ds < 0 || ds >= 30) return Stop::Error;
                const unsigned dist = DIST_BASE[ds] + b.take(DIST_EXTRA[ds]);
                if (b.overrun) return Stop::Error;
                if (!out.copy(len, dist, have_window)) return Stop::Error;
                if (out.n > limit_cells) return Stop::OutOfLimit;
            }
            if (b.overrun) return Stop::Error;
        }
        if (final_block) return Stop::FinalBlock;
    }
}
inline Stop inflate_cells(Bits& b, Cells& cells, bool have_window, size_t stop_bit, bool strict, unsigned max_blocks, size_t limit_cells) {
    CellsOut out(cells);
    return inflate_to(b, out, have_window, stop_bit, strict, max_blocks, limit_cells);
}

// first bit position >= from (and < to) where a plausible non-final dynamic block starts, followed by another valid block header
size_t find_block_start(const uint8_t* d, size_t n, size_t from_bit, size_t to_bit) {
    Cells scratch;
    if (!scratch.reserve(4u << 20)) return SIZE_MAX;
    Bits b;
    for (size_t bp = from_bit; bp < to_bit; bp++) {
        // BFINAL = 0, BTYPE = 2 (bits, LSB first: 0, 0, 1 -> value 4), HLIT <= 29, HDIST <= 29: a few shifts before any table is built
        const size_t byte = bp >> 3;
        if (byte + 4 > n) return SIZE_MAX;
        uint32_t w;
        memcpy(&w, d + byte, 4);
        w >>= bp & 7;
        if ((w & 7) != 4) continue;
        if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
        b.init(d, n, bp);
        scratch.n = 0;
        // two blocks: the candidate must decode cleanly to its end-of-block AND be followed by a block that does the same
        // (or at least starts validly and stays clean until the cell limit)
        const Stop r = inflate_cells(b, scratch, true, SIZE_MAX, true, 2, 3u << 20);
        if (r == Stop::OutOfLimit || r == Stop::FinalBlock) return bp;
    }
    return SIZE_MAX;
}

// CRC-32 of a buffer: libdeflate's (carry-less multiply, several GB/s) when the system has the library — bound with dlopen, as in
// feed.cpp — else zlib's (slicing tables, ~1 GB/s: a third of this path's time with 64 threads)
uint32_t crc32_of(const uint8_t* p, size_t n) {
    typedef uint32_t (*crc_fn)(uint32_t, const void*, size_t);
    static const crc_fn fast = [] {
        if (getenv("SYLPH_HIP_NO_LIBDEFLATE")) return (crc_fn) nullptr;
        void* lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        return lib ? (crc_fn)dlsym(lib, "libdeflate_crc32") : (crc_fn) nullptr;
    }();
    if (fast) return fast(0, p, n);
    uint32_t x = (uint32_t)crc32(0L, Z_NULL, 0);
    for (size_t q = 0; q < n; q += 1u << 30) x = (uint32_t)crc32(x, p + q, (uInt)std::min<size_t>(n - q, 1u << 30));
    return x;
}

// The decoders' threads live as long as the process: a file is three short parallel phases (tens of ms each), and creating and
// joining 128 threads for each of them — a stack mapping, a malloc arena, their page faults, all under the process's one
// address-space lock, while other threads of the feed are faulting pages in — made a fifth of the decoders start 60 ms late
// (per-stretch times of 10 ms and 70 ms side by side in tools/gz_trace.sh's output; everybody waits for the last).
class DecoderPool {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> workers;              // worker i runs index i + 1 (the caller is index 0)
    const std::function<void(unsigned)>* job = nullptr;
    unsigned want = 0, running = 0;
    unsigned long gen = 0;
    std::atomic<bool> failed{false};
    void loop(unsigned idx) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(unsigned)>* f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (idx >= want) continue;
                f = job;
            }
            try { (*f)(idx); } catch (...) { failed.store(true); }      // (a bad_alloc in a decoder: the attempt is void, never std::terminate)
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--running == 0) cv_done.notify_one();
            }
        }
    }
   public:
    // f(0) .. f(n - 1), f(0) on the calling thread; returns when ALL are done (also when one of them threw: the workers hold a
    // pointer to f) — false if any threw.  One caller at a time (parallel_gunzip's file lock).
    bool run(unsigned n, const std::function<void(unsigned)>& f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            while (workers.size() + 1 < n) {
                const unsigned idx = (unsigned)workers.size() + 1;
                workers.emplace_back([this, idx] { loop(idx); });
                workers.back().detach();                                  // (they end with the process)
            }
            job = &f;
            want = n;
            running = n - 1;
            failed.store(false);
            gen++;
        }
        cv_job.notify_all();
        try { f(0); } catch (...) { failed.store(true); }
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return running == 0; });
        return !failed.load();
    }
};
DecoderPool& decoder_pool() { static DecoderPool* p = new DecoderPool(); return *p; }
template <class F>
bool run_threads(unsigned n, F&& f) {
    if (n <= 1) { try { f(0u); } catch (...) { return false; } return true; }
    const std::function<void(unsigned)> g = [&f](unsigned w) { f(w); };
    return decoder_pool().run(n, g);
}

}  // namespace

bool parallel_gunzip(const uint8_t* gz, size_t n, unsigned threads, uint8_t** out, size_t* out_size, size_t* out_map, size_t memory_budget) {
    *out = nullptr;
    *out_size = 0;
    *out_map = 0;
    if (getenv("SYLPH_HIP_NO_PGUNZIP")) return false;
    // ---- member header (RFC 1952)
    if (n < 18 + 8 || gz[0] != 0x1f || gz[1] != 0x8b || gz[2] != 8 || (gz[3] & 0xE0)) return false;
    size_t p = 10;
    const unsigned flg = gz[3];
    if (flg & 4) { if (p + 2 > n) return false; p += 2 + (gz[p] | (size_t)gz[p + 1] << 8); }
    if (flg & 8) { while (p < n && gz[p]) p++; p++; }
    if (flg & 16) { while (p < n && gz[p]) p++; p++; }
    if (flg & 2) p += 2;
    if (p + 8 >= n) return false;
    const size_t body0 = p, body_end = n - 8;                              // (if this is the only member: checked below)
    const uint8_t* tr = gz + n - 8;
    const uint32_t want_crc = tr[0] | (uint32_t)tr[1] << 8 | (uint32_t)tr[2] << 16 | (uint32_t)tr[3] << 24;
    const uint32_t want_len = tr[4] | (uint32_t)tr[5] << 8 | (uint32_t)tr[6] << 16 | (uint32_t)tr[7] << 24;
    const size_t min_stretch = getenv("SYLPH_HIP_PGZ_STRETCH")   /* (tests lower it; read per call) */ ? (size_t)atol(getenv("SYLPH_HIP_PGZ_STRETCH")) : (1u << 20);
    const size_t body = body_end - body0;
    // One file at a time per process, on all the threads the host can spare: two mate files inflated side by side with 64 threads each
    // took 0.4-0.9 s apiece on the GPU box where one alone takes 0.1-0.2 s (tools/pgunzip_bench.py, profiles/r04_feed.txt) — the
    // decoders are compute-bound and interfere; queueing the files costs nothing and evens the times out.
    static std::mutex one_file_at_a_time;
    std::lock_guard<std::mutex> file_lock(one_file_at_a_time);
    // (as many as the process may really use — effective_cpus() knows about CPU quotas: decoders beyond the quota only burn it in a
    //  burst and leave every thread of the process frozen for the rest of the scheduler's period)
    if (!getenv("SYLPH_HIP_PARSE_THREADS")) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency()), cpus = effective_cpus();
        threads = cpus < hw ? std::max(2u, cpus) : std::max(threads, std::min(128u, hw / 2));
    }
    // T stretches for NT threads, about three per thread: the threads take them off a counter, so that one that starts late or shares
    // its core (the feed of the previous sample is still gathering and pushing meanwhile) holds the others up by a third of its share
    // at most — with one stretch per thread a tenth of the decoders took three times as long as the rest, and everybody waited
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)threads * 3, body / std::max<size_t>(min_stretch, 1024)));
    if (T < 2) return false;                                               // nothing to gain: the sequential reader
    const unsigned NT = std::min(T, threads);
    std::atomic<bool> threw{false};                                        // a decoder thread ran out of memory: the attempt is void
    auto for_stretches = [&](auto&& f) {
        std::atomic<unsigned> next{0};
        if (!run_threads(NT, [&](unsigned) { for (unsigned w = next++; w < T; w = next++) f(w); })) threw = true;
    };
    // memory: the output only (estimate: the trailer's length, or 3x the file if that wrapped)
    const size_t est = std::max<size_t>(want_len, body * 3);
    if (memory_budget && est > memory_budget) return false;
    static const bool trace = getenv("SYLPH_HIP_FEED_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const double t = now();
        fprintf(stderr, "[sylph_hip pgunzip] %-24s %8.3f ms (%u stretches on %u threads)\n", what, (t - t_prev) * 1e3, T, NT);
        t_prev = t;
    };
    // ---- 1. block starts
    std::vector<size_t> start(T + 1, SIZE_MAX);
    start[0] = body0 * 8;
    std::atomic<bool> bad{false};
    for_stretches([&](unsigned w) {
        if (w == 0) return;
        const size_t from = (body0 + body / T * w) * 8, to = std::min((body0 + body / T * (w + 1)) * 8, body_end * 8);
        start[w] = find_block_start(gz, body_end, from, to);
        if (start[w] == SIZE_MAX) bad = true;
    });
    if (bad || threw) return false;
    lap("block starts");
    // Two ways from here, the same bytes either way.  The default: TWO passes, the first keeping nothing but a ring of cells in the
    // core's cache, the second writing plain bytes.  The alternative (SYLPH_HIP_PGZ_PASSES=1; what this file did first): ONE decode into
    // 16-bit cells for the whole stretch, then a translation — one decode less, but 2 B per inflated byte written to memory and read
    // back.  Measured on the GPU box, four 1 Gbp .gz pairs in one command: 2.33 s with two passes, 3.70 s with one under the box's quota
    // of 16 CPUs (decoding into cells runs at 330 MB/s per thread there, into the ring at 780); 2.9 s vs 3.2-3.8 s with 128 threads
    // before the feed knew about the quota (profiles/r04_feed_gz_ab.txt).
    const char* passes_env = getenv("SYLPH_HIP_PGZ_PASSES");
    const bool single_pass = passes_env && atoi(passes_env) == 1;
    std::vector<size_t> off(T + 1, 0);
    std::vector<uint32_t> crc(T, 0);
    size_t total = 0, map_bytes = 0;
    void* buf = nullptr;
    uint8_t* o = nullptr;
    if (single_pass) {
        // ---- 2'. every stretch into cells: regions of one mapping, 12 cells per compressed byte each (address space, not memory:
        // only what is written gets pages; FASTQ deflates 4-6x, a stretch beyond 12x gives the attempt up)
        std::vector<Cells> cells(T);
        std::vector<size_t> region(T + 1, 0);
        for (unsigned w = 0; w < T; w++) {
            const size_t in_bytes = ((w + 1 < T ? start[w + 1] : body_end * 8) - start[w]) / 8 + 16;
            region[w + 1] = region[w] + ((in_bytes * 12 + (1u << 20)) & ~(size_t)((1u << 20) - 1));
        }
        static void* cell_map = nullptr;                                    // (guarded by the file lock above)
        static size_t cell_map_bytes = 0;
        if (cell_map_bytes < region[T] * 2) {
            if (cell_map) munmap(cell_map, cell_map_bytes);
            cell_map_bytes = region[T] * 2 + region[T] / 2;
            cell_map = mmap(nullptr, cell_map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (cell_map == MAP_FAILED) { cell_map = nullptr; cell_map_bytes = 0; return false; }
            (void)madvise(cell_map, cell_map_bytes, MADV_HUGEPAGE);
        }
        for (unsigned w = 0; w < T; w++) cells[w].carve((uint16_t*)cell_map + region[w], region[w + 1] - region[w]);
        std::vector<Stop> how(T, Stop::Error);
        std::vector<size_t> end_bit(T, 0);
        for_stretches([&](unsigned w) {
            Bits b;
            b.init(gz, body_end, start[w]);
            how[w] = inflate_cells(b, cells[w], w != 0, w + 1 < T ? start[w + 1] : SIZE_MAX, false, 0, SIZE_MAX);
            end_bit[w] = b.bitpos();
        });
        if (threw) return false;
        for (unsigned w = 0; w < T; w++)
            if (how[w] != (w + 1 < T ? Stop::AtStopBit : Stop::FinalBlock)) return false;
        if ((end_bit[T - 1] + 7) / 8 != body_end) return false;           // bytes behind the final block: another member, or garbage
        lap("inflate to cells");
        // ---- 3'. windows down the chain, then all stretches translated in parallel into the output
        for (unsigned w = 0; w < T; w++) off[w + 1] = off[w] + cells[w].n;
        total = off[T];
        if ((uint32_t)total != want_len || total < 4) return false;
        buf = inflated_acquire(total, &map_bytes);
        if (!buf) return false;
        o = (uint8_t*)buf;
        std::vector<std::vector<uint8_t>> win(T);
        win[0].assign(WINDOW, 0);
        for (unsigned w = 0; w + 1 < T; w++) {
            std::vector<uint8_t>& nx = win[w + 1];
            nx.assign(WINDOW, 0);
            const Cells& c = cells[w];
            const size_t take = std::min<size_t>(c.n, WINDOW);
            if (take < WINDOW) memcpy(nx.data(), win[w].data() + take, WINDOW - take);   // a short stretch: the rest is the older window
            for (size_t i = 0; i < take; i++) {
                const uint16_t v = c.v[c.n - take + i];
                nx[WINDOW - take + i] = v < 256 ? (uint8_t)v : win[w][v - 256];
            }
        }
        lap("windows");
        for_stretches([&](unsigned w) {
            const Cells& c = cells[w];
            uint8_t* dst = o + off[w];
            const uint8_t* wn = win[w].data();
            for (size_t i = 0; i < c.n; i++) { const uint16_t v = c.v[i]; dst[i] = v < 256 ? (uint8_t)v : wn[v - 256]; }
            crc[w] = crc32_of(dst, c.n);
        });
        if (threw) { inflated_release(buf, map_bytes); return false; }
        lap("translate + crc");
    } else {
    // ---- 2. pass 1: every stretch decoded for its length and its last 32 KiB only (a ring of cells in the core's cache: nothing
    // but the compressed bytes is read from memory, nothing is written to it)
    std::vector<std::vector<uint16_t>> tails(T);
    std::vector<size_t> n_out(T, 0);
    std::vector<Stop> how(T, Stop::Error);
    std::vector<size_t> end_bit(T, 0);
    for_stretches([&](unsigned w) {
        static thread_local std::unique_ptr<RingOut> ring_keep;           // (64 KiB per decoder thread, allocated once)
        if (!ring_keep) ring_keep.reset(new RingOut());
        RingOut* ring = ring_keep.get();
        ring->n = 0;
        Bits b;
        b.init(gz, body_end, start[w]);
        const double t0 = trace ? now() : 0;
        how[w] = inflate_to(b, *ring, w != 0, w + 1 < T ? start[w + 1] : SIZE_MAX, false, 0, SIZE_MAX);
        end_bit[w] = b.bitpos();
        n_out[w] = ring->n;
        ring->tail(tails[w]);
        if (trace) fprintf(stderr, "[sylph_hip pgunzip]   stretch %u: %zu compressed bytes -> %zu bytes, pass 1 in %.2f ms\n", w, (end_bit[w] - start[w]) / 8, n_out[w], (now() - t0) * 1e3);
    });
    if (threw) return false;
    for (unsigned w = 0; w < T; w++)
        if (how[w] != (w + 1 < T ? Stop::AtStopBit : Stop::FinalBlock)) return false;
    if ((end_bit[T - 1] + 7) / 8 != body_end) return false;               // bytes behind the final block: another member, or garbage
    lap("pass 1 (lengths, windows)");
    // ---- 3. windows down the chain, then pass 2: every stretch decoded again, as bytes, straight into its place in the output
    for (unsigned w = 0; w < T; w++) off[w + 1] = off[w] + n_out[w];
    total = off[T];
    if ((uint32_t)total != want_len || total < 4) return false;
    buf = inflated_acquire(total, &map_bytes);                            // (a recycled buffer where there is one: no first touch)
    if (!buf) return false;
    o = (uint8_t*)buf;
    std::vector<std::vector<uint8_t>> win(T);                             // win[w]: the WINDOW bytes in front of stretch w
    win[0].assign(WINDOW, 0);
    for (unsigned w = 0; w + 1 < T; w++) {
        std::vector<uint8_t>& nx = win[w + 1];
        nx.assign(WINDOW, 0);
        const std::vector<uint16_t>& t = tails[w];
        const size_t take = t.size();
        if (take < WINDOW) memcpy(nx.data(), win[w].data() + take, WINDOW - take);   // a short stretch: the rest is the older window
        for (size_t i = 0; i < take; i++) nx[WINDOW - take + i] = t[i] < 256 ? (uint8_t)t[i] : win[w][t[i] - 256];
    }
    lap("windows");
    std::atomic<bool> differs{false};
    for_stretches([&](unsigned w) {
        ByteOut out{o + off[w], n_out[w], w ? win[w].data() : nullptr};
        Bits b;
        b.init(gz, body_end, start[w]);
        const Stop r = inflate_to(b, out, w != 0, w + 1 < T ? start[w + 1] : SIZE_MAX, false, 0, SIZE_MAX);
        if (r != how[w] || b.bitpos() != end_bit[w] || out.n != n_out[w]) { differs = true; return; }   // (cannot happen: the same bits, the same decoder)
        crc[w] = crc32_of(o + off[w], n_out[w]);
    });
    if (differs || threw) { inflated_release(buf, map_bytes); return false; }
    lap("pass 2 (bytes) + crc");
    }
    // ---- 4. the member's CRC-32
    uint32_t all = crc[0];
    for (unsigned w = 1; w < T; w++) all = (uint32_t)crc32_combine(all, crc[w], (z_off_t)(off[w + 1] - off[w]));
    if (all != want_crc) { inflated_release(buf, map_bytes); return false; }
    *out = o;
    *out_size = total;
    *out_map = map_bytes;
    return true;
}

}  // namespace sylph_host
