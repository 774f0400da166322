"""Seeded synthetic workloads of SURVEY.md §8d / BASELINE.md (bench.py and the scale tests use these; not part of the product package).

Every random number is ONE word of a splitmix64 stream, addressed by its index (a counter-based generator: word i of stream
`seed` = mix(seed + (i + 1) * 0x9E3779B97F4A7C15), the splitmix64 output function) and everything derived from the words is integer
arithmetic, so a C++ / Rust harness that implements the 6-line mixer regenerates the same genomes, reads and decoy sketches bit
for bit.  The only floating point is in a handful of per-WORKLOAD tables computed on the host with numpy from the same streams
(log-normal abundances of the <= 100 community genomes, decoy genome lengths, long-read lengths): Box-Muller in float64, i.e.
libm's log / cos / exp — identical on the machines this runs on, and at worst an ulp away elsewhere, which moves a table entry with
probability ~1e-10.
Everything large is generated with torch on the device the caller names (GPU for the full-size configs), so a 1 Gbp read set and
a GTDB-R220-scale database appear in HBM without touching the host.  Plumbing only — no sylph logic.

Streams: stream(seed, tag) = mix of (seed, tag) — one per purpose (genome g, picks, inserts, starts, errors, ...).
"""
import math

import numpy as np
import torch

_ACGT = (65, 67, 71, 84)
_GOLDEN = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_MASK = (1 << 64) - 1


def _s64(x):
    """unsigned 64-bit value -> the Python int with the same bits as an int64 (what torch constants need)"""
    x &= _MASK
    return x - (1 << 64) if x >= (1 << 63) else x


def mix_int(z):
    """splitmix64's output function on a Python int"""
    z &= _MASK
    z = ((z ^ (z >> 30)) * _M1) & _MASK
    z = ((z ^ (z >> 27)) * _M2) & _MASK
    return z ^ (z >> 31)


def stream(seed, tag):
    """the seed of the stream `tag` of generator `seed` (both small integers)"""
    return mix_int(mix_int(seed & _MASK) + tag * _GOLDEN)


def _lsr(z, s):
    return (z >> s) & ((1 << (64 - s)) - 1)


def sm64(seed, idx):
    """word idx (int64 tensor, any shape) of stream `seed` as an int64 tensor holding the 64 bits (wrapping arithmetic)"""
    z = (idx + 1) * _s64(_GOLDEN) + _s64(seed)
    z = (z ^ _lsr(z, 30)) * _s64(_M1)
    z = (z ^ _lsr(z, 27)) * _s64(_M2)
    return z ^ _lsr(z, 31)


def sm64_np(seed, idx):
    """the same on the host: idx = numpy integer array -> uint64 array"""
    with np.errstate(over="ignore"):
        z = (np.asarray(idx).astype(np.uint64) + np.uint64(1)) * np.uint64(_GOLDEN) + np.uint64(seed & _MASK)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        return z ^ (z >> np.uint64(31))


def normals_np(seed, n):
    """n standard normals on the host: Box-Muller over words (2i, 2i + 1) of the stream, 53-bit uniforms in (0, 1]"""
    w = sm64_np(seed, np.arange(2 * n, dtype=np.uint64))
    u1 = ((w[0::2] >> np.uint64(11)).astype(np.float64) + 1.0) / float(1 << 53)
    u2 = (w[1::2] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


def _u32(words):
    """the 32 most significant bits of every word: uniform in [0, 2^32)"""
    return _lsr(words, 32)


def _scaled(words, n):
    """uniform integer in [0, n) from the high 32 bits (n <= 2^31): floor(u32 * n / 2^32)"""
    return (_u32(words) * n) >> 32


def random_genomes(n, length, device, seed, mutated_frac=0.10, identity=0.97):
    """n genomes x length bases (uint8 ASCII), single contig each: base i of genome g = the two most significant bits of word i of
    stream(seed, g); the last mutated_frac*n are substitution-only copies (identity) of the first ones (stream(seed, 10^6 + g):
    word i mutates base i when its high 32 bits are below (1 - identity) * 2^32, by adding 1 + (bits 30..31 mod 3))."""
    lut = torch.tensor(_ACGT, dtype=torch.uint8, device=device)
    out = torch.empty((n, length), dtype=torch.uint8, device=device)
    idx = torch.arange(length, dtype=torch.int64, device=device)
    for g in range(n):
        out[g] = _lsr(sm64(stream(seed, g), idx), 62).to(torch.uint8)
    n_mut = int(n * mutated_frac)
    thr = int((1.0 - identity) * (1 << 32))
    for j in range(n_mut):
        src, dst = j, n - n_mut + j
        w = sm64(stream(seed, 1_000_000 + dst), idx)
        mask = _u32(w) < thr
        shift = (1 + (_lsr(w, 30) & 3) % 3).to(torch.uint8)
        out[dst] = torch.where(mask, (out[src] + shift) % 4, out[src])
    return lut[out.long()] if n * length <= (1 << 28) else _codes_to_ascii(out)


def _codes_to_ascii(codes):
    # A=65 C=67 G=71 T=84 from codes 0..3, in uint8 arithmetic (no int64 index temporaries)
    return 65 + 2 * codes + 2 * (codes >= 2).to(torch.uint8) + 11 * (codes == 3).to(torch.uint8)


def abundance_weights(n_gen, seed, sigma=1.0):
    """integer sampling weights (sum ~ 2^32) of a log-normal community: host table, see the module header"""
    ab = np.exp(normals_np(stream(seed, 900_001), n_gen) * sigma)
    w = np.maximum(1, np.floor(ab / ab.sum() * float(1 << 32))).astype(np.int64)
    return w


def paired_reads(genomes, n_pairs, read_len=150, insert_mean=350.0, insert_sd=30.0, abundance_sigma=1.0, err=0.005,
                 dup_frac=0.02, seed=0, chunk=1 << 19):
    """2 x read_len paired reads sampled from a community (log-normal abundances) -> (bases uint8 [n_pairs*2*read_len + 64],
    rec_off int64 [2*n_pairs+1]) with records interleaved mate1, mate2, ... as sylph_sketch_push wants them.
    Mate 1 is the fragment's 5' end, mate 2 the reverse complement of its 3' end; fragments come from either strand.
    Pair p: genome = the bin of (high 32 bits of pick word p) mod sum(weights) in the cumulative weights; insert = mean + ((sum of the
    four 16-bit fields of insert word p - 131070) * round(sd * sqrt(3) * 2^16 / 65536 ...)) — an Irwin-Hall(4) bell of standard
    deviation sd, integers only; start = floor(u32 * (genome length - insert) / 2^32); strand = bit 0 of the start word; base j of
    mate m is replaced when the high 32 bits of error word ((2p + m) * read_len + j) are below err * 2^32, by ACGT[bits 30..31];
    floor(dup_frac * n_pairs) pairs are then overwritten by copies of other pairs of the undisturbed set (distinct destinations)."""
    device = genomes.device
    n_gen, glen = genomes.shape
    cum = torch.from_numpy(np.cumsum(abundance_weights(n_gen, seed, abundance_sigma))).to(device)
    total_w = int(cum[-1].item())
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    lut = torch.tensor(_ACGT, dtype=torch.uint8, device=device)
    out = torch.empty((n_pairs, 2, read_len), dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device)
    flat = genomes.reshape(-1)
    s_pick, s_ins, s_start, s_err = stream(seed, 1), stream(seed, 2), stream(seed, 3), stream(seed, 4)
    sd_scale = int(round(insert_sd * math.sqrt(3.0)))           # Irwin-Hall(4) on [0, 4): variance 1/3
    err_thr = int(err * (1 << 32))
    for s in range(0, n_pairs, chunk):
        m = min(chunk, n_pairs - s)
        p = torch.arange(s, s + m, dtype=torch.int64, device=device)
        gid = torch.searchsorted(cum, _u32(sm64(s_pick, p)) % total_w, right=True).clamp(max=n_gen - 1)
        wi = sm64(s_ins, p)
        ih = (wi & 0xFFFF) + (_lsr(wi, 16) & 0xFFFF) + (_lsr(wi, 32) & 0xFFFF) + _lsr(wi, 48) - 131070
        ins = (int(round(insert_mean)) + ((ih * sd_scale) >> 16)).clamp(read_len, 3 * int(insert_mean))
        ws = sm64(s_start, p)
        start = (_u32(ws) * (glen - ins)) >> 32
        base = gid * glen + start
        fwd = flat[(base[:, None] + ar[None, :])]                                   # fragment 5' end, + strand
        tail = flat[(base + ins - 1)[:, None] - ar[None, :]]                        # fragment 3' end read backwards
        m1, m2 = fwd, comp[tail.long()]
        flip = (ws & 1) == 1                                                        # fragment from the - strand
        m1f, m2f = comp[tail.long()], fwd
        m1 = torch.where(flip[:, None], m1f, m1)
        m2 = torch.where(flip[:, None], m2f, m2)
        pair = torch.stack([m1, m2], dim=1)
        e_idx = (p[:, None, None] * 2 + torch.arange(2, device=device)[None, :, None]) * read_len + ar[None, None, :]
        we = sm64(s_err, e_idx)
        pair = torch.where(_u32(we) < err_thr, lut[(_lsr(we, 30) & 3)], pair)
        out[s:s + m] = pair
    n_dup = int(n_pairs * dup_frac)
    if n_dup:
        i = torch.arange(n_dup, dtype=torch.int64, device=device)
        src = _scaled(sm64(stream(seed, 5), i), n_pairs)
        a = 2_654_435_761 % n_pairs or 1                                           # a multiplier coprime to n_pairs: distinct destinations
        while math.gcd(a, n_pairs) != 1:
            a += 1
        dst = (i * a + 12_345) % n_pairs
        out[dst] = out[src].clone()                                                 # exact PCR-duplicate pairs
    bases = out.reshape(-1)
    pad = torch.zeros(64, dtype=torch.uint8, device=device)                         # 16 B read slack for the ABI
    bases = torch.cat([bases, pad])
    rec_off = torch.arange(0, 2 * n_pairs + 1, device=device, dtype=torch.int64) * read_len
    return bases, rec_off


def ragged_paired_reads(genomes, n_pairs, min_len=35, max_len=151, n_frac=0.001, seed=0, **kw):
    """Pairs as paired_reads() makes them, every mate then trimmed to its own length in [min_len, max_len] (adapter/quality
    trimming; length of record r = min_len + floor(u32 of word r of stream 6 * (max_len - min_len + 1) / 2^32)) and with a
    fraction n_frac of the bases called N (stream 7, word = position in the trimmed stream): the 'honest input' variant of the
    short-read workload (no two lanes of a wavefront walk the same number of k-mers; the exact ASCII->2-bit path is taken).
    -> (bases, rec_off) as above."""
    device = genomes.device
    bases, _ = paired_reads(genomes, n_pairs, read_len=max_len, seed=seed, **kw)
    full = bases[:n_pairs * 2 * max_len].reshape(2 * n_pairs, max_len)
    r = torch.arange(2 * n_pairs, dtype=torch.int64, device=device)
    lens = min_len + _scaled(sm64(stream(seed, 6), r), max_len - min_len + 1)
    keep = torch.arange(max_len, device=device)[None, :] < lens[:, None]
    rec_off = torch.zeros(2 * n_pairs + 1, dtype=torch.int64, device=device)
    rec_off[1:] = torch.cumsum(lens, 0)
    n_thr = int(n_frac * (1 << 32))
    out = []
    step = 1 << 20
    for s in range(0, 2 * n_pairs, step):                                   # boolean compaction in slabs (bounded temporaries)
        blk = full[s:s + step][keep[s:s + step]]
        pos = torch.arange(blk.numel(), dtype=torch.int64, device=device) + rec_off[s]
        nmask = _u32(sm64(stream(seed, 7), pos)) < n_thr
        out.append(torch.where(nmask, torch.full_like(blk, 78), blk))
    out.append(torch.zeros(64, dtype=torch.uint8, device=device))
    return torch.cat(out), rec_off


def decoy_sketches(n_genomes, c=200, device="cuda", seed=0, mean_len=3.3e6, sigma=0.45, kept_frac=0.87,
                   lo=0.5e6, hi=15e6):
    """Sketch-only genomes: n_kept = round(kept_frac * len / c) k-mers below the FracMinHash threshold (valid because mm_hash64
    is a bijection); lengths: host table (log-normal, see the module header); k-mer j of the concatenation = (word j of stream 11
    >> 1) mod threshold.  -> (kmers int64 [N] (bit pattern of u64 < 2^63), genome_off int64 [n+1])"""
    ln = np.clip(np.exp(normals_np(stream(seed, 10), n_genomes) * sigma + math.log(mean_len)), lo, hi)
    n_kept = np.round(kept_frac * ln / c).astype(np.int64)
    off = torch.zeros(n_genomes + 1, dtype=torch.int64, device=device)
    off[1:] = torch.from_numpy(np.cumsum(n_kept)).to(device)
    thr = (2**64 - 1) // c
    total = int(off[-1].item())
    kmers = torch.empty(total, dtype=torch.int64, device=device)
    s_k = stream(seed, 11)
    step = 1 << 26
    for a in range(0, total, step):
        b = min(total, a + step)
        kmers[a:b] = _lsr(sm64(s_k, torch.arange(a, b, dtype=torch.int64, device=device)), 1) % thr
    return kmers, off


def long_reads(genomes, total_bases, n50=10_000, sigma=0.8, err=0.05, abundance_sigma=1.0, seed=0, chunk_bases=1 << 28,
               min_len=500, max_len=200_000, indels=True):
    """ONT-like single-end reads: log-normal lengths with the given N50 (length-weighted median = exp(mu + sigma^2); host table),
    `err` errors per emitted base as SURVEY 8d specifies for C5: substitution : insertion : deletion = 2 : 1 : 1 (indels=False:
    substitutions only, rounds 1-3).  Every emitted base draws one word of stream 23: its high 32 bits < err * 2^32 -> an error, bits
    33:32 (the lowest two of those) pick its kind (0, 1 substitution; 2 insertion: a random base, the source does not advance; 3 deletion: one source base is
    skipped before this one is copied), bits 31:30 the random base.  The source offset of a base is the running sum of the advances
    inside its read; reverse-strand reads walk their source window backwards and are complemented.  Genome / start / strand as in
    paired_reads (streams 21, 22).  -> (bases uint8 [+64 pad], rec_off int64 [n+1])."""
    device = genomes.device
    n_gen, glen = genomes.shape
    mu = math.log(n50) - sigma * sigma
    mean_len = math.exp(mu + sigma * sigma / 2)
    n_guess = int(total_bases / mean_len * 1.2) + 16
    lens_h = np.clip(np.exp(normals_np(stream(seed, 20), n_guess) * sigma + mu).astype(np.int64), min_len, min(max_len, glen - 1))
    n = int(np.searchsorted(np.cumsum(lens_h), total_bases)) + 1
    lens = torch.from_numpy(lens_h[:n]).to(device)
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(lens, 0)
    total = int(off[-1].item())
    cum = torch.from_numpy(np.cumsum(abundance_weights(n_gen, seed, abundance_sigma))).to(device)
    total_w = int(cum[-1].item())
    r = torch.arange(n, dtype=torch.int64, device=device)
    gid = torch.searchsorted(cum, _u32(sm64(stream(seed, 21), r)) % total_w, right=True).clamp(max=n_gen - 1)
    ws = sm64(stream(seed, 22), r)
    # source window of a read: its length plus room for the deletions it may draw (each skips one source base)
    win = torch.clamp(lens + (lens >> 3) + 8, max=glen) if indels else lens
    start = (_u32(ws) * (glen - win)) >> 32
    src0 = gid * glen + start
    flip = (ws & 1) == 1
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    lut = torch.tensor(_ACGT, dtype=torch.uint8, device=device)
    out = torch.empty(total + 64, dtype=torch.uint8, device=device)
    out[total:] = 0
    flat = genomes.reshape(-1)
    err_thr = int(err * (1 << 32))
    s_err = stream(seed, 23)
    r0 = 0
    while r0 < n:                                       # chunks of whole reads
        r1 = int(torch.searchsorted(off, off[r0] + chunk_bases).item())
        r1 = max(r0 + 1, min(r1, n))
        b0, b1 = int(off[r0].item()), int(off[r1].item())
        rel = torch.arange(b1 - b0, device=device)
        rid = torch.repeat_interleave(torch.arange(r0, r1, device=device), lens[r0:r1])
        within = rel - (off[rid] - b0)
        we = sm64(s_err, rel + b0)
        is_err = _u32(we) < err_thr
        if indels:
            kind = _lsr(we, 32) & 3
            ins = is_err & (kind == 2)
            dele = is_err & (kind == 3)
            adv = 1 - ins.long() + dele.long()                    # source bases consumed by this emitted base
            cs = torch.cumsum(adv, 0) - adv                       # exclusive, over the chunk (whole reads)
            within_src = cs - cs[off[rid] - b0] + dele.long()     # offset into the read's source window
            within_src = torch.minimum(within_src, win[rid] - 1)  # (a read that drew more deletions than its window has room for)
            random_base = is_err & (kind != 3)                    # substitutions and insertions emit a random base
        else:
            within_src = within
            random_base = is_err
        fwd_idx = src0[rid] + within_src
        rev_idx = src0[rid] + (win[rid] - 1 - within_src)
        f = flip[rid]
        bases = flat[torch.where(f, rev_idx, fwd_idx)]
        bases = torch.where(f, comp[bases.long()], bases)
        out[b0:b1] = torch.where(random_base, lut[(_lsr(we, 30) & 3)], bases)
        r0 = r1
    return out, off
