"""Seeded synthetic workloads of SURVEY.md §8d / BASELINE.md (bench.py and the scale tests use these; not part of the product package).

Everything is generated with torch on the device the caller names (GPU for the full-size configs), so a 1 Gbp
read set and a GTDB-R220-scale database appear in HBM without touching the host.  Plumbing only — no sylph logic.
"""
import math

import torch

_ACGT = (65, 67, 71, 84)


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def random_genomes(n, length, device, seed, mutated_frac=0.10, identity=0.97):
    """n genomes x length bases (uint8 ASCII), single contig each; the last mutated_frac*n are substitution-only
    copies (identity) of the first ones."""
    g = _gen(device, seed)
    codes = torch.randint(0, 4, (n, length), generator=g, device=device, dtype=torch.uint8)
    n_mut = int(n * mutated_frac)
    for j in range(n_mut):
        src, dst = j, n - n_mut + j
        mask = torch.rand(length, generator=g, device=device) < (1.0 - identity)
        shift = torch.randint(1, 4, (length,), generator=g, device=device, dtype=torch.uint8)
        codes[dst] = torch.where(mask, (codes[src] + shift) % 4, codes[src])
    # A=65 C=67 G=71 T=84 from codes 0..3, in uint8 arithmetic (no int64 index temporaries)
    return 65 + 2 * codes + 2 * (codes >= 2).to(torch.uint8) + 11 * (codes == 3).to(torch.uint8)


def paired_reads(genomes, n_pairs, read_len=150, insert_mean=350.0, insert_sd=30.0, abundance_sigma=1.0, err=0.005,
                 dup_frac=0.02, seed=0, chunk=1 << 19):
    """2 x read_len paired reads sampled from a community (log-normal abundances) -> (bases uint8 [n_pairs*2*read_len],
    rec_off int64 [2*n_pairs+1]) with records interleaved mate1, mate2, ... as sylph_sketch_push wants them.
    Mate 1 is the fragment's 5' end, mate 2 the reverse complement of its 3' end; fragments come from either strand."""
    device = genomes.device
    g = _gen(device, seed)
    n_gen, glen = genomes.shape
    ab = torch.exp(torch.randn(n_gen, generator=g, device=device) * abundance_sigma)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    lut = torch.tensor(_ACGT, dtype=torch.uint8, device=device)
    out = torch.empty((n_pairs, 2, read_len), dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device)
    flat = genomes.reshape(-1)
    for s in range(0, n_pairs, chunk):
        m = min(chunk, n_pairs - s)
        gid = torch.multinomial(ab, m, replacement=True, generator=g)
        ins = (torch.randn(m, generator=g, device=device) * insert_sd + insert_mean).round().long().clamp(read_len, 3 * int(insert_mean))
        start = (torch.rand(m, generator=g, device=device) * (glen - ins).float()).long().clamp(min=0)
        base = gid * glen + start
        fwd = flat[(base[:, None] + ar[None, :])]                                   # fragment 5' end, + strand
        tail = flat[(base + ins - 1)[:, None] - ar[None, :]]                        # fragment 3' end read backwards
        m1, m2 = fwd, comp[tail.long()]
        flip = torch.rand(m, generator=g, device=device) < 0.5                      # fragment from the - strand
        m1f, m2f = comp[tail.long()], fwd
        m1 = torch.where(flip[:, None], m1f, m1)
        m2 = torch.where(flip[:, None], m2f, m2)
        pair = torch.stack([m1, m2], dim=1)
        e = torch.rand(pair.shape, generator=g, device=device) < err
        rnd = lut[torch.randint(0, 4, pair.shape, generator=g, device=device)]
        out[s:s + m] = torch.where(e, rnd, pair)
    n_dup = int(n_pairs * dup_frac)
    if n_dup:
        src = torch.randint(0, n_pairs, (n_dup,), generator=g, device=device)
        dst = torch.randint(0, n_pairs, (n_dup,), generator=g, device=device)
        out[dst] = out[src].clone()                                                 # exact PCR-duplicate pairs
    bases = out.reshape(-1)
    pad = torch.zeros(64, dtype=torch.uint8, device=device)                         # 16 B read slack for the ABI
    bases = torch.cat([bases, pad])
    rec_off = torch.arange(0, 2 * n_pairs + 1, device=device, dtype=torch.int64) * read_len
    return bases, rec_off


def ragged_paired_reads(genomes, n_pairs, min_len=35, max_len=151, n_frac=0.001, seed=0, **kw):
    """Pairs as paired_reads() makes them, every mate then trimmed to its own length in [min_len, max_len] (adapter/quality
    trimming) and with a fraction n_frac of the bases called N: the 'honest input' variant of the short-read workload (no two
    lanes of a wavefront walk the same number of k-mers; the exact ASCII->2-bit path is taken).  -> (bases, rec_off) as above."""
    device = genomes.device
    bases, _ = paired_reads(genomes, n_pairs, read_len=max_len, seed=seed, **kw)
    g = _gen(device, seed + 911)
    full = bases[:n_pairs * 2 * max_len].reshape(2 * n_pairs, max_len)
    lens = torch.randint(min_len, max_len + 1, (2 * n_pairs,), generator=g, device=device)
    keep = torch.arange(max_len, device=device)[None, :] < lens[:, None]
    out = []
    step = 1 << 20
    for s in range(0, 2 * n_pairs, step):                                   # boolean compaction in slabs (bounded temporaries)
        blk = full[s:s + step][keep[s:s + step]]
        nmask = torch.rand(blk.shape, generator=g, device=device) < n_frac
        out.append(torch.where(nmask, torch.full_like(blk, 78), blk))
    out.append(torch.zeros(64, dtype=torch.uint8, device=device))
    rec_off = torch.zeros(2 * n_pairs + 1, dtype=torch.int64, device=device)
    rec_off[1:] = torch.cumsum(lens, 0)
    return torch.cat(out), rec_off


def decoy_sketches(n_genomes, c=200, device="cuda", seed=0, mean_len=3.3e6, sigma=0.45, kept_frac=0.87,
                   lo=0.5e6, hi=15e6):
    """Sketch-only genomes: n_kept = round(kept_frac * len / c) uniform u64 below the FracMinHash threshold (valid
    because mm_hash64 is a bijection).  -> (kmers int64 [N] (bit pattern of u64 < 2^63), genome_off int64 [n+1])"""
    g = _gen(device, seed)
    ln = torch.exp(torch.randn(n_genomes, generator=g, device=device, dtype=torch.float64) * sigma + math.log(mean_len))
    ln = ln.clamp(lo, hi)
    n_kept = (kept_frac * ln / c).round().long()
    off = torch.zeros(n_genomes + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(n_kept, 0)
    thr = (2**64 - 1) // c
    kmers = torch.randint(0, thr, (int(off[-1].item()),), generator=g, device=device, dtype=torch.int64)
    return kmers, off


def long_reads(genomes, total_bases, n50=10_000, sigma=0.8, err=0.05, abundance_sigma=1.0, seed=0, chunk_bases=1 << 28,
               min_len=500, max_len=200_000):
    """ONT-like single-end reads: log-normal lengths with the given N50 (length-weighted median = exp(mu + sigma^2)),
    substitution errors only (indels do not change the kernels' work).  -> (bases uint8 [+64 pad], rec_off int64 [n+1])."""
    device = genomes.device
    g = _gen(device, seed)
    n_gen, glen = genomes.shape
    mu = math.log(n50) - sigma * sigma
    mean_len = math.exp(mu + sigma * sigma / 2)
    n_guess = int(total_bases / mean_len * 1.2) + 16
    lens = torch.exp(torch.randn(n_guess, generator=g, device=device) * sigma + mu).long().clamp(min_len, min(max_len, glen - 1))
    csum = torch.cumsum(lens, 0)
    n = int(torch.searchsorted(csum, torch.tensor([total_bases], device=device)).item()) + 1
    lens = lens[:n]
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(lens, 0)
    total = int(off[-1].item())
    ab = torch.exp(torch.randn(n_gen, generator=g, device=device) * abundance_sigma)
    gid = torch.multinomial(ab, n, replacement=True, generator=g)
    start = (torch.rand(n, generator=g, device=device) * (glen - lens).float()).long().clamp(min=0)
    src0 = gid * glen + start
    flip = torch.rand(n, generator=g, device=device) < 0.5
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    lut = torch.tensor(_ACGT, dtype=torch.uint8, device=device)
    out = torch.empty(total + 64, dtype=torch.uint8, device=device)
    out[total:] = 0
    flat = genomes.reshape(-1)
    r0 = 0
    while r0 < n:                                       # chunks of whole reads
        r1 = int(torch.searchsorted(off, off[r0] + chunk_bases).item())
        r1 = max(r0 + 1, min(r1, n))
        b0, b1 = int(off[r0].item()), int(off[r1].item())
        rel = torch.arange(b1 - b0, device=device)
        rid = torch.repeat_interleave(torch.arange(r0, r1, device=device), lens[r0:r1])
        within = rel - (off[rid] - b0)
        fwd_idx = src0[rid] + within
        rev_idx = src0[rid] + (lens[rid] - 1 - within)
        f = flip[rid]
        bases = flat[torch.where(f, rev_idx, fwd_idx)]
        bases = torch.where(f, comp[bases.long()], bases)
        e = torch.rand(b1 - b0, generator=g, device=device) < err
        rnd = lut[torch.randint(0, 4, (b1 - b0,), generator=g, device=device)]
        out[b0:b1] = torch.where(e, rnd, bases)
        r0 = r1
    return out, off
