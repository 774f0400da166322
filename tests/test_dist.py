"""N>1 path on CPU: world_size-2 and -3 gloo runs of the k-mer-range sharded containment exchange.  The product's exchange runs
in the HIP library (csrc/shard.hip) and needs a GPU; what runs here is its executable specification,
sylph_amd.shard.model_contain_batch_sharded — the same five steps on host arrays — with the oracle standing in for the HIP probe.
Every rank must recover, for its OWN samples, exactly the single-process answer over the WHOLE database.  (tests/test_gpu_parity.py
runs the library's exchange itself over gloo with two ranks on one GPU, and over RCCL with one rank.)"""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import sylph_amd as S
from oracle import oracle as O
from sylph_amd import shard as SH


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_db(seed=5, G=37):
    rng = np.random.default_rng(seed)
    thr = O.threshold(200)
    pool = np.unique(rng.integers(0, thr, size=30000, dtype=np.uint64))
    lens = rng.integers(0, 900, size=G)
    lens[3] = 0
    lens[7] = 49
    genomes = [rng.choice(pool, size=int(n), replace=False) for n in lens]
    genomes[11] = np.concatenate([genomes[11], genomes[11][:5]])       # a k-mer twice in one genome: counted twice (contain.rs:632)
    return pool, genomes


def make_samples(pool, rank, n):
    rng = np.random.default_rng(100 + rank)
    out = []
    for i in range(n):
        k = np.sort(rng.choice(pool, size=3000 + 500 * rank + 100 * i, replace=False)) if (rank, i) != (1, 1) else np.zeros(0, np.uint64)
        out.append((k, rng.integers(0, 9, size=len(k)).astype(np.uint32)))
    return out


def shard_probe(genomes, lo, hi, min_number_kmers=50.0):
    """CPU stand-in for the resident shard: genomes cut to the k-mer range, full genome lengths kept for the :627 test."""
    cut = [g[(g >= lo) & ((g < hi) if hi else True)] for g in genomes]
    off = np.zeros(len(cut) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(g) for g in cut])
    flat = np.concatenate(cut) if len(cut) else np.zeros(0, np.uint64)
    full_len = [len(g) for g in genomes]

    def probe(k, c):
        cc, covs, _ = O.contain(k, c, flat, off, min_number_kmers=0.0)
        return [(g, int(x)) for g in range(len(cut)) if full_len[g] >= min_number_kmers for x in covs[g]]
    return probe


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pool, genomes = make_db()
        G = len(genomes)
        bounds = SH.shard_bounds(int(max(g.max() for g in genomes if len(g))), world)
        hi = int(bounds[rank + 1]) if rank + 1 < world else 0
        samples = make_samples(pool, rank, [2, 3, 0][rank])      # different batch sizes per rank, one rank may bring none
        cc, covs = SH.model_contain_batch_sharded(dist, bounds, G, samples, shard_probe(genomes, int(bounds[rank]), hi))
        db = np.concatenate(genomes)
        goff = np.zeros(G + 1, dtype=np.uint64)
        goff[1:] = np.cumsum([len(g) for g in genomes])
        ok = cc.shape == (len(samples), G)
        for s, (k, c) in enumerate(samples):
            ecc, ecov, _ = O.contain(k, c, db, goff)
            ok = ok and np.array_equal(cc[s], ecc) and all(np.array_equal(covs[s][g], np.sort(ecov[g])) for g in range(G))
            ok = ok and (len(k) == 0 or int(ecc.sum()) > 0)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_containment_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == [True] * world


def test_shard_bounds_match_library_and_cover_the_range():
    for mx in (0, 1, 6, 2**64 // 200, 2**64 - 2, 2**64 - 1):
        for w in (1, 2, 3, 8):
            b = SH.shard_bounds(mx, w)
            assert np.array_equal(b, S.shard_bounds(mx, w)), (mx, w)          # host function of the C ABI: runs without a GPU
            assert b[0] == 0 and all(b[i] <= b[i + 1] for i in range(w)) and int(b[w]) >= min(mx, 2**64 - 2)
    b = [int(x) for x in SH.shard_bounds(2**64 // 200, 8)]
    w = [b[i + 1] - b[i] for i in range(8)]
    assert max(w) - min(w) <= 2          # equal widths: uniform hashes => equal postings and equal slices per rank
