"""N>1 path on CPU: world_size-2 (and 3) gloo run of the genome-sharded containment exchange (sylph_amd/shard.py),
with the oracle standing in for the HIP probe.  Each rank must recover, for its own sample, exactly the
single-process answer over the whole database."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from sylph_amd import shard as SH


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_db(seed=5, G=37):
    rng = np.random.default_rng(seed)
    thr = O.threshold(200)
    pool = np.unique(rng.integers(0, thr, size=30000, dtype=np.uint64))
    lens = rng.integers(0, 900, size=G)
    lens[3] = 0
    lens[7] = 49
    genomes = [rng.choice(pool, size=int(n), replace=False) for n in lens]
    return pool, genomes


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pool, genomes = _make_db()
        lens = np.array([len(g) for g in genomes])
        owner = SH.partition_genomes(lens, world)
        rank_genomes = [np.nonzero(owner == r)[0] for r in range(world)]
        mine = rank_genomes[rank]
        shard_k = np.concatenate([genomes[g] for g in mine]) if len(mine) else np.zeros(0, dtype=np.uint64)
        shard_off = np.zeros(len(mine) + 1, dtype=np.uint64)
        shard_off[1:] = np.cumsum([len(genomes[g]) for g in mine])
        rng = np.random.default_rng(100 + rank)
        sk = np.sort(rng.choice(pool, size=4000 + 500 * rank, replace=False))
        sc = rng.integers(0, 9, size=len(sk)).astype(np.uint32)

        def contain_fn(k, c):   # CPU stand-in for Database.contain: sorted covs per genome
            kk = k.numpy().view(np.uint64)
            cc_ = c.numpy().view(np.uint32)
            cc, covs, _ = O.contain(kk, cc_, shard_k, shard_off)
            off = np.zeros(len(mine) + 1, dtype=np.uint64)
            off[1:] = np.cumsum(cc.astype(np.uint64))
            flat = np.concatenate([np.sort(x) for x in covs]) if len(covs) and off[-1] else np.zeros(0, dtype=np.uint32)
            return cc, off, flat

        group = SH.TorchGroup(dist, torch.device("cpu"))
        res = SH.exchange_and_profile(contain_fn, group, torch.from_numpy(sk.view(np.int64)), torch.from_numpy(sc.view(np.int32)),
                                      owner, rank_genomes)
        # single-process answer over the whole database
        db = np.concatenate(genomes)
        goff = np.zeros(len(genomes) + 1, dtype=np.uint64)
        goff[1:] = np.cumsum(lens)
        ecc, ecov, _ = O.contain(sk, sc, db, goff)
        ok = np.array_equal(res["contain_count"], ecc)
        for g in range(len(genomes)):
            got = res["covs"][int(res["cov_off"][g]):int(res["cov_off"][g + 1])]
            ok = ok and np.array_equal(got, np.sort(ecov[g]))
        # replicated mode: one all-gather of the per-sample counts
        allc = SH.gather_counts(group, ecc, torch.device("cpu"))
        ok = ok and allc.shape == (world, len(genomes)) and np.array_equal(allc[rank].numpy().view(np.uint32), ecc)
        ret[rank] = bool(ok) and int(ecc.sum()) > 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_containment_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == [True] * world


def test_partition_balances():
    rng = np.random.default_rng(0)
    lens = rng.integers(500, 70000, size=113104)
    for w in (2, 4, 8):
        owner = SH.partition_genomes(lens, w)
        loads = np.bincount(owner, weights=lens, minlength=w)
        counts = np.bincount(owner, minlength=w)
        assert loads.max() / loads.mean() < 1.001 and counts.max() - counts.min() <= 1
    assert SH.partition_genomes(lens[:5], 1).tolist() == [0] * 5
