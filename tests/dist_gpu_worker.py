"""Worker for tests/test_gpu_parity.py::test_sharded_containment_two_ranks_one_gpu — run under torch.distributed.run.
Each rank owns a genome shard resident on the GPU (real HIP probe), samples are exchanged with gloo, and every rank
checks its own sample against the oracle over the whole database."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sylph_amd as S  # noqa: E402
from oracle import oracle as O  # noqa: E402
from sylph_amd import shard as SH  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    thr = O.threshold(200)
    pool = np.unique(rng.integers(0, thr, size=60000, dtype=np.uint64))
    lens = rng.integers(0, 1500, size=61)
    lens[5] = 0
    lens[9] = 49
    genomes = [rng.choice(pool, size=int(n), replace=False) for n in lens]
    owner = SH.partition_genomes(lens, world)
    rank_genomes = [np.nonzero(owner == r)[0] for r in range(world)]
    mine = rank_genomes[rank]
    shard_k = np.concatenate([genomes[g] for g in mine]) if len(mine) else np.zeros(0, dtype=np.uint64)
    shard_off = np.zeros(len(mine) + 1, dtype=np.uint64)
    shard_off[1:] = np.cumsum([len(genomes[g]) for g in mine])
    ctx = S.Context(0)
    db = S.Database(ctx, shard_k, shard_off)
    r2 = np.random.default_rng(100 + rank)
    sk = np.sort(r2.choice(pool, size=9000 + 1000 * rank, replace=False))
    sc = r2.integers(0, 9, size=len(sk)).astype(np.uint32)
    tk = torch.from_numpy(sk.view(np.int64)).to(dev)
    tc = torch.from_numpy(sc.view(np.int32)).to(dev)
    torch.cuda.synchronize()

    def contain_fn(k, c):
        k, c = k.to(dev).contiguous(), c.to(dev).contiguous()
        torch.cuda.synchronize()
        return db.contain(k.data_ptr(), c.data_ptr(), device_ptrs=True, n=k.numel())

    group = SH.TorchGroup(dist, dev)
    res = SH.exchange_and_profile(contain_fn, group, tk, tc, owner, rank_genomes)
    full = np.concatenate(genomes)
    goff = np.zeros(len(genomes) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum(lens)
    ecc, ecov, _ = O.contain(sk, sc, full, goff)
    assert np.array_equal(res["contain_count"], ecc), (rank, res["contain_count"][:10], ecc[:10])
    for g in range(len(genomes)):
        got = res["covs"][int(res["cov_off"][g]):int(res["cov_off"][g + 1])]
        assert np.array_equal(got, np.sort(ecov[g])), (rank, g)
    assert int(ecc.sum()) > 0
    db.close()
    ctx.close()
    dist.barrier()
    if rank == 0:
        print("DIST_GPU_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
