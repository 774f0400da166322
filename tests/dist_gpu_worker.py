"""Worker for tests/test_gpu_parity.py::test_sharded_containment_two_ranks_one_gpu — run under torch.distributed.run.
Each rank holds one k-mer-range shard of the database on the GPU and calls sylph_db_contain_batch_sharded (the library's own
exchange: slice boundaries all-gathered, slices all-to-all, one probe launch, hit groups all-to-all to the owners) with its own
samples; the collectives go through torch.distributed (gloo) callbacks because RCCL cannot put two ranks on one device.
Every rank checks its samples against the oracle over the WHOLE database."""
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
    torch.cuda.set_device(dev)
    rng = np.random.default_rng(5)
    thr = O.threshold(200)
    pool = np.unique(rng.integers(0, thr, size=60000, dtype=np.uint64))
    lens = rng.integers(0, 1500, size=61)
    lens[5] = 0
    lens[9] = 49
    genomes = [rng.choice(pool, size=int(n), replace=False) for n in lens]
    genomes[20] = np.concatenate([genomes[20], pool[:300]])
    genomes[21] = np.concatenate([genomes[21], pool[:300], genomes[21][:10]])     # shared and repeated k-mers
    full = np.concatenate(genomes)
    goff = np.zeros(len(genomes) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g) for g in genomes])
    G = len(genomes)
    ctx = S.Context(0)
    ctx.set_option("shard_reduce", os.environ.get("SYLPH_TEST_SHARD_REDUCE", "alltoall"))   # round 6: the hits by all-to-all, or by ONE all-gather
    if os.environ.get("SYLPH_TEST_SHARD_BY", "kmer") == "genome":       # round 5: the cut by genome (north_star's wording), same exchange
        gb = SH.genome_shard_bounds(goff, world)
        db = S.Database(ctx, full, goff, genome_shard=(gb, world, rank))
        assert db.n_kmers == int(goff[int(gb[rank + 1])] - goff[int(gb[rank])]) and 0 < db.n_kmers < len(full)
    else:
        bounds = S.shard_bounds(int(full.max()), world)
        db = S.Database(ctx, full, goff, shard=(bounds, world, rank))
        assert 0 < db.n_kmers < len(full)
    comm = SH.torch_callback_comm(dist, dev)
    r2 = np.random.default_rng(100 + rank)
    for step in range(3):
        n_local = [2, 3][rank] if step < 2 else [0, 1][rank]         # different batch sizes per rank; one rank may bring none
        samples = []
        for i in range(n_local):
            k = np.sort(r2.choice(pool, size=9000 + 1000 * rank + 10 * i, replace=False)) if (step, i) != (0, 1) else np.zeros(0, np.uint64)
            samples.append((k, r2.integers(0, 9, size=len(k)).astype(np.uint32)))
        if step == 1:                                                # device-resident tables
            tk = [torch.from_numpy(k.view(np.int64)).to(dev) for k, _ in samples]
            tc = [torch.from_numpy(c.view(np.int32)).to(dev) for _, c in samples]
            torch.cuda.synchronize()
            cc, off, covs = db.contain_batch_sharded(comm, [(a.data_ptr() if a.numel() else 0, b.data_ptr() if b.numel() else 0, a.numel())
                                                            for a, b in zip(tk, tc)], device_ptrs=True)
        else:
            cc, off, covs = db.contain_batch_sharded(comm, samples)
        assert len(cc) == n_local * G
        for s, (k, c) in enumerate(samples):
            ecc, ecov, _ = O.contain(k, c, full, goff)
            assert np.array_equal(cc[s * G:(s + 1) * G], ecc), (rank, step, s)
            for g in range(G):
                assert np.array_equal(covs[int(off[s * G + g]):int(off[s * G + g + 1])].astype(np.uint32), np.sort(ecov[g])), (rank, step, s, g)
            assert len(k) == 0 or int(ecc.sum()) > 0
    # one rank fails between the collectives of a batch: EVERY rank must come back from the call with an error (nobody waits in
    # the second all-to-all for hits that never come), and the next batch works as if nothing had happened
    k = np.sort(r2.choice(pool, size=5000, replace=False))
    c = np.ones(len(k), np.uint32)
    if rank == world - 1:
        ctx.set_option("fail_next_shard_probe", "1")
    try:
        db.contain_batch_sharded(comm, [(k, c)])
        raise AssertionError(f"rank {rank}: the batch with a failing rank returned normally")
    except S.SylphHipError as e:
        assert ("injected failure" in str(e)) == (rank == world - 1), str(e)
        assert rank == world - 1 or f"rank {world - 1}" in str(e), str(e)
    cc, off, covs = db.contain_batch_sharded(comm, [(k, c)])
    ecc, _, _ = O.contain(k, c, full, goff)
    assert np.array_equal(cc, ecc)
    db.close()
    comm.close()
    ctx.close()
    dist.barrier()
    if rank == 0:
        print("DIST_GPU_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
