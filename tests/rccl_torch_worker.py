"""One rank, backend nccl (= RCCL): the sharded containment call with its collectives served by torch.distributed ON the device
buffers (sylph_amd.shard.torch_device_comm: all_gather_into_tensor / all_to_all_single of torch's own communicator) and, beside
it, by the library's own RCCL communicator (rccl_comm: the id broadcast through the process group); both must return what the
unsharded batch call returns.  Started by tests/test_gpu_parity.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sylph_amd as S  # noqa: E402
from sylph_amd import shard as SH  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = S.Context(0, stream=stream.cuda_stream)
    rng = np.random.default_rng(12)
    thr = (2**64 - 1) // 200
    pool = np.unique(rng.integers(0, thr, size=50000, dtype=np.uint64))
    genomes = [rng.choice(pool, size=int(n), replace=False) for n in rng.integers(50, 1200, size=70)]
    db = np.concatenate(genomes)
    goff = np.zeros(len(genomes) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(g) for g in genomes])
    samples = [(np.sort(rng.choice(pool, size=n, replace=False)), rng.integers(0, 9, size=n).astype(np.uint32)) for n in (5000, 0, 12000)]
    bounds = S.shard_bounds(int(db.max()), 1)
    d1 = S.Database(ctx, db, goff, shard=(bounds, 1, 0))
    d0 = S.Database(ctx, db, goff)
    want = [x.copy() for x in d0.contain_batch(samples)]
    for name, comm in (("torch", SH.torch_device_comm(dist, dev)), ("library", SH.rccl_comm(dist, ctx, dev))):
        for subset in (samples, []):
            got = [x.copy() for x in d1.contain_batch_sharded(comm, subset)]
            ref = want if subset else [x.copy() for x in d0.contain_batch([])]
            assert all(np.array_equal(a, b) for a, b in zip(got, ref)), name
        comm.close()
    d1.close(); d0.close(); ctx.close()
    dist.destroy_process_group()
    print("RCCL_TORCH_OK")


if __name__ == "__main__":
    main()
