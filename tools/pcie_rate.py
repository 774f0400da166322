#!/usr/bin/env python3
"""Sketch-stage rate as a function of where the batch lives: device memory (bench.py's timed region), page-locked host memory
(SYLPH_MEM_HOST_PINNED: one DMA over PCIe) and ordinary host memory (SYLPH_MEM_HOST: staging memcpy + DMA).  Informational:
bench.py's `value` never includes the transfer."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sylph_amd as S  # noqa: E402
from sylph_amd import synth  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n_pairs = 1_000_000
    comm = synth.random_genomes(20, 2_000_000, dev, 1, mutated_frac=0.0)
    bases, off = synth.paired_reads(comm, n_pairs, seed=5)
    torch.cuda.synchronize()
    n_rec, n_bases = 2 * n_pairs, 2 * n_pairs * 150
    ctx = S.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    hb = bases[:n_bases].cpu().numpy()
    ho = off.cpu().numpy().astype(np.uint64)
    pb, po = S.PinnedBuffer(n_bases), S.PinnedBuffer(8 * (n_rec + 1))
    pb.array[:] = hb
    pov = po.array.view(np.uint64)
    pov[:] = ho
    res = {}
    for name in ("device", "pinned_host", "pageable_host"):
        best = 1e9
        for _ in range(4):
            t = time.perf_counter()
            sk = S.ReadSketcher(ctx, paired=True)
            if name == "device":
                sk.push_device(bases.data_ptr(), off.data_ptr(), n_rec, n_bases)
            elif name == "pinned_host":
                sk.push_pinned(pb, n_bases, pov)
            else:
                sk.push(hb, ho)
            sk.finish_device()
            sk.close()
            best = min(best, time.perf_counter() - t)
        res[name] = {"ms": round(best * 1e3, 3), "gbp_per_s": round(n_bases / 1e9 / best, 2)}
    print(res)


if __name__ == "__main__":
    main()
