"""Sketch time of the bench's 1 Gbp sample over the (c, k) the CLI accepts: the seeding kernel's slot sizing, the survivors' pass
and the dedup/count stage all scale with 1/c.  GPU box: python tools/ck_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch
import sylph_amd as S
import synth
dev = torch.device("cuda", 0)
ctx = S.Context(0)
genomes = synth.random_genomes(100, 5_000_000, dev, 3, mutated_frac=0.0)
bases, off = synth.paired_reads(genomes, 3_333_334, seed=11)
torch.cuda.synchronize()
nb = int(off[-1].item())
for k in (31, 21):
    for c in (1000, 200, 100, 50, 20):
        ts = []
        for rep in range(3):
            ctx.profile(True)
            torch.cuda.synchronize(); t = time.perf_counter()
            sk = S.ReadSketcher(ctx, c=c, k=k, paired=True)
            sk.push_device(bases.data_ptr(), off.data_ptr(), off.numel() - 1, nb)
            dk, dc, n, dup = sk.finish_device()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            st = {f: tuple(round(x, 3) for x in ctx.kernel_stats(f)) for f in ("seeds", "seeds_spill", "compact", "sort", "replay")}
            ctx.profile(False)
            sk.close()
        print(f"k={k} c={c}: sketch {min(ts) * 1e3:.2f} ms, table {n}, dup {dup}, kernels {st}", flush=True)
