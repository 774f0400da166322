import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch
import sylph_amd as S
import synth
dev = torch.device("cuda", 0)
ctx = S.Context(0)
for n_gen in (1, 10):
    genomes = synth.random_genomes(n_gen, 5_000_000, dev, 3, mutated_frac=0.0)
    bases, off = synth.long_reads(genomes, 2_500_000_000, seed=5)
    torch.cuda.synchronize()
    nb = int(off[-1].item())
    ts = []
    for rep in range(3):
        ctx.profile(True)
        torch.cuda.synchronize(); t = time.perf_counter()
        sk = S.ReadSketcher(ctx, c=100, k=31, paired=False)
        sk.push_device(bases.data_ptr(), off.data_ptr(), off.numel() - 1, nb)
        dk, dc, n, dup = sk.finish_device()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        st = {f: tuple(round(x, 3) for x in ctx.kernel_stats(f)) for f in ("seeds", "annotate", "compact", "sort", "replay", "replay_overflow")}
        ctx.profile(False)
        sk.close()
    print(f"long reads, {n_gen} genome(s) x 5 Mbp, {nb/1e9:.2f} Gbp ({nb/n_gen/5e6:.0f}x) c=100: sketch {min(ts)*1e3:.2f} ms, table {n}, kernels {st}", flush=True)
