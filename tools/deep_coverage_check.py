"""Sketch time of 1 Gbp of 2x150 bp reads drawn from 1 / 4 / 100 genomes of 5 Mbp (200x / 50x / 2x coverage): how the dedup/count
stage copes with k-mers of hundreds of occurrences.  GPU box: python tools/deep_coverage_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
import sylph_amd as S
import synth
dev = torch.device("cuda", 0)
ctx = S.Context(0)
CASES = {"1": ((1, 5_000_000),), "all": ((1, 5_000_000), (4, 5_000_000), (100, 5_000_000)),
         "extreme": ((1, 1_000_000), (1, 200_000), (1, 30_000), (20, 30_000))}[os.environ.get("DEEP_CASES", "all")]
for n_gen, glen in CASES:
    genomes = synth.random_genomes(n_gen, glen, dev, 3, mutated_frac=0.0)
    bases, off = synth.paired_reads(genomes, 3_333_334, seed=11)
    torch.cuda.synchronize()
    nb = int(off[-1].item())
    fpr = float(os.environ.get("DEEP_DEDUP_FPR", "0"))        # > 0: the pairs behind the cuckoo filter (csrc/a10.hip) instead of the exact set
    for paired in ((True,) if fpr else (True, False)):
        ts = []
        for rep in range(4):
            ctx.profile(True)
            torch.cuda.synchronize(); t = time.perf_counter()
            sk = S.ReadSketcher(ctx, c=200, k=31, paired=paired, dedup_fpr=fpr if paired else 0.0)
            sk.push_device(bases.data_ptr(), off.data_ptr(), off.numel() - 1, nb)
            dk, dc, n, dup = sk.finish_device()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            st = {f: ctx.kernel_stats(f) for f in ("seeds", "compact", "sort", "replay", "replay_overflow", "a10")}
            st = {k: (round(v[0], 3), v[1]) for k, v in st.items()}
            ctx.profile(False)
            sk.close()
        print(f"{n_gen} genomes x {glen} ({nb/n_gen/glen:.0f}x coverage) paired={paired}{' filter dedup fpr ' + str(fpr) if fpr and paired else ''}: sketch {min(ts)*1e3:.2f} ms, table {n}, dup {dup}, kernels {st}", flush=True)
