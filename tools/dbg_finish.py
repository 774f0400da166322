import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import sylph_amd as S
z = np.load("tests/golden/k12_reads.npz")
def rec(b, o): return [b[int(o[i]):int(o[i + 1])] for i in range(len(o) - 1)]
R1, R2 = rec(z["r1_bases"], z["r1_off"]), rec(z["r2_bases"], z["r2_off"])
inter = lambda a, b: [x for p in zip(a, b) for x in p]
def concat(records):
    off = np.zeros(len(records) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(r) for r in records])
    return np.concatenate(records).astype(np.uint8), off
cases = {"k12_single": (R1, False, False), "k12_single_nodedup": (R1, False, True), "k12_single_x2": (R1 + R1, False, False),
         "k12_single_x6": (R1 * 6, False, False), "k12_paired": (inter(R1, R2), True, False), "k12_paired_x2": (inter(R1 + R1, R2 + R2), True, False)}
ctx = S.Context(0)
for name, (rr, paired, nd) in cases.items():
    b, off = concat(rr)
    for mode in (0, 1):
        print("case", name, mode, len(off) - 1, file=sys.stderr, flush=True)
        sk = S.ReadSketcher(ctx, paired=paired, no_dedup=nd, seed_mode=mode)
        sk.push(b, off)
        r = sk.finish()
        print("   ok", len(r["kmers"]), int(r["counts"].sum()), r["dup_removed"], file=sys.stderr, flush=True)
        sk.close()
