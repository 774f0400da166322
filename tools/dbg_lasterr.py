import os, sys, ctypes, numpy as np
sys.path.insert(0, os.getcwd())
import sylph_amd as S
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p
def last(tag):
    e = hip.hipGetLastError()
    print(tag, e, hip.hipGetErrorString(e), file=sys.stderr, flush=True)
ctx = S.Context(0)
last("after ctx")
rng = np.random.default_rng(0)
g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=200000).astype(np.uint8)
recs = [g[i:i + 150] for i in rng.integers(0, len(g) - 150, size=4000)]
off = np.zeros(len(recs) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(r) for r in recs])
b = np.concatenate(recs)
sk = S.ReadSketcher(ctx, c=20); last("after begin")
sk.push(b, off); last("after push")
r = sk.finish(); last("after finish")
sk.close()
gs = ctx.sketch_genome(g, np.array([0, len(g)], dtype=np.uint64), c=20); last("after genome")
db = S.Database(ctx, gs["genome_kmers"], np.array([0, len(gs["genome_kmers"])], dtype=np.uint64)); last("after db")
cc, o, cv = db.contain(r["kmers"], r["counts"]); last("after contain")
import torch
print(torch.cuda.is_available(), torch.cuda.device_count(), file=sys.stderr)
