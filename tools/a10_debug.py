"""Debug aid: where the filters of the approximate pair dedup open — csrc/a10.hip (SYLPH_HIP_A10_TRACE) against the numpy
formulation over the oracle's item stream."""
import ctypes as C
import os, sys
os.environ["SYLPH_HIP_A10_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import sylph_amd as S
from oracle import oracle as O
import tests.test_gpu_parity as T
import tests.test_oracle as TO

L = O.lib()
L.orc_pair_filter_items.restype = C.c_uint64
L.orc_pair_filter_items.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int] + [C.c_void_p] * 4


def items(b, off, c, mode):
    n = L.orc_pair_filter_items(b.ctypes.data, off.ctypes.data, len(off) - 1, c, 31, mode, None, None, None, None)
    km, mk, rec, seed = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    L.orc_pair_filter_items(b.ctypes.data, off.ctypes.data, len(off) - 1, c, 31, mode, km.ctypes.data, mk.ctypes.data, rec.ctypes.data, seed.ctypes.data)
    return km, mk, rec, seed


def cuts(km, mk, fpr, cap0):
    h = TO._a10_item_hash(km, mk)
    n, begin, j, closed, out = len(km), 0, 0, [], []
    while True:
        fpr_j, cap_j = fpr * 0.9 ** j, cap0 << j
        idx = np.arange(begin, n)
        prior = np.zeros(len(idx), dtype=bool)
        for (fq, cq, members) in closed:
            prior |= np.isin(TO._a10_reduced_key(h[idx], fq, cq), members)
        g = TO._a10_reduced_key(h[idx], fpr_j, cap_j)
        reach = np.flatnonzero(~prior)
        _, first = np.unique(g[reach], return_index=True)
        inserts = np.sort(reach[first])
        if len(inserts) <= cap_j:
            return out
        cut = int(inserts[cap_j])
        out.append(begin + cut)
        closed.append((fpr_j, cap_j, np.unique(g[inserts[:cap_j]])))
        begin += cut
        j += 1


ctx = S.Context(0)
rng = np.random.default_rng(77)
genome = T.random_seq(rng, 30000)
c, n, Lr = 20, 4000, 150
recs = T.make_reads(rng, genome, n, Lr, paired=True, dup_frac=0.3)
b, off = T.concat(recs)
b, off = np.ascontiguousarray(b, dtype=np.uint8), np.ascontiguousarray(off, dtype=np.uint64)
for gm, om in T.MODES:
    km, mk, rec, seed = items(b, off, c, om)
    for fpr, cap in ((0.05, 2500),):
        for at in cuts(km, mk, fpr, cap):
            print(f"expected: mode {om}: next filter opens at item {at}: record {rec[at]}, seed #{seed[at]} of its record, marker {at & 1}", flush=True)
        e = O.sketch_reads_cuckoo_model(b, off, c=c, mode=om, fpr=fpr, initial_capacity=cap)
        for mode, seeds in (("auto", "auto"), ("generic", "unordered")):
            ctx.set_option("finish", mode)
            ctx.set_option("seeds", seeds)
            sys.stderr.flush()
            print(f"--- gpu {mode}/{seeds}", flush=True)
            g = T._sketch_gpu_once(ctx, b, off, True, False, gm, c, 31, 1, dedup_fpr=fpr, dedup_capacity=cap)
            ctx.set_option("finish", "auto")
            ctx.set_option("seeds", "auto")
            print("differing counts", int((g["counts"] != e["counts"]).sum()), flush=True)
