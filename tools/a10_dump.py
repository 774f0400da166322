import os, sys
os.environ["SYLPH_HIP_A10_TRACE"] = "1"
os.environ["SYLPH_HIP_A10_DUMP"] = "gpurun_out/a10/occ.bin"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import sylph_amd as S
import tests.test_gpu_parity as T
ctx = S.Context(0)
rng = np.random.default_rng(77)
genome = T.random_seq(rng, 30000)
recs = T.make_reads(rng, genome, 4000, 150, paired=True, dup_frac=0.3)
b, off = T.concat(recs)
np.savez("gpurun_out/a10/reads.npz", b=b, off=off)
g = T._sketch_gpu_once(ctx, b, off, True, False, S.SEED_SCALAR, 20, 31, 1, dedup_fpr=0.05, dedup_capacity=2500)
print(len(g["kmers"]), g["dup_removed"])
