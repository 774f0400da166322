import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
import sylph_amd as S
from sylph_amd import synth
dev=torch.device('cuda',0)
comm=synth.random_genomes(100, 5_000_000, dev, 1, mutated_frac=0.0)
bases,off=synth.paired_reads(comm, 3_333_334, seed=5)
torch.cuda.synchronize()
ctx=S.Context(0)
dk_, doff_ = synth.decoy_sketches(int(sys.argv[1]) if len(sys.argv)>1 else 20000, device=dev, seed=3)
db=S.Database(ctx, dk_.data_ptr(), doff_.data_ptr(), device_ptrs=True, n_genomes=len(doff_)-1)
ctx.synchronize()
def run(variant, steps=6):
    for it in range(steps):
        t=time.perf_counter(); sk=S.ReadSketcher(ctx, paired=True); t1=time.perf_counter()
        sk.push_device(bases.data_ptr(), off.data_ptr(), 2*3_333_334); t2=time.perf_counter()
        dk,dc,n,dup=sk.finish_device(); t3=time.perf_counter()
        if variant>=1:
            cc,co,cv=db.contain(dk,dc,device_ptrs=True,n=n)
        t4=time.perf_counter()
        if variant==2: time.sleep(0.02)
        sk.close()
        print(f'v{variant} step{it}: begin {1e3*(t1-t):.3f} push {1e3*(t2-t1):.3f} finish {1e3*(t3-t2):.3f} contain {1e3*(t4-t3):.3f}', file=sys.stderr)
run(0); run(1); run(2)
