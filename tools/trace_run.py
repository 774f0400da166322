import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
import sylph_amd as S
import synth
dev=torch.device('cuda',0)
comm=synth.random_genomes(100, 5_000_000, dev, 1, mutated_frac=0.0)
bases,off=synth.paired_reads(comm, 3_333_334, seed=5)
torch.cuda.synchronize()
ctx=S.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for it in range(3):
    print('--- step', it, file=sys.stderr)
    t=time.perf_counter(); sk=S.ReadSketcher(ctx, paired=True); t1=time.perf_counter()
    sk.push_device(bases.data_ptr(), off.data_ptr(), 2*3_333_334); t2=time.perf_counter()
    r=sk.finish_device(); t3=time.perf_counter(); sk.close(); t4=time.perf_counter()
    print(f'begin {1e3*(t1-t):.3f} push {1e3*(t2-t1):.3f} finish {1e3*(t3-t2):.3f} close {1e3*(t4-t3):.3f}', file=sys.stderr)
