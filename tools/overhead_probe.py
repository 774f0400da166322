import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
import sylph_amd as S
from sylph_amd import synth
dev=torch.device('cuda',0)
comm=synth.random_genomes(100, 5_000_000, dev, 1, mutated_frac=0.0)
bases,off=synth.paired_reads(comm, 3_333_334, seed=5)
torch.cuda.synchronize()
ctx=S.Context(0)
# sample table once, to build a DB that produces hits (genomes = slices of the sample's own k-mers) + decoys
sk=S.ReadSketcher(ctx, paired=True); sk.push_device(bases.data_ptr(), off.data_ptr(), 2*3_333_334); r=sk.finish(); sk.close()
km=r['kmers']; G=100; L=len(km)//G
own=torch.from_numpy(km[:G*L].view(np.int64).copy()).to(dev)
dk_, doff_ = synth.decoy_sketches(int(sys.argv[1]) if len(sys.argv)>1 else 20000, device=dev, seed=3)
allk=torch.cat([own, dk_]); alloff=torch.cat([torch.arange(0,G*L+1,L,device=dev,dtype=torch.int64), doff_[1:]+G*L])
torch.cuda.synchronize()
db=S.Database(ctx, allk.data_ptr(), alloff.data_ptr(), device_ptrs=True, n_genomes=len(alloff)-1)
ctx.synchronize()
if os.environ.get('CPU_COPIES'):
    for i in range(int(os.environ['CPU_COPIES'])):
        x = comm[i % 100].cpu().numpy(); x = None
if os.environ.get('GENOMES'):
    o1=np.array([0,5_000_000],dtype=np.uint64)
    for i in range(int(os.environ['GENOMES'])):
        ctx.sketch_genome(comm[i % 100].cpu().numpy(), o1)
if os.environ.get('PROFILE'): ctx.profile(True)
keep=[]
print('db', db.n_genomes, db.n_kmers, 'sample', len(km), file=sys.stderr)
def run(variant, steps=8):
    if os.environ.get('SLEEP'): time.sleep(float(os.environ['SLEEP']))
    for it in range(steps):
        t=time.perf_counter(); sk=S.ReadSketcher(ctx, paired=True); t1=time.perf_counter()
        sk.push_device(bases.data_ptr(), off.data_ptr(), 2*3_333_334); t2=time.perf_counter()
        dk,dc,n,dup=sk.finish_device(); t3=time.perf_counter()
        nh=0
        if variant>=1:
            cc,co,cv=db.contain(dk,dc,device_ptrs=True,n=n); nh=(len(cv), int(cc.sum()), n)
        t4=time.perf_counter()
        sk.close()
        if os.environ.get('RETAIN') and variant>=1: keep.append((cc,co,cv))
        print(f'v{variant} step{it}: push {1e3*(t2-t1):.3f} finish {1e3*(t3-t2):.3f} contain {1e3*(t4-t3):.3f} hits {nh}', file=sys.stderr)
run(int(os.environ.get('VARIANT','1')), steps=int(os.environ.get('STEPS','8')))
