import os, sys, torch
sys.path.insert(0, "/root/repo")
from pvpuformer_amd import ops
dev="cuda"; D,Hh=768,12
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps
for n in (196, 784):
    for nb in (3, 6, 12, 24, 48, 96):
        qkv=torch.randn(nb*n,3*D,device=dev).to(torch.bfloat16)
        o=torch.empty(nb*n,D,device=dev,dtype=torch.bfloat16); lse=torch.empty(nb*Hh,n,device=dev)
        us=timeit(lambda: ops.attn_fwd((qkv,0),(qkv,D),(qkv,2*D),o,lse,nb,Hh,n,64,3*D,D,0.125))
        print(f"n={n} nb={nb:3d} blocks={(n+127)//128*nb*Hh:5d} {us:8.1f} us")
