import sys, os, json, torch
sys.path.insert(0, "/root/repo")
from scail_amd import ops
DEV="cuda"
def timeit(fn, iters=10):
    fn(); fn()
    ev=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a,b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts=sorted(a.elapsed_time(b) for a,b in ev); return ts[len(ts)//2]
B,H,Lq=2,40,48832; D=H*128
g=torch.Generator(device=DEV).manual_seed(0)
rn=lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
qkv=rn(B,Lq,3*D); q=qkv[...,:D]
o=torch.empty(B,Lq,D,device=DEV,dtype=torch.bfloat16)
for Lk in (512, 1024, 2048):
    k,v=rn(B,Lk,D),rn(B,Lk,D); vt=ops.transpose_v(v,H)
    fl=4.0*B*H*Lq*Lk*128
    for pre in (True, False):
        ms=timeit(lambda: ops.flash_attn(q,k,vt,out=o,q_prescaled=pre))
        print(json.dumps({"Lk":Lk,"prescaled":pre,"ms":ms,"TFLOPs":fl/ms/1e9}))
