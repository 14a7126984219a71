"""Sweep the q8 GEMM tile-group height (knob gemm_group_m) over the 14B step's per-token GEMM shapes."""
import torch, json, sys
sys.path.insert(0, '.')
from scail_amd import lib, ops
lib.load()
dev='cuda'
M=97664
def run(N,K):
    x=torch.randn(M,K,device=dev).to(torch.bfloat16); w=(torch.randn(N,K,device=dev)*0.02).to(torch.bfloat16); b=torch.randn(N,device=dev); y=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    def t(n=9):
        for _ in range(2): ops.gemm(x,w,b,out=y)
        ts=[]
        for _ in range(n):
            s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
            s.record(); ops.gemm(x,w,b,out=y); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        return sorted(ts)[len(ts)//2]
    res={}
    for rnd in range(3):
        for gm in (1,2,3,4,6,8):
            lib.tune_set("gemm_group_m", gm)
            res.setdefault(gm,[]).append(2.0*M*N*K/t()/1e9)
    print(json.dumps(dict(N=N,K=K,tflops={k:[round(v) for v in vs] for k,vs in res.items()})))
for N,K in ((15360,5120),(5120,5120),(5120,13824),(13824,5120)):
    run(N,K)
