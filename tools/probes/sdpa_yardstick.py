"""Yardstick only (never on the product path): the flash attention PyTorch-ROCm ships, which is what the reference calls
(sat/transformer_defaults.py:67-72), on an 8-head slice of the config-2 self-attention shape."""
import torch, time, json
import torch.nn.functional as F
dev='cuda'
B,H,L,D=2,8,48832,128
q=torch.randn(B,H,L,D,device=dev,dtype=torch.bfloat16); k=torch.randn_like(q); v=torch.randn_like(q)
fl=4.0*B*H*L*L*D
for name in ("default",):
    for _ in range(2): o=F.scaled_dot_product_attention(q,k,v)
    torch.cuda.synchronize()
    ts=[]
    for _ in range(5):
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record(); o=F.scaled_dot_product_attention(q,k,v); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ms=sorted(ts)[2]
    print(json.dumps(dict(case="torch SDPA (vendor flash attention yardstick)", B=B,H=H,L=L,ms=ms,tflops=fl/ms/1e9)))
