"""Time the row kernels of a config-2 layer (97 664 x 5120 bf16): ln_modulate, layernorm_affine, rmsnorm_rope (in place, strided view)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import ops
DEV = "cuda"
def timeit(fn, iters=20):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev); return ts[len(ts) // 2]
B, L, D = 2, 48832, 5120
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.randn(B, L, D, device=DEV, generator=g).to(torch.bfloat16)
y = torch.empty_like(x)
sh, sc = torch.randn(B, D, device=DEV, generator=g), torch.randn(B, D, device=DEV, generator=g)
w, b = torch.randn(D, device=DEV, generator=g), torch.randn(D, device=DEV, generator=g)
qkv = torch.randn(B, L, 3 * D, device=DEV, generator=g).to(torch.bfloat16)
cos, sin = torch.randn(L, 64, device=DEV, generator=g), torch.randn(L, 64, device=DEV, generator=g)
gb = 2.0 * B * L * D * 2 / 1e9
out = {}
for name, fn in (("ln_modulate", lambda: ops.ln_modulate(x, sh, sc, out=y)), ("layernorm_affine", lambda: ops.layernorm_affine(x, w, b, out=y)),
                 ("rmsnorm_rope", lambda: ops.rmsnorm_rope(qkv[..., D:2 * D], w, cos, sin, rows_per_batch=L))):
    ms = timeit(fn)
    out[name] = {"ms": ms, "TB/s": gb / ms}
print(json.dumps(out))
