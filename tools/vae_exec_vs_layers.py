"""Where do the C executor (scail_vae_decode / scail_vae_encode) and the layer-by-layer host path part at BASELINE config 4's size?
(tests/test_vae_gpu.py::test_c_executor_equals_layerwise_path holds bit-identity at small sizes; tests/debug_vae_decode_bisect.py saw
cos 0.99996 between the two at 81 x 512 x 896.)  Runs each path twice (run-to-run determinism), then maps the differing elements.

  python tools/vae_exec_vs_layers.py [decode|encode] [T h w]"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scail_amd.wan_vae import WanVAE_           # noqa: E402

DEV = "cuda:0"


def where(tag, a, b):
    d = (a - b).abs()[0]                      # (C, T, H, W)
    nz = d > 0
    n = int(nz.sum())
    print(f"{tag}: equal {n == 0}; {n} of {d.numel()} elements differ ({n / d.numel():.3e}), max abs diff {float(d.max()):.3e}", flush=True)
    if n == 0:
        return
    per_t = nz.sum(dim=(0, 2, 3)).tolist()
    print("  differing elements per frame:", per_t)
    t = max(range(len(per_t)), key=lambda i: per_t[i])
    m = nz[:, t].any(dim=0)                   # (H, W)
    rows = m.any(dim=1).nonzero().flatten()
    cols = m.any(dim=0).nonzero().flatten()
    print(f"  frame {t}: rows {int(rows[0])}..{int(rows[-1])} ({rows.numel()} rows), cols {int(cols[0])}..{int(cols[-1])} ({cols.numel()} cols)")
    # 32 x 32 pixel blocks with any difference (coarse map, up to 16 x 28 cells)
    H, W = m.shape
    bh, bw = max(H // 16, 1), max(W // 28, 1)
    cells = m[:H // bh * bh, :W // bw * bw].reshape(H // bh, bh, W // bw, bw).float().mean(dim=(1, 3))
    for r in range(cells.shape[0]):
        print("   ", "".join(" .:-=+*#%@"[min(int(float(c) * 9.999), 9)] for c in cells[r]))


def frame_sums(x):
    """exact per-frame integer checksum of a (T, H, W, C) bf16 tensor (the bit patterns summed as int16 into int64)."""
    return x.view(torch.int16).sum(dim=(1, 2, 3), dtype=torch.int64)


def trace_both(m, f, inp):
    """the launch-by-launch walk: scail_vae_set_trace on the C executor against the same launches of the layer path (ops.* wrapped)."""
    from scail_amd import ops
    rec_c, rec_p = [], []
    m.use_c_exec = True
    m._c().set_trace(lambda i, op, t: rec_c.append((op, tuple(t.shape), frame_sums(t))))
    f(inp)
    m._c().set_trace(None)
    m.use_c_exec = False
    m._cvae = None
    torch.cuda.empty_cache()
    o_conv, o_norm, o_rms, o_attn, o_rn = ops.conv3d_cl, ops.conv3d_cl_norm, ops.rms_silu, m._attn, ops.conv3d_cl_resid_norm

    def conv(x, wp, out_shape, **kw):
        y = o_conv(x, wp, out_shape, **kw)
        if kw.get("ot_mul", 1) == 1:
            rec_p.append(("conv", tuple(y.shape), frame_sums(y)))
        return y

    def norm(x, wp, gamma, out=None):
        y = o_norm(x, wp, gamma, out=out)
        rec_p.append(("conv_norm", tuple(y.shape), frame_sums(y)))
        return y

    def rms(x, gamma, silu=True, out=None):
        y = o_rms(x, gamma, silu=silu, out=out)
        rec_p.append(("rms_silu", tuple(y.shape), frame_sums(y)))
        return y

    def attn(W, n, x):
        y = o_attn(W, n, x)
        rec_p.append(("attn", tuple(y.shape), frame_sums(y)))
        return y

    def resid_norm(x, wp, resid, gamma, want_raw=True):
        raw, nrm = o_rn(x, wp, resid, gamma, want_raw=want_raw)
        if raw is not None:
            rec_p.append(("conv", tuple(raw.shape), frame_sums(raw)))
        rec_p.append(("conv_resid_norm", tuple(nrm.shape), frame_sums(nrm)))
        return raw, nrm

    ops.conv3d_cl, ops.conv3d_cl_norm, ops.rms_silu, m._attn, ops.conv3d_cl_resid_norm = conv, norm, rms, attn, resid_norm
    try:
        f(inp)
    finally:
        ops.conv3d_cl, ops.conv3d_cl_norm, ops.rms_silu, ops.conv3d_cl_resid_norm = o_conv, o_norm, o_rms, o_rn
        del m._attn
    print(f"launch walk: C executor {len(rec_c)} records, layer path {len(rec_p)} records")
    first = True
    for i, (a, b) in enumerate(zip(rec_c, rec_p)):
        same_kind = a[0] == b[0] and a[1] == b[1]
        eq = same_kind and bool(torch.equal(a[2], b[2]))
        flag = "==" if eq else ("!=" if same_kind else "??")
        if not eq or i < 3:
            nd = int((a[2] != b[2]).sum()) if same_kind else -1
            print(f"  {i:3d} {flag} C: {a[0]:9s} {str(a[1]):24s} | layers: {b[0]:9s} {str(b[1]):24s} frames differing: {nd} of {a[1][0]}"
                  + ("   <-- first difference" if (not eq and first) else ""))
            first = first and eq


def main():
    direction = sys.argv[1] if len(sys.argv) > 1 else "decode"
    T, h, w = (int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (21, 64, 112)
    torch.manual_seed(4321)
    m = WanVAE_(dim=96, z_dim=16, device=DEV)             # random-init weights of the shipped architecture
    g = torch.Generator(device=DEV).manual_seed(5)
    if direction == "decode":
        inp = torch.randn(1, 16, T, h, w, device=DEV, generator=g).to(torch.bfloat16).float()
        f = m.decode
    else:
        inp = (torch.rand(1, 3, 1 + 4 * (T - 1), 8 * h, 8 * w, device=DEV, generator=g) * 2 - 1).to(torch.bfloat16).float()
        f = m.encode
    with torch.no_grad():
        c1 = f(inp); c2 = f(inp)
        where("C executor run 1 vs run 2", c1, c2)
        del c2
        m.use_c_exec = False
        m._cvae = None
        torch.cuda.empty_cache()
        p1 = f(inp); p2 = f(inp)
        where("layer path run 1 vs run 2", p1, p2)
        del p2
        where("C executor vs layer path", c1, p1)
        del c1, p1
        trace_both(m, f, inp)


if __name__ == "__main__":
    main()
