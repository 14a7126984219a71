"""Stage-by-stage comparison of the product VAE decoder (layer-by-layer host path AND the C executor's final output) with the fp32
oracle evaluated on the GPU, at BASELINE config 4's own size.  Written to localise the full-size decode mismatch that
tests/test_vae_gpu.py::test_vae_fullsize_vs_fp32[decode] reported on its first hardware run (round 4).

  python tests/debug_vae_decode_bisect.py [T h w]        default 21 64 112      (a checker: lives under tests/ because it evaluates the oracle)
Per stage: cosine(product bf16, oracle fp32), cosine per frame (first / worst), and -- for the stage where they part -- the oracle op
re-evaluated on the PRODUCT's input, which tells which side moved.  For 'up' stages torch's nearest-exact interpolate is also checked
against an index restatement (repeat_interleave) because the tensors pass 2^31 elements there."""
import sys
import os
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import wan_vae_oracle as V          # noqa: E402  (test infrastructure: the oracle is the checker here)
from scail_amd import ops                       # noqa: E402
from scail_amd.wan_vae import WanVAE_           # noqa: E402

DEV = "cuda:0"


def cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def cmp(tag, got_cl, want):
    """got_cl (T,H,W,C) bf16; want (1,C,T,H,W) fp32."""
    w = want[0].permute(1, 2, 3, 0)
    T = w.shape[0]
    per = []
    for t in range(T):
        per.append(cos(got_cl[t].float(), w[t]))
    worst = min(range(T), key=lambda t: per[t])
    err = 0.0
    for t in range(T):
        err = max(err, float((got_cl[t].float() - w[t]).abs().max()))
    print(f"{tag:34s} shape {tuple(got_cl.shape)}  cos all {sum(per) / T:.6f}  first {per[0]:.6f}  worst frame {worst}: {per[worst]:.6f}  last {per[-1]:.6f}  max abs {err:.3e}",
          flush=True)
    return per[worst]


def to_ncthw(x_cl):
    return x_cl.float().permute(3, 0, 1, 2).unsqueeze(0).contiguous()


def main():
    T, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (21, 64, 112)
    cfg = V.VAEConfig(dim=96, z_dim=16)
    sd = V.make_state_dict(cfg, seed=4321)
    m = WanVAE_(dim=96, z_dim=16, device=DEV)
    m.load_state_dict(sd, strict=True)
    g = torch.Generator(device=DEV).manual_seed(5)
    z = torch.randn(1, 16, T, h, w, device=DEV, generator=g).to(torch.bfloat16).float()
    sdg = {k: v.to(DEV) for k, v in sd.items()}
    V.CONV_IMPL = "taps"
    W = m.prepare()
    with torch.no_grad():
        full_c = m.decode(z)                          # C executor
        m.use_c_exec = False
        m._cvae = None
        torch.cuda.empty_cache()
        # ---- product, stage by stage
        x = ops.to_channels_last(z[0], m.z_dim, a=W["std"], b=W["mean"])
        mean = torch.tensor(V.LATENT_MEAN[:16], device=DEV).view(1, -1, 1, 1, 1)
        std = torch.tensor(V.LATENT_STD[:16], device=DEV).view(1, -1, 1, 1, 1)
        o = z * std + mean
        cmp("scale+shift", x[..., :16], o)
        x = ops.conv3d_cl(x, W["conv2"], x.shape[:3]); o = V.causal_conv3d(o, sdg["conv2.weight"], sdg["conv2.bias"])
        cmp("conv2", x[..., :16], o)
        x = ops.conv3d_cl(x, W["decoder.conv1"], x.shape[:3]); o = V.causal_conv3d(o, sdg["decoder.conv1.weight"], sdg["decoder.conv1.bias"])
        cmp("decoder.conv1", x, o)
        x, _ = m._res(W, "decoder.middle.0", x); o = V.residual_block(sdg, "decoder.middle.0", o); cmp("middle.0", x, o)
        x = m._attn(W, "decoder.middle.1", x); o = V.attention_block(sdg, "decoder.middle.1", o); cmp("middle.1 (attn)", x, o)
        x, _ = m._res(W, "decoder.middle.2", x); o = V.residual_block(sdg, "decoder.middle.2", o); cmp("middle.2", x, o)
        for kind, n, a, b in m.decoder_plan():
            x_in = x
            if kind == "res":
                x, _ = m._res(W, n, x); o = V.residual_block(sdg, n, o)
            else:
                x = m._up(W, n, x, b)        # (no next_gamma: plain raw output)
                mode = "upsample3d" if b else "upsample2d"
                o = V.upsample(sdg, n, o, mode)
            c = cmp(f"{kind} {n} {b if kind != 'res' else ''}", x, o)
            if c < 0.995:
                # which side moved?  the oracle op on the product's own input
                xi = to_ncthw(x_in)
                o2 = V.residual_block(sdg, n, xi) if kind == "res" else V.upsample(sdg, n, xi, "upsample3d" if b else "upsample2d")
                cmp("  product vs oracle(product input)", x, o2)
                if kind != "res":
                    # the interpolate alone, against an index restatement, on the frames of the oracle's input
                    xt = xi
                    if b and xt.shape[2] > 1:
                        tail = V.causal_conv3d(xt[:, :, 1:], sdg[n + ".time_conv.weight"], sdg[n + ".time_conv.bias"])
                        bb, c2, tt, hh, ww = tail.shape
                        tail = tail.reshape(bb, 2, c2 // 2, tt, hh, ww).permute(0, 2, 3, 1, 4, 5).reshape(bb, c2 // 2, 2 * tt, hh, ww)
                        xt = torch.cat([xt[:, :, :1], tail], dim=2)
                    y = xt.permute(0, 2, 1, 3, 4).reshape(xt.shape[2], xt.shape[1], xt.shape[3], xt.shape[4])
                    yi = torch.nn.functional.interpolate(y, scale_factor=(2.0, 2.0), mode="nearest-exact")
                    bad = 0
                    for f in range(y.shape[0]):
                        ref = y[f].repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
                        bad += int(not torch.equal(ref, yi[f]))
                    print(f"  torch nearest-exact on {tuple(y.shape)} ({y.numel() * 4:.3e} output elements): {bad} of {y.shape[0]} frames differ from the index restatement",
                          flush=True)
                    del y, yi
                del o2, xi
                o = to_ncthw(x)                # continue from the product's tensor so later stages are judged on their own
                print("  (oracle chain re-seeded from the product's tensor)")
            del x_in
        ops.rms_silu(x, W["decoder.head.0.gamma"], out=x)
        o = torch.nn.functional.silu(V.rms_norm(o, sdg["decoder.head.0.gamma"]))
        cmp("head norm+silu", x, o)
        x = ops.conv3d_cl(x, W["decoder.head.2"], x.shape[:3]); o = V.causal_conv3d(o, sdg["decoder.head.2.weight"], sdg["decoder.head.2.bias"])
        cmp("head conv", x[..., :3], o)
        lay = ops.from_channels_last(x, 3).unsqueeze(0)
        print("layer path vs C executor identical:", torch.equal(lay, full_c), " cos", cos(lay, full_c))
        print("from_channels_last vs oracle:", cos(lay, o))


if __name__ == "__main__":
    main()
