"""GPU parity tests for the Wan2.1 VAE HIP path: op level vs torch fp32 conv references, network level
vs the REAL chunked reference's outputs (tests/golden/vae_tiny.npz) and the CPU oracle.
Tolerance: bf16 activations through ~30 conv layers vs the fp32 reference -> rtol 3e-2 / atol 3e-2 and
cosine >= 0.999."""
import os

import numpy as np
import math
import pytest
import torch
import torch.nn.functional as F

from oracle import wan_vae_oracle as V

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bfr(x):
    return x.to(torch.bfloat16).float()


def _cl(x):      # (C,T,H,W) fp32 -> (T,H,W,C) bf16 on the GPU
    return x.permute(1, 2, 3, 0).contiguous().to(torch.bfloat16).to(DEV)


def _pl(y):      # (T,H,W,C) -> (C,T,H,W) fp32 cpu
    return y.float().cpu().permute(3, 0, 1, 2)


def _cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm()))


@pytest.fixture(params=[44, 4, 3, 2, 1, 0])
def conv_halo(request):
    """every path of the 3x3x3 convolution: the generated conv4 kernels where eligible (44: the product configuration), else / with the option
    "conv4" off (4) the halo-tile kernel with 32-channel slices padded / one W buffer, two frames per workgroup; measurement build: 32-channel
    slices swizzled / two W buffers, 48-channel slices, and the gather kernel"""
    from scail_amd import lib as L
    if request.param == 44:
        yield 44
        return
    L.set_option("conv4", 0)
    try:
        if request.param == 4:
            yield 4
            return
        if not L.ABLATIONS:
            pytest.skip("kernel variant of the measurement build (run with SCAIL_ABLATIONS=1)")
        L.tune_set("conv_halo", request.param)
        yield request.param
        L.tune_set("conv_halo", 4)
    finally:
        L.set_option("conv4", 1)


@pytest.mark.parametrize("cin,cout,k,thw", [(16, 32, (3, 3, 3), (5, 10, 12)), (96, 96, (3, 3, 3), (5, 10, 12)),
                                           (96, 192, (1, 1, 1), (5, 10, 12)), (32, 64, (3, 1, 1), (5, 10, 12)),
                                           (8, 96, (3, 3, 3), (5, 10, 12)), (192, 96, (3, 3, 3), (3, 17, 35)),
                                           (96, 200, (3, 3, 3), (2, 8, 16)), (64, 48, (3, 3, 3), (4, 9, 33)), (96, 3, (3, 3, 3), (3, 11, 21)),
                                           (32, 24, (3, 3, 3), (2, 8, 16)), (96, 96, (3, 3, 3), (1, 10, 12)), (192, 192, (3, 3, 3), (7, 9, 20))])
def test_causal_conv3d(cin, cout, k, thw, conv_halo):
    from scail_amd import ops
    g = torch.Generator().manual_seed(0)
    T, H, W = thw
    x = bfr(torch.randn(cin, T, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, *k, generator=g) / (cin * k[0] * k[1] * k[2]) ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = V.causal_conv3d(x[None], w, b)[0]
    wp = ops.prep_conv_weight(w.to(DEV), b.to(DEV))
    y = ops.conv3d_cl(_cl(x), wp, (T, H, W))
    torch.testing.assert_close(_pl(y)[:cout], ref, rtol=2e-2, atol=2e-2)
    if cout % 8:                                                   # residual tensors are channel-padded like the outputs
        return
    r = bfr(torch.randn(cout, T, H, W, generator=g))
    y2 = ops.conv3d_cl(_cl(x), wp, (T, H, W), resid=_cl(r))
    torch.testing.assert_close(_pl(y2)[:cout], ref + r, rtol=2e-2, atol=2e-2)


def _geom(Ti, H, W, Cin, To, N, Kpad, pt=2, ot_mul=1, ot_off=0):
    import ctypes as C
    return (C.c_int32 * 21)(Ti, H, W, Cin, To, H, W, 3, 3, 3, 1, 1, 1, pt, 1, 1, 0, ot_mul, ot_off, N, Kpad)


@pytest.mark.parametrize("cin,cout,thw", [(96, 96, (5, 33, 40)),        # 3 slices, ragged tiles, odd frame count
                                          (224, 192, (4, 16, 50)),      # 7 slices (the frame-slot ring wraps), 2 n tiles
                                          (32, 96, (2, 16, 16)),        # one slice, one tile
                                          (384, 384, (6, 20, 36))])     # 12 slices, 4 n tiles
def test_conv4_generated_kernels(cin, cout, thw):
    """the generated kernels (csrc/conv4.s; scail_conv3d_kernel_for == 4) against torch's fp32 convolution of the bf16-rounded operands:
    plain, residual, and -- on the first shape -- a chunk with its two cache frames in front (pt = 0), interleaved output frames
    (ot_mul / ot_off) and an output that is a channel slice of a wider tensor (ldc > N)."""
    import ctypes as C
    from scail_amd import lib as L, ops
    g = torch.Generator().manual_seed(7)
    T, H, W = thw
    x = bfr(torch.randn(cin, T, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5)
    b = torch.randn(cout, generator=g)
    r = bfr(torch.randn(cout, T, H, W, generator=g))
    ref = V.causal_conv3d(x[None], w, b)[0]
    wp = ops.prep_conv_weight(w.to(DEV), b.to(DEV))
    assert L.load().scail_conv3d_kernel_for(C.cast(_geom(T, H, W, cin, T, cout, wp["Kpad"]), C.c_void_p), cout, cout, 0) == 4
    assert L.load().scail_conv3d_kernel_for(C.cast(_geom(T, H, W, cin, T, cout, wp["Kpad"]), C.c_void_p), cout, 0, 1) == (4 if cout == 96 else 0)   # fused norm: generated for one n tile
    assert L.load().scail_conv3d_kernel_for(C.cast(_geom(1, H, W, cin, 1, cout, wp["Kpad"]), C.c_void_p), cout, 0, 0) == 0      # a single frame
    y = torch.full((T, H, W, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.conv3d_cl(_cl(x), wp, (T, H, W), out=y)
    torch.testing.assert_close(_pl(y), ref, rtol=2e-2, atol=2e-2)
    y2 = ops.conv3d_cl(_cl(x), wp, (T, H, W), resid=_cl(r))
    torch.testing.assert_close(_pl(y2), ref + r, rtol=2e-2, atol=2e-2)
    L.set_option("conv4", 0)
    try:
        y_old = ops.conv3d_cl(_cl(x), wp, (T, H, W))
    finally:
        L.set_option("conv4", 1)
    assert float((y.float() - y_old.float()).abs().max()) <= 2.0 ** -6 * float(ref.abs().max()), "conv4 vs the halo kernel: summation order only"
    if cin != 96:
        return
    # a chunk of 3 output frames whose input carries the 2 cache frames: no padding in front; outputs land in slots 1, 3, 5 of a 7-frame,
    # 128-channel tensor
    To = T - 2
    out = torch.full((2 * To + 1, H, W, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.conv3d_cl(_cl(x), wp, (To, H, W), pad=(0, 1, 1), out=out, ot_mul=2, ot_off=1)
    want = ref[:, 2:]                                              # frames 2.. of the causal result = 'valid' in time
    torch.testing.assert_close(out[1::2, :, :, :cout].float().cpu().permute(3, 0, 1, 2), want, rtol=2e-2, atol=2e-2)
    assert torch.isnan(out[0::2].float()).all() and torch.isnan(out[:, :, :, cout:].float()).all(), "nothing else is written"


@pytest.mark.parametrize("C,T,H,W", [(32, 5, 8, 12), (192, 2, 9, 21), (384, 3, 6, 10)])
def test_resample_convs(C, T, H, W, conv_halo):
    from scail_amd import ops
    g = torch.Generator().manual_seed(1)
    x = bfr(torch.randn(C, T, H, W, generator=g))
    w2 = bfr(torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5)
    b2 = torch.randn(C, generator=g)
    xf = x.permute(1, 0, 2, 3)                                           # (T,C,H,W)
    # downsample2d: ZeroPad2d((0,1,0,1)) + stride-2 conv (wan_vae.py:87-90)
    ref = F.conv2d(F.pad(xf, (0, 1, 0, 1)), w2, b2, stride=2).permute(1, 0, 2, 3)
    y = ops.conv3d_cl(_cl(x), ops.prep_conv_weight(w2.to(DEV), b2.to(DEV)), (T, ref.shape[2], ref.shape[3]), stride=(1, 2, 2), pad=(0, 0, 0))
    torch.testing.assert_close(_pl(y), ref, rtol=2e-2, atol=2e-2)
    # upsample2d: nearest-exact x2 + conv pad 1 (wan_vae.py:76-79)
    wu = bfr(torch.randn(C // 2, C, 3, 3, generator=g) / (9 * C) ** 0.5)
    bu = torch.randn(C // 2, generator=g)
    refu = F.conv2d(F.interpolate(xf, scale_factor=(2.0, 2.0), mode="nearest-exact"), wu, bu, padding=1).permute(1, 0, 2, 3)
    yu = ops.conv3d_cl(_cl(x), ops.prep_conv_weight(wu.to(DEV), bu.to(DEV)), (T, 2 * H, 2 * W), pad=(0, 1, 1), ups=True)
    torch.testing.assert_close(_pl(yu), refu, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("name,cin,cout,k,stride,pad,thw", [
    ("stem", 3, 96, (3, 3, 3), (1, 1, 1), None, (5, 40, 56)),                  # CausalConv3d(3, 96, 3): Cin padded to 8, 14 k-steps
    ("temporal stride 2", 32, 96, (3, 1, 1), (2, 1, 1), (0, 0, 0), (9, 31, 40)),   # a strided, unpadded temporal convolution: 6 k-steps, ragged M, frame offset
    ("shortcut", 96, 192, (1, 1, 1), (1, 1, 1), None, (3, 40, 41)),           # ResidualBlock.shortcut 96 -> 192
    ("pointwise 384", 64, 384, (1, 1, 1), (1, 1, 1), None, (2, 48, 50)),      # 384 output channels: two launches of 192
    ("pointwise 64", 32, 64, (1, 1, 1), (1, 1, 1), None, (4, 33, 35)),
    ("wide K stays on the gather kernel", 192, 384, (1, 1, 1), (1, 1, 1), None, (2, 33, 40))])
def test_conv_direct_gather_kernel(name, cin, cout, k, stride, pad, thw):
    """conv_direct_kernel (csrc/conv.hip; option conv_direct): the HBM-bound convolutions of the VAE -- stem (wan_vae.py:283), downsample3d's
    temporal stride-2 convolution (:143-159), the 1 x 1 x 1 shortcuts (:186-205) -- against torch's fp32 convolution of the bf16-rounded
    operands and against the gather kernel they ran on before (summation order only)."""
    from scail_amd import lib as L, ops
    g = torch.Generator(device=DEV).manual_seed(12)
    T, H, W = thw
    cpad = (cin + 7) // 8 * 8
    x = torch.zeros(T, H, W, cpad, device=DEV, dtype=torch.bfloat16)
    x[..., :cin] = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    fan = cin * k[0] * k[1] * k[2]
    w = (torch.randn(cout, cin, *k, device=DEV, generator=g) / fan ** 0.5).to(torch.bfloat16).float()
    b = torch.randn(cout, device=DEV, generator=g)
    wp = ops.prep_conv_weight(w, b, cin_pad=cpad)
    xf = x[..., :cin].float().permute(3, 0, 1, 2)[None]
    if pad is None:                                        # causal 'same'
        ref = F.conv3d(F.pad(xf, (k[2] // 2, k[2] // 2, k[1] // 2, k[1] // 2, k[0] - 1, 0)), w, b)[0].permute(1, 2, 3, 0)
        To, kw_ = T, {}
    else:
        ref = F.conv3d(xf, w, b, stride=stride)[0].permute(1, 2, 3, 0)
        To, kw_ = ref.shape[0], dict(stride=stride, pad=pad)
    out = torch.full((To + 1, H, W, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.conv3d_cl(x, wp, (To, H, W), out=out, ot_off=1, **kw_)
    assert torch.isnan(out[0].float()).all(), "the frame in front of the offset is not written"
    torch.testing.assert_close(out[1:].float(), ref, rtol=2e-2, atol=2e-2)
    L.set_option("conv_direct", 0)
    try:
        old = ops.conv3d_cl(x, wp, (To, H, W), **kw_)
    finally:
        L.set_option("conv_direct", 1)
    assert float((out[1:].float() - old.float()).abs().max()) <= 2.0 ** -6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cin,cout,thw", [(96, 96, (4, 40, 56)),          # encoder stage 0's widths: 3 slices, one n tile, whole tiles
                                          (32, 96, (3, 17, 35)),          # odd extents: the padding row / column is read, ragged tiles, odd frame count
                                          (192, 192, (5, 34, 50)),        # two n tiles on one XCD, 6 slices
                                          (64, 384, (2, 16, 16)),         # four n tiles, a single voxel tile
                                          (96, 96, (1, 64, 96))])         # one frame: the one-frame kernel whatever the option says
@pytest.mark.parametrize("nf", [1, 2])
def test_conv_s2_downsample_kernel(cin, cout, thw, nf):
    """conv_s2_kernel (csrc/conv.hip; option conv_s2 = output frames per workgroup): Resample's ZeroPad2d((0, 1, 0, 1)) + Conv2d(dim, dim, 3, stride 2) per
    frame (wan_vae.py:87-96) against torch's fp32 convolution of the bf16-rounded operands and against the gather kernel it ran on before (summation
    order only); frame offset / stride of the output honoured, nothing else written."""
    from scail_amd import lib as L, ops
    g = torch.Generator(device=DEV).manual_seed(21)
    T, H, W = thw
    x = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (9 * cin) ** 0.5).to(torch.bfloat16).float()
    b = torch.randn(cout, device=DEV, generator=g)
    wp = ops.prep_conv_weight(w, b)
    ref = F.conv2d(F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1)), w, b, stride=2).permute(0, 2, 3, 1)       # (T, Ho, Wo, cout)
    Ho, Wo = ref.shape[1:3]
    out = torch.full((2 * T + 1, Ho, Wo, cout + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
    L.set_option("conv_s2", nf)
    try:
        ops.conv3d_cl(x, wp, (T, Ho, Wo), stride=(1, 2, 2), pad=(0, 0, 0), out=out, ot_mul=2, ot_off=1)
        L.set_option("conv_s2", 0)
        old = ops.conv3d_cl(x, wp, (T, Ho, Wo), stride=(1, 2, 2), pad=(0, 0, 0))
    finally:
        L.set_option("conv_s2", 1)
    torch.testing.assert_close(out[1::2, :, :, :cout].float(), ref, rtol=2e-2, atol=2e-2)
    assert torch.isnan(out[0::2].float()).all() and torch.isnan(out[..., cout:].float()).all(), "nothing else is written"
    assert float((out[1::2, :, :, :cout].float() - old.float()).abs().max()) <= 2.0 ** -6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cin,thw", [(96, (9, 40, 56)), (192, (7, 128, 224)), (96, (21, 256, 448)), (32, (5, 512, 896))])
def test_conv4c_tile_continuation_is_bit_identical(cin, thw):
    """option conv4_cont (default on): scail_conv4c_e0 / e3 / e4 keep the patch and W rings going across the frame pairs a workgroup walks; the
    arithmetic is that of scail_conv4_e0 / e3 / scail_conv4f_e4 -- bit-identical outputs on shapes with short and long runs per workgroup."""
    from scail_amd import lib as L, ops
    g = torch.Generator(device=DEV).manual_seed(15)
    T, H, W = thw
    x = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    wp = ops.prep_conv_weight(torch.randn(96, cin, 3, 3, 3, device=DEV, generator=g) / (27 * cin) ** 0.5, torch.randn(96, device=DEV, generator=g))
    r = torch.randn(T, H, W, 96, device=DEV, generator=g).to(torch.bfloat16)
    gam = 1 + 0.1 * torch.randn(96, device=DEV, generator=g)
    res = {}
    for mode in (0, 1):
        L.set_option("conv4_cont", mode)
        try:
            res[mode] = (ops.conv3d_cl(x, wp, (T, H, W)), ops.conv3d_cl(x, wp, (T, H, W), resid=r), ops.conv3d_cl_norm(x, wp, gam))
        finally:
            L.set_option("conv4_cont", 1)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    if cin == 96:       # the narrow kernel (RGB head) with and without continuation
        wn = ops.prep_conv_weight(torch.randn(3, cin, 3, 3, 3, device=DEV, generator=g) / (27 * cin) ** 0.5, torch.randn(3, device=DEV, generator=g))
        outs = []
        for mode in (0, 1):
            L.set_option("conv4_cont", mode)
            try:
                outs.append(ops.conv3d_cl(x, wn, (T, H, W)))
            finally:
                L.set_option("conv4_cont", 1)
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("cin,thw", [(96, (5, 33, 40)), (192, (4, 64, 112)), (96, (3, 256, 448))])
def test_conv4f_generated_norm_epilogue(cin, thw):
    """scail_conv4f_e4 behind scail_conv3d_cl_norm (96 output channels): equal to scail_conv3d_cl followed by scail_rms_silu up to the order
    of the norm's sum and the reciprocal (one bf16 step on a few elements), and to the oracle's conv -> RMS_norm -> SiLU on sampled frames."""
    from scail_amd import ops
    g = torch.Generator(device=DEV).manual_seed(6)
    T, H, W = thw
    x = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(96, cin, 3, 3, 3, device=DEV, generator=g) / (27 * cin) ** 0.5).to(torch.bfloat16).float()
    b = torch.randn(96, device=DEV, generator=g)
    gam = 1 + 0.1 * torch.randn(96, device=DEV, generator=g)
    wp = ops.prep_conv_weight(w, b)
    assert ops.conv_norm_generated(wp, x.shape)
    fused = ops.conv3d_cl_norm(x, wp, gam)
    two = ops.rms_silu(ops.conv3d_cl(x, wp, (T, H, W)), gam)
    d = (fused.float() - two.float()).abs()
    assert float(d.max()) <= 2.0 ** -6 * max(1.0, float(two.float().abs().max())) and float(d.mean()) < 2e-4, (float(d.max()), float(d.mean()))
    conv = F.conv3d(F.pad(x.float().permute(3, 0, 1, 2)[None], (1, 1, 1, 1, 2, 0)), w, b)                        # (1, 96, T, H, W) fp32
    ref = F.silu(V.rms_norm(conv, gam.cpu().to(conv.device)))[0].permute(1, 2, 3, 0)
    torch.testing.assert_close(fused.float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("cin,cout,thw", [(96, 3, (5, 40, 56)), (96, 3, (4, 128, 224)), (64, 16, (3, 19, 33)), (32, 12, (2, 16, 16))])
def test_conv4n_generated_narrow_conv(cin, cout, thw):
    """scail_conv4n_e0 (csrc/conv4u.s, asmgen Cfg.nb = 1) behind scail_conv3d_cl: 3x3x3 causal convolutions with at most 16 output channels --
    the decoder head CausalConv3d(96, 3, 3) (reference wan_vae.py:417-419) -- against torch's fp32 convolution of the bf16-rounded operands
    and against the hipcc halo kernel it replaces; the padding channels of the output (3 -> 8) hold zeros like before."""
    import ctypes as C
    from scail_amd import lib as L, ops
    g = torch.Generator(device=DEV).manual_seed(8)
    T, H, W = thw
    x = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=g) / (27 * cin) ** 0.5).to(torch.bfloat16).float()
    b = torch.randn(cout, device=DEV, generator=g)
    wp = ops.prep_conv_weight(w, b)
    N = wp["N"]
    assert N in (8, 16) and L.load().scail_conv3d_kernel_for(C.cast(_geom(T, H, W, cin, T, N, wp["Kpad"]), C.c_void_p), N, 0, 0) == 4
    y = torch.full((T, H, W, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.conv3d_cl(x, wp, (T, H, W), out=y)
    ref = F.conv3d(F.pad(x.float().permute(3, 0, 1, 2)[None], (1, 1, 1, 1, 2, 0)), w, b)[0].permute(1, 2, 3, 0)
    torch.testing.assert_close(y[..., :cout].float(), ref, rtol=2e-2, atol=2e-2)
    assert float(y[..., cout:].float().abs().max() if cout < N else 0.0) == 0.0, "padding channels: zero weights, zero bias"
    L.set_option("conv4", 0)
    try:
        y_old = ops.conv3d_cl(x, wp, (T, H, W))
    finally:
        L.set_option("conv4", 1)
    assert float((y.float() - y_old.float()).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())


@pytest.mark.parametrize("cin,cout,thw,ups", [(192, 96, (3, 24, 40), True),       # decoder stage 11's widths: 6 slices, one n tile, odd frame count
                                              (384, 192, (4, 17, 23), True),      # stage 3 / 7's widths: 12 slices, 2 n tiles, ragged 34 x 46 output
                                              (96, 96, (2, 32, 48), False),       # the same kernel without the upsample (a per-frame 3 x 3 convolution)
                                              (192, 96, (5, 128, 224), True)])    # 256 x 448 output: 224 spatial tiles x 3 frame pairs, runs of tiles per workgroup
def test_conv4u_generated_upsample_conv(cin, cout, thw, ups):
    """scail_conv4u_e0 (csrc/conv4u.s, asmgen Cfg.kt = 1) behind scail_conv3d_cl: Resample's nearest-exact 2x upsample + Conv2d(dim, dim / 2, 3,
    padding 1) per frame (reference wan_vae.py:76-85) against torch's fp32 convolution of the bf16-rounded operands on the GPU, against the
    hipcc halo kernel it replaces (summation order only), into interleaved output frames of a wider tensor (ot_mul / ot_off, ldc > N)."""
    import ctypes as C
    from scail_amd import lib as L, ops
    g = torch.Generator(device=DEV).manual_seed(3)
    T, H, W = thw
    x = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (9 * cin) ** 0.5).to(torch.bfloat16).float()
    b = torch.randn(cout, device=DEV, generator=g)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    wp = ops.prep_conv_weight(w, b)
    geom = (C.c_int32 * 21)(T, H, W, cin, T, Ho, Wo, 1, 3, 3, 1, 1, 1, 0, 1, 1, 1 if ups else 0, 1, 0, wp["N"], wp["Kpad"])
    assert L.load().scail_conv3d_kernel_for(C.cast(geom, C.c_void_p), cout, 0, 0) == 4
    y = torch.full((T, Ho, Wo, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.conv3d_cl(x, wp, (T, Ho, Wo), pad=(0, 1, 1), ups=ups, out=y)
    xf = x.float().permute(0, 3, 1, 2)
    if ups:
        xf = F.interpolate(xf, scale_factor=(2.0, 2.0), mode="nearest-exact")
    ref = F.conv2d(xf, w, b, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=2e-2)
    L.set_option("conv4", 0)
    try:
        y_old = ops.conv3d_cl(x, wp, (T, Ho, Wo), pad=(0, 1, 1), ups=ups)
    finally:
        L.set_option("conv4", 1)
    assert float((y.float() - y_old.float()).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())
    out = torch.full((2 * T + 1, Ho, Wo, cout + 32), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.conv3d_cl(x, wp, (T, Ho, Wo), pad=(0, 1, 1), ups=ups, out=out, ot_mul=2, ot_off=1)
    assert torch.equal(out[1::2, :, :, :cout], y)
    assert torch.isnan(out[0::2].float()).all() and torch.isnan(out[:, :, :, cout:].float()).all(), "nothing else is written"


@pytest.mark.parametrize("cin,cout,thw", [(96, 96, (5, 10, 12)), (32, 32, (4, 9, 17)), (64, 64, (2, 8, 16)), (96, 96, (1, 16, 16)),
                                          (192, 96, (3, 11, 21)), (32, 40, (3, 8, 16))])
def test_conv_with_fused_rms_silu(cin, cout, thw):
    """scail_conv3d_cl_norm == scail_conv3d_cl followed by scail_rms_silu (the raw conv output just never goes to HBM), and
    matches the oracle's conv -> RMS_norm -> SiLU."""
    from scail_amd import ops
    g = torch.Generator().manual_seed(5)
    T, H, W = thw
    x = bfr(torch.randn(cin, T, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5)
    b = torch.randn(cout, generator=g)
    gam = 1 + 0.1 * torch.randn(cout, generator=g)
    wp = ops.prep_conv_weight(w.to(DEV), b.to(DEV))
    assert ops.conv_norm_fusable(wp, cin)
    fused = ops.conv3d_cl_norm(_cl(x), wp, gam.to(DEV))
    two = ops.rms_silu(ops.conv3d_cl(_cl(x), wp, (T, H, W)), gam.to(DEV))
    torch.testing.assert_close(fused.float(), two.float(), rtol=1e-2, atol=1e-2)        # summation order of the norm only
    assert float((fused.float() - two.float()).abs().mean()) < 1e-4
    ref = F.silu(V.rms_norm(V.causal_conv3d(x[None], w, b), gam)[0])
    torch.testing.assert_close(_pl(fused)[:cout], ref, rtol=2e-2, atol=2e-2)
    big = ops.prep_conv_weight(bfr(torch.randn(128, cin, 3, 3, 3, generator=g)).to(DEV), torch.zeros(128, device=DEV))
    assert not ops.conv_norm_fusable(big, cin)
    from scail_amd import lib as L
    with pytest.raises(L.ScailHipError, match="N <= 96"):
        ops.conv3d_cl_norm(_cl(x), big, torch.ones(128, device=DEV))


def test_rms_silu_softmax_transpose():
    from scail_amd import ops
    g = torch.Generator().manual_seed(2)
    for C in (32, 96, 192, 384):
        x = bfr(torch.randn(77, C, generator=g) * 2)
        gam = 1 + 0.1 * torch.randn(C, generator=g)
        ref = F.silu(V.rms_norm(x.t()[None], gam.view(-1, 1))[0].t())
        y = ops.rms_silu(x.to(torch.bfloat16).to(DEV), gam.to(DEV))
        torch.testing.assert_close(y.float().cpu(), ref, rtol=2e-2, atol=2e-2)
    s = bfr(torch.randn(40, 128, generator=g) * 3)
    sg = s.to(torch.bfloat16).to(DEV)
    ops.softmax_rows_(sg, 120, 0.5)
    torch.testing.assert_close(sg[:, :120].float().cpu(), torch.softmax(0.5 * s[:, :120], -1), rtol=2e-2, atol=2e-3)
    assert torch.equal(sg[:, 120:].float().cpu(), s[:, 120:])
    # ragged n (columns n .. ceil8(n) come back as zeros) and rows longer than 8192
    for n, ld in ((15, 64), (123, 128), (9001, 9024), (20000, 20032)):
        s = bfr(torch.randn(3, ld, generator=g) * 2)
        sg = s.to(torch.bfloat16).to(DEV)
        ops.softmax_rows_(sg, n, 0.7)
        got = sg.float().cpu()
        torch.testing.assert_close(got[:, :n], torch.softmax(0.7 * s[:, :n], -1), rtol=2e-2, atol=2e-3)
        assert float(got[:, n:(n + 7) // 8 * 8].abs().max() if n % 8 else 0.0) == 0.0
        torch.testing.assert_close(got[:, (n + 7) // 8 * 8:], s[:, (n + 7) // 8 * 8:])          # beyond ceil8(n): untouched
    t = bfr(torch.randn(2, 70, 96, generator=g))
    out = torch.zeros(2, 96, 128, device=DEV, dtype=torch.bfloat16)
    ops.transpose2d(t.to(torch.bfloat16).to(DEV), out)
    assert torch.equal(out[:, :, :70].float().cpu(), t.transpose(1, 2)) and float(out[:, :, 70:].abs().max()) == 0


def _load(golden_dir, name="vae_tiny.npz"):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _model(g):
    from scail_amd.wan_vae import WanVAE_
    cfg = V.VAEConfig(dim=int(g["dim"]), z_dim=16)
    sd = V.make_state_dict(cfg, seed=int(g["seed"]))
    m = WanVAE_(dim=cfg.dim, z_dim=16, device=DEV)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return cfg, sd, m


@pytest.mark.parametrize("name", ["vae_tiny.npz", "vae_tiny2.npz", "vae_dim96.npz"])
def test_vae_encode_vs_reference_golden(golden_dir, name):
    g = _load(golden_dir, name)
    cfg, sd, m = _model(g)
    mu = m.encode(g["video"].to(DEV)).cpu()
    assert mu.shape == g["mu"].shape
    torch.testing.assert_close(mu, g["mu"], rtol=3e-2, atol=3e-2)
    assert _cos(mu, g["mu"]) >= 0.999
    mu1 = m.encode(g["video"][:, :, :1].to(DEV)).cpu()                   # single image
    torch.testing.assert_close(mu1, g["mu1"], rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("name", ["vae_tiny.npz", "vae_tiny2.npz", "vae_dim96.npz"])
def test_vae_decode_vs_reference_golden(golden_dir, name):
    g = _load(golden_dir, name)
    cfg, sd, m = _model(g)
    rec = m.decode(g["z_in"].to(DEV)).clamp(-1, 1).cpu()
    assert rec.shape == g["rec"].shape
    torch.testing.assert_close(rec, g["rec"], rtol=3e-2, atol=3e-2)
    assert _cos(rec, g["rec"]) >= 0.999


def test_vae_wrapper_and_causality():
    """WanVAE wrapper API + a size-independent property: temporal causality -- the first k latent frames of
    a long clip equal the latents of the clip truncated to 1 + 4(k-1) frames (what makes the reference's
    chunked streaming and this whole-sequence pass interchangeable)."""
    from scail_amd.wan_vae import WanVAE
    vae = WanVAE(z_dim=16, vae_pth=None, dtype="torch.bfloat16", device=DEV, dim=32)
    g = torch.Generator().manual_seed(3)
    vid = (torch.rand(3, 13, 32, 32, generator=g) * 2 - 1).to(DEV)
    z = vae.encode([vid])
    assert z.shape == (1, 16, 4, 4, 4) and z.dtype == torch.float32
    z5 = vae.encode([vid[:, :5]])
    torch.testing.assert_close(z[:, :, :2], z5, rtol=1e-3, atol=1e-3)
    x = vae.decode([z[0]])
    assert x.shape == (1, 3, 13, 32, 32) and float(x.abs().max()) <= 1.0


def test_c_executor_equals_layerwise_path():
    """include/scail_vae.h: encode / decode as one C call each enqueue the same kernels in the same order as the layer-by-layer
    Python path -> bit-identical results (the golden tests above run through the C executor, the default)."""
    from scail_amd.wan_vae import WanVAE_
    m = WanVAE_(dim=32, z_dim=16, device=DEV)
    g = torch.Generator().manual_seed(9)
    vid = (torch.rand(1, 3, 9, 48, 64, generator=g) * 2 - 1).to(DEV)       # mid-block attention: 6 x 8 = 48 tokens per frame
    assert m.use_c_exec
    z_c = m.encode(vid)
    x_c = m.decode(z_c)
    assert m._cvae is not None
    m.use_c_exec = False
    z_p = m.encode(vid)
    x_p = m.decode(z_p)
    assert z_c.shape == (1, 16, 3, 6, 8) and x_c.shape == (1, 3, 9, 48, 64)
    assert torch.equal(z_c, z_p) and torch.equal(x_c, x_p)
    with pytest.raises(ValueError):
        m.use_c_exec = True
        m.encode(vid[:, :, :8])
    # the shipped width: 384 attention channels, where 1 / sqrt(C) rounds differently in float and in double (the C executor computed it
    # in float until round 4 and so differed from the layer path in the last bit -- found at 81 x 512 x 896 by tools/vae_exec_vs_layers.py);
    # the launch-by-launch trace of include/scail_vae.h sees the same tensors as the layer path's launches
    m96 = WanVAE_(dim=96, z_dim=16, device=DEV)
    v96 = (torch.rand(1, 3, 9, 32, 48, generator=g) * 2 - 1).to(DEV)
    seen = []
    m96.prepare()
    m96._c().set_trace(lambda i, op, t: seen.append((i, op, tuple(t.shape), float(t.float().abs().sum()))))
    z96 = m96.encode(v96)
    n_enc = len(seen)
    x96 = m96.decode(z96)
    m96._c().set_trace(None)
    m96.decode(z96)
    assert [s_[0] for s_ in seen[:n_enc]] == list(range(n_enc)) and len(seen) > n_enc + 30, "trace indices count the launches of one call"
    assert {s_[1] for s_ in seen} - {"conv_resid_norm"} in ({"conv", "rms_silu", "attn", "conv_norm"}, {"conv", "rms_silu", "attn"})
    assert seen[0][2] == (9, 32, 48, 96) and seen[n_enc - 1][2] == (3, 4, 6, 32) and all(math.isfinite(s_[3]) for s_ in seen)
    m96.use_c_exec = False
    assert torch.equal(z96, m96.encode(v96)) and torch.equal(x96, m96.decode(z96))
    # a single large frame: the attention score matrix (4096 x 4096), not an activation, sizes the workspace slots
    img = (torch.rand(1, 3, 1, 512, 512, generator=g) * 2 - 1).to(DEV)
    z1 = m.encode(img)
    m.use_c_exec = False
    assert torch.equal(z1, m.encode(img))


@pytest.mark.parametrize("thw,want_raw", [((9, 512, 896), True), ((9, 512, 896), False), ((5, 48, 80), True), ((3, 50, 70), False)])
def test_conv3d_cl_resid_norm_against_the_two_calls(thw, want_raw):
    """scail_conv3d_cl_resid_norm (round 6; include/scail_hip.h): the last convolution of a ResidualBlock (wan_vae.py:180-218) with the next consumer's
    RMS_norm + SiLU (:39-54) in its epilogue -- scail_conv4c_e5 (raw sum + normalised copy) / e6 (normalised only) at the full-resolution 96-channel
    shape of config 4 and at ragged small ones.  Against the two calls it replaces: the raw sum bit for bit, the normalised tensor up to the order of
    the norm's sum of squares (1 bf16 ulp of the larger magnitude); option "conv4_resnorm" = 0 runs exactly the two calls."""
    from scail_amd import lib as L, ops
    T, H, W = thw
    C = 96
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.randn(T, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(C, C, 3, 3, 3, device=DEV, generator=g) / (27 * C) ** 0.5)
    b = torch.randn(C, device=DEV, generator=g)
    r = torch.randn(T, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
    gam = (1 + 0.1 * torch.randn(C, device=DEV, generator=g)).float()
    wp = ops.prep_conv_weight(w, b)
    assert ops.conv_resid_norm_generated(wp, x.shape)
    raw, nrm = ops.conv3d_cl_resid_norm(x, wp, r, gam, want_raw=want_raw)
    y = ops.conv3d_cl(x, wp, (T, H, W), resid=r)
    n2 = ops.rms_silu(y, gam)
    assert (raw is None) == (not want_raw)
    if want_raw:
        assert torch.equal(raw, y)
    assert torch.isfinite(nrm.float()).all()
    d = (nrm.float() - n2.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * max(1.0, float(n2.float().abs().max())), float(d.max())
    assert float((d > 0).float().mean()) < 0.02          # the two sums of squares round differently for a few voxels only
    L.set_option("conv4_resnorm", 0)
    try:
        assert not ops.conv_resid_norm_generated(wp, x.shape)
        raw0, nrm0 = ops.conv3d_cl_resid_norm(x, wp, r, gam, want_raw=want_raw)
    finally:
        L.set_option("conv4_resnorm", 1)
    assert torch.equal(nrm0, n2) and (raw0 is None or torch.equal(raw0, y))


@pytest.mark.parametrize("thw", [(9, 512, 896), (5, 40, 56)])
def test_conv3d_cl_resid_norm_on_the_stem_convolution(thw):
    """the encoder's stem (CausalConv3d(3, 96, 3), wan_vae.py:283; 3 channels padded to 8) with the first ResidualBlock's input norm: the hipcc
    direct-gather kernel's dual-output form (conv_direct_kernel<14, 3, true>; scail_conv3d_kernel_for(.., 2) == 2) against the two calls."""
    from scail_amd import lib as L, ops
    import ctypes as C
    T, H, W = thw
    g = torch.Generator(device=DEV).manual_seed(19)
    x = torch.zeros(T, H, W, 8, device=DEV, dtype=torch.bfloat16)
    x[..., :3] = torch.randn(T, H, W, 3, device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randn(96, 3, 3, 3, 3, device=DEV, generator=g) / 81 ** 0.5
    b = torch.randn(96, device=DEV, generator=g)
    gam = (1 + 0.1 * torch.randn(96, device=DEV, generator=g)).float()
    wp = ops.prep_conv_weight(w, b, cin_pad=8)
    geom = ops._next_norm_geom(wp, x.shape, (T, H, W), None, False)
    assert L.load().scail_conv3d_kernel_for(C.cast(geom, C.c_void_p), 96, 0, 2) == 2
    raw, nrm = ops.conv3d_cl_resid_norm(x, wp, None, gam)
    y = ops.conv3d_cl(x, wp, (T, H, W))
    n2 = ops.rms_silu(y, gam)
    assert torch.equal(raw, y) and torch.isfinite(nrm.float()).all()
    d = (nrm.float() - n2.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * max(1.0, float(n2.float().abs().max())), float(d.max())


@pytest.mark.parametrize("thw", [(9, 256, 448), (3, 24, 40)])
def test_conv3d_cl_resid_norm_without_residual_on_the_upsample_convolution(thw):
    """scail_conv3d_cl_resid_norm with resid = NULL on Resample's 1 x 3 x 3 convolution behind the nearest 2x upsample (wan_vae.py:76-85; 192 -> 96
    channels, full resolution out): scail_conv4u_e7 writes the raw output (bit-identical to scail_conv4u_e0's) and the next ResidualBlock's
    normalised input."""
    from scail_amd import lib as L, ops
    T, H, W = thw
    Cin, C = 192, 96
    g = torch.Generator(device=DEV).manual_seed(17)
    x = torch.randn(T, H, W, Cin, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(C, Cin, 1, 3, 3, device=DEV, generator=g) / (9 * Cin) ** 0.5)
    b = torch.randn(C, device=DEV, generator=g)
    gam = (1 + 0.1 * torch.randn(C, device=DEV, generator=g)).float()
    wp = ops.prep_conv_weight(w, b)
    out_shape = (T, 2 * H, 2 * W)
    assert ops.conv_resid_norm_generated(wp, x.shape, out_shape, pad=(0, 1, 1), ups=True, resid=False)
    raw, nrm = ops.conv3d_cl_resid_norm(x, wp, None, gam, out_shape=out_shape, pad=(0, 1, 1), ups=True)
    y = ops.conv3d_cl(x, wp, out_shape, pad=(0, 1, 1), ups=True)
    n2 = ops.rms_silu(y, gam)
    assert torch.equal(raw, y) and torch.isfinite(nrm.float()).all()
    d = (nrm.float() - n2.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * max(1.0, float(n2.float().abs().max())), float(d.max())
    L.set_option("conv4_resnorm", 0)
    try:
        raw0, nrm0 = ops.conv3d_cl_resid_norm(x, wp, None, gam, out_shape=out_shape, pad=(0, 1, 1), ups=True)
    finally:
        L.set_option("conv4_resnorm", 1)
    assert torch.equal(nrm0, n2) and torch.equal(raw0, y)


@pytest.mark.parametrize("C,thw", [(96, (9, 512, 896)), (96, (5, 720, 1280)), (192, (9, 256, 448)), (384, (7, 128, 224))])
def test_conv4_at_vae_resolutions(C, thw):
    """the generated convolution kernels at the sizes BASELINE config 4 runs them (512 x 896 full / half / quarter resolution; 720 x 1280: 3 600
    spatial tiles, 32-bit offsets up to 177 MB per frame), odd frame counts: sampled output voxels (corners, tile seams, interior, first / last
    frame) against an fp32 convolution of the 3 x 3 x 3 neighbourhood, plain and with the residual; the whole tensor against the hipcc halo
    kernel (summation order only)."""
    from scail_amd import lib as L, ops
    T, H, W = thw
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(T, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(C, C, 3, 3, 3, device=DEV, generator=g) / (27 * C) ** 0.5)
    b = torch.randn(C, device=DEV, generator=g)
    r = torch.randn(T, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
    wp = ops.prep_conv_weight(w, b)
    assert ops.conv_generated(wp, x.shape)
    y = ops.conv3d_cl(x, wp, (T, H, W))
    y2 = ops.conv3d_cl(x, wp, (T, H, W), resid=r)
    L.set_option("conv4", 0)
    try:
        y_old = ops.conv3d_cl(x, wp, (T, H, W))
    finally:
        L.set_option("conv4", 1)
    assert torch.isfinite(y.float()).all() and torch.isfinite(y2.float()).all()
    scale = float(y_old.float().abs().max())
    assert float((y.float() - y_old.float()).abs().max()) <= 2.0 ** -6 * scale
    assert float((y2.float() - (y_old.float() + r.float())).abs().max()) <= 2.0 ** -5 * scale + 2.0 ** -7 * float(r.float().abs().max())
    # sampled voxels against fp32 (bf16-rounded operands): frames / rows / columns at the tensor's edges, across tile seams, inside
    wr = w.to(torch.bfloat16).float()
    ts = sorted({0, 1, T // 2, T - 1})
    hs = sorted({0, 1, 15, 16, 17, H // 2, H - 17, H - 16, H - 1})
    ws_ = sorted({0, 1, 15, 16, 31, 32, W // 2 + 5, W - 16, W - 1})
    xp = torch.zeros(T + 2, H + 2, W + 2, C, device=DEV)
    xp[2:, 1:-1, 1:-1] = x.float()
    for t in ts:
        for h in hs:
            patch = xp[t:t + 3, h:h + 3][:, :, [c for wv in ws_ for c in (wv, wv + 1, wv + 2)]]          # (3, 3, 3 * n, C)
            patch = patch.view(3, 3, len(ws_), 3, C).permute(2, 0, 1, 3, 4)                              # (n, dt, dh, dw, C)
            ref = torch.einsum("vtabc,nctab->vn", patch, wr) + b
            got = y[t, h, ws_].float()
            torch.testing.assert_close(got, ref, rtol=2e-2, atol=2e-2)


def _tile_stats(err, th, tw):
    """err (..., H, W) -> per-(th x tw)-tile max and mean over the trailing two axes (ragged edges ignored: H, W are multiples here)."""
    *lead, H, W = err.shape
    t = err.reshape(*lead, H // th, th, W // tw, tw)
    return t.amax(dim=(-3, -1)), t.mean(dim=(-3, -1))


@pytest.mark.parametrize("direction", ["encode", "decode"])
def test_vae_fullsize_vs_fp32(direction):
    """BASELINE config 4 at its OWN size through the C executor (scail_vae_encode / scail_vae_decode: video (1,3,81,512,896), latent
    (1,16,21,64,112)) against oracle/wan_vae_oracle.py evaluated in fp32 ON THE GPU (CONV_IMPL = "taps": every convolution as fp32 matmuls
    per tap, pinned to the real chunked reference by tests/test_vae_oracle_golden.py): the whole 26 / 33-convolution chain with its temporal
    rules, the stride-2 / upsample convolutions, whole-sequence (unchunked) execution and the 7 168-token mid attention
    (reference wan_vae.py:516-568).
    Criterion (bf16 activations through ~30 layers vs fp32; same form as the DiT's full-size test): cosine >= 0.999; at most 1e-4 of the
    elements beyond rtol / atol 3e-2 and none beyond 4x that bound; and PER TILE (16 x 16 output pixels per frame for the decoder -- the
    convolution kernels' own tile; 4 x 4 latent pixels = 32 x 32 video pixels per frame for the encoder) the worst error inside the same 4x bound
    and the tile's mean error <= 5x the global mean error + 1e-3, so a wrong halo / seam / frame-slot cannot hide in the statistics."""
    from scail_amd.wan_vae import WanVAE_
    cfg = V.VAEConfig(dim=96, z_dim=16)
    sd = V.make_state_dict(cfg, seed=4321)
    m = WanVAE_(dim=96, z_dim=16, device=DEV)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected and m.use_c_exec
    g = torch.Generator(device=DEV).manual_seed(5)
    if direction == "encode":
        inp = (torch.rand(1, 3, 81, 512, 896, device=DEV, generator=g) * 2 - 1).to(torch.bfloat16).float()
        got = m.encode(inp)
    else:
        inp = torch.randn(1, 16, 21, 64, 112, device=DEV, generator=g).to(torch.bfloat16).float()
        got = m.decode(inp)
    torch.cuda.synchronize()
    assert m._cvae is not None, "the C executor (include/scail_vae.h) must have run"
    m._cvae = None                                       # release the executor's arena before the fp32 oracle allocates
    torch.cuda.empty_cache()
    sdg = {k: v.to(DEV) for k, v in sd.items()}
    old = V.CONV_IMPL
    V.CONV_IMPL = "taps"
    try:
        with torch.no_grad():
            want = V.encode(cfg, sdg, inp) if direction == "encode" else V.decode(cfg, sdg, inp, clamp=False)
    finally:
        V.CONV_IMPL = old
    assert got.shape == want.shape == ((1, 16, 21, 64, 112) if direction == "encode" else (1, 3, 81, 512, 896))
    assert torch.isfinite(got).all()
    tol, frac, hard = 3e-2, 1e-4, 4.0
    err = (got - want).abs()
    lim = tol + tol * want.abs()
    rel = err / lim
    n_bad = int((rel > 1).sum())
    worst = float(rel.max())
    gmean = float(err.mean())
    cos = _cos(got, want)
    th, tw = (4, 4) if direction == "encode" else (16, 16)
    tmax, tmean = _tile_stats(rel, th, tw)[0], _tile_stats(err, th, tw)[1]
    print(f"VAE {direction} 81x512x896 vs fp32 oracle: cosine {cos:.6f}, {n_bad} of {err.numel()} beyond rtol/atol {tol} ({n_bad / err.numel():.2e}), "
          f"worst {worst:.2f}x the bound, max abs err {float(err.max()):.3e}, mean {gmean:.3e}, |ref| max {float(want.abs().max()):.2f} rms "
          f"{float(want.pow(2).mean().sqrt()):.3f}; worst tile: max {float(tmax.max()):.2f}x, mean err {float(tmean.max()):.3e}")
    assert cos >= 0.999
    assert n_bad <= frac * err.numel(), f"{n_bad} elements beyond tolerance"
    assert worst <= hard
    assert float(tmax.max()) <= hard
    assert float(tmean.max()) <= 5.0 * gmean + 1e-3, "a tile whose mean error stands out: wrong halo / seam / frame slot"
