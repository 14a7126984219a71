"""CPU emulation (tools/asm_emu.py) of the generated 3x3x3 causal convolution kernels (scail_amd/asmgen/conv4.py, csrc/conv4.s): the
generator produces hazard-free code whose results equal the fp64 convolution of the bf16-rounded operands (reference CausalConv3d,
sgm/models/wan_vae.py:17-36), under the emulator's lazy (latest-allowed) completion of LDS / memory operations -- the mode that exposes
a missing or too-weak s_waitcnt.  Covers the frame-slot ring (more slices than slots), ragged tiles, an odd frame count, two n tiles,
several tiles per persistent workgroup, the residual epilogue, padded output rows and the chunked-decode frame mapping."""
import numpy as np
import pytest

from scail_amd.asmgen import conv4
from tools import conv4_emu_run as R


def _case(Ti, H, W, Cin, N, resid, seed=0, frames=None):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((Ti, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((N, Cin, 3, 3, 3)) / np.sqrt(27 * Cin)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((frames or Ti, H, W, N)).astype(np.float32) if resid else None
    return x, w, b, r


def _cfg(name):
    return [c for c in conv4.DEFAULTS + conv4.variant_cfgs() if c.name == name][0]


def test_conv4_static_hazards():
    for cfg in conv4.DEFAULTS + conv4.UPSAMPLE + conv4.NARROW + conv4.FUSED + conv4.CONT + conv4.RESNORM:
        assert R.check_static(cfg) == [], cfg.name


@pytest.mark.parametrize("shape,cus,ldc", [((2, 16, 16, 96, 8), 256, None),        # the RGB head's widths (3 channels padded to 8): lanes 32..63 store nothing
                                           ((3, 18, 20, 64, 16), 8, None),         # 16 channels, ragged tiles, odd frame count, several tiles per workgroup
                                           ((5, 16, 40, 32, 8), 8, 16)])           # output rows wider than N
def test_conv4n_emulated(shape, cus, ldc):
    """scail_conv4n_e0 (Cfg.nb = 1): 3x3x3 causal convolution with N <= 16 output channels (the decoder head CausalConv3d(96, 3, 3),
    reference wan_vae.py:417-419): one 16-channel block per tile, range-checked bias / W rows, 8-byte stores from the accumulator layout."""
    cfg = conv4.NARROW[0]
    Ti, H, W, Cin, N = shape
    x, w, b, _ = _case(Ti, H, W, Cin, N, False, seed=2)
    y, _ = R.run(cfg, x, w, b, None, cus=cus, ldc=ldc)
    ref = R.reference(x, w, b, None)
    assert not np.isnan(y[..., :N]).any(), "every output voxel is written"
    assert np.isnan(y[..., N:]).all(), "nothing past the N channels is written"
    assert np.abs(y[..., :N] - ref).max() <= 2.0 ** -7 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("ups,shape,cus", [(1, (2, 8, 8, 224, 96), 256),      # 7 slices of 9 taps: the two slot pairs alternate
                                           (1, (3, 9, 10, 64, 192), 8),       # odd frame count, ragged 18 x 20 output, 2 n tiles, several tiles per workgroup
                                           (0, (5, 16, 40, 32, 96), 8),       # no upsample: a plain per-frame 3 x 3 convolution, 9 tiles on 8 workgroups
                                           (1, (4, 8, 24, 96, 96), 8)])       # runs of tiles within a spatial tile and across
def test_conv4u_emulated(ups, shape, cus):
    """scail_conv4u_e0 (Cfg.kt = 1): the 3 x 3 convolution of Resample behind the nearest 2x upsample (reference wan_vae.py:76-85), folded into
    the patch gather -- patch voxel (h, w) reads input (h >> 1, w >> 1)."""
    cfg = conv4.UPSAMPLE[0]
    T, Hi, Wi, Cin, N = shape
    rng = np.random.default_rng(5)
    x = rng.standard_normal((T, Hi, Wi, Cin)).astype(np.float32)
    w = (rng.standard_normal((N, Cin, 1, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    y, _ = R.run_k1(cfg, x, w, b, ups, cus=cus)
    ref = R.reference_k1(x, w, b, ups)
    assert not np.isnan(y).any(), "every output voxel is written"
    assert np.abs(y - ref).max() <= 2.0 ** -7 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name,shape,cus", [("scail_conv4_e0", (2, 16, 16, 32, 96), 256),         # one tile, one slice
                                            ("scail_conv4_e3", (2, 16, 16, 224, 96), 256),        # 7 slices: the 5-slot frame ring wraps
                                            ("scail_conv4_e3", (3, 18, 20, 64, 192), 8),          # ragged tiles, odd frame count, 2 n tiles, 2 tiles per workgroup
                                            ("scail_conv4_e0", (5, 16, 40, 32, 96), 8),           # 9 tiles on 8 workgroups: one walks two, a frame pair past the end
                                            ("scail_conv4_e0", (5, 16, 96, 32, 96), 8)])          # 18 tiles: runs of 2-3 tiles, within a spatial tile (offsets kept) and across
def test_conv4_emulated(name, shape, cus):
    cfg = _cfg(name)
    x, w, b, r = _case(*shape, resid=cfg.epi == 3)
    y, _ = R.run(cfg, x, w, b, r, cus=cus)
    ref = R.reference(x, w, b, r)
    err = np.abs(y - ref)
    assert not np.isnan(y).any(), "every output voxel is written"
    assert err.max() <= 2.0 ** -7 * max(1.0, np.abs(ref).max()), float(err.max())     # one bf16 rounding of the output


def test_conv4_emulated_row_stride_frame_mapping_no_bias():
    """ldc > N (the output is a channel slice of a wider tensor), output frame t -> slot 2 t + 1 (the interleaving of the decoder's temporal
    upsampling), pt = 0 with two extra input frames in front (a chunk with its cache frames), bias == NULL."""
    cfg = _cfg("scail_conv4_e0")
    Ti, H, W, Cin, N = 4, 16, 16, 32, 96
    x, w, _, _ = _case(Ti, H, W, Cin, N, False, seed=3)
    To = Ti - 2
    y, _ = R.run(cfg, x, w, None, None, pt=0, To=To, ot_mul=2, ot_off=1, y_frames=2 * To + 1, ldc=128)
    ref = R.reference(x, w, None, None, pt=0, To=To)
    assert np.isnan(y[0::2]).all() and np.isnan(y[:, :, :, N:]).all(), "nothing outside the addressed frames / channels is written"
    err = np.abs(y[1::2, :, :, :N] - ref)
    assert err.max() <= 2.0 ** -7 * max(1.0, np.abs(ref).max()), float(err.max())


@pytest.mark.parametrize("shape,cus", [((2, 16, 16, 32, 96), 256), ((3, 18, 20, 32, 96), 8)])
def test_conv4f_emulated_norm_epilogue(shape, cus):
    """scail_conv4f_e4 (Cfg.epi = 4): conv -> RMS_norm -> SiLU in the epilogue (ResidualBlock.residual[2..4], reference wan_vae.py:190-196 with
    RMS_norm :39-54): the sum of squares over a voxel's 96 channels = 24 in-lane terms + two permlane swaps, applied to the bf16-rounded
    convolution output like the separate rms_silu pass."""
    cfg = conv4.FUSED[0]
    Ti, H, W, Cin, N = shape
    x, w, b, _ = _case(Ti, H, W, Cin, N, False, seed=4)
    gam = (1 + 0.1 * np.random.default_rng(9).standard_normal(N)).astype(np.float32)
    y, _ = R.run(cfg, x, w, b, None, cus=cus, gamma=gam)
    ref = R.reference_norm_silu(R.reference(x, w, b, None), gam)
    assert not np.isnan(y).any()
    err = np.abs(y - ref)
    assert err.max() <= 2.0 ** -6 * max(1.0, np.abs(ref).max()) and err.mean() <= 2e-3, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("name,shape", [("scail_conv4c_e0", (21, 16, 16, 32, 96)),      # 11 frame pairs of ONE spatial tile on 8 workgroups: runs of 1-2 tiles, one slice per tile
                                        ("scail_conv4c_e3", (5, 18, 20, 64, 96)),       # two slices, residual; runs that cross into the next spatial tile
                                        ("scail_conv4c_e4", (9, 16, 32, 32, 96)),
                                        ("scail_conv4cn_e0", (13, 16, 32, 32, 8))])     # the narrow kernel with continuing rings (no staging strip to move)
def test_conv4c_emulated_tile_continuation(name, shape):
    """Cfg.cont: when a workgroup's next tile is the next frame pair of the same spatial tile, the last slice prefetches that tile's first
    frames and W taps and the rings continue (no first loads, the epilogue's staging strip in the slot of the dead frame 3); otherwise the
    rings restart.  Same results as the shipped kernels."""
    cfg = [c for c in conv4.CONT if c.name == name][0]
    Ti, H, W, Cin, N = shape
    x, w, b, r = _case(Ti, H, W, Cin, N, cfg.epi == 3, seed=6)
    gam = (1 + 0.1 * np.random.default_rng(3).standard_normal(N)).astype(np.float32) if cfg.epi == 4 else None
    y, _ = R.run(cfg, x, w, b, r, cus=8, gamma=gam)
    ref = R.reference(x, w, b, r)
    if cfg.epi == 4:
        ref = R.reference_norm_silu(ref, gam)
    assert not np.isnan(y[..., :N]).any()
    assert np.abs(y[..., :N] - ref).max() <= 2.0 ** -6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name,shape", [("scail_conv4c_e5", (5, 18, 20, 64, 96)),      # two slices; runs that cross into the next spatial tile (rings restart / continue)
                                        ("scail_conv4c_e5", (9, 16, 32, 32, 96)),
                                        ("scail_conv4c_e6", (5, 18, 20, 64, 96)),
                                        ("scail_conv4c_e6", (21, 16, 16, 32, 96))])
def test_conv4c_emulated_residual_plus_next_norm(name, shape):
    """Cfg.epi 5 / 6 (round 6): the LAST convolution of a ResidualBlock (wan_vae.py:180-218: conv + shortcut) with the RMS_norm + SiLU of whatever
    reads the block's output next (the next block's residual[0..1], or the decoder head :417) in its epilogue.  epi 5 stores the raw sum (the
    next block's shortcut operand) AND the normalised copy at y + (y2 - y); epi 6 the normalised tensor only.  The norm is applied to the
    bf16-ROUNDED sum, as the separate rms_silu pass it replaces does; the raw output must equal scail_conv4c_e3's bit for bit."""
    cfg = [c for c in conv4.RESNORM if c.name == name][0]
    assert cfg.kt == 3
    Ti, H, W, Cin, N = shape
    x, w, b, r = _case(Ti, H, W, Cin, N, True, seed=8)
    gam = (1 + 0.1 * np.random.default_rng(5).standard_normal(N)).astype(np.float32)
    out, _ = R.run(cfg, x, w, b, r, cus=8, gamma=gam)
    raw_ref = R.reference(x, w, b, r)
    e3 = [c for c in conv4.CONT if c.name == "scail_conv4c_e3"][0]
    y3, _ = R.run(e3, x, w, b, r, cus=8)
    nrm_ref = R.reference_norm_silu(y3, gam)                     # of what the e3 kernel stores (bf16), like rms_silu_kernel reading it back
    if cfg.epi == 5:
        raw, nrm = out
        assert np.array_equal(raw, y3)
        assert np.abs(raw - raw_ref).max() <= 2.0 ** -6 * max(1.0, np.abs(raw_ref).max())
    else:
        nrm = out
    assert not np.isnan(nrm).any()
    err = np.abs(nrm - nrm_ref)
    assert err.max() <= 2.0 ** -7 * max(1.0, np.abs(nrm_ref).max()) and err.mean() <= 1e-3, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("ups,shape,cus", [(1, (2, 8, 8, 64, 96), 256), (1, (3, 9, 12, 96, 96), 8), (0, (3, 18, 20, 96, 96), 256)])
def test_conv4u_emulated_dual_output_next_norm(ups, shape, cus):
    """scail_conv4u_e7 (Cfg.epi 7, kt = 1): Resample's 1 x 3 x 3 convolution behind the nearest 2x upsample (wan_vae.py:76-85) with the RMS_norm + SiLU
    of the ResidualBlock that reads it next: the raw output (that block's shortcut operand) bit-identical to scail_conv4u_e0's, the normalised
    copy = the separate pass on the bf16-rounded output."""
    cfg = [c for c in conv4.RESNORM if c.name == "scail_conv4u_e7"][0]
    T, Hi, Wi, Cin, N = shape
    rng = np.random.default_rng(12)
    x = rng.standard_normal((T, Hi, Wi, Cin)).astype(np.float32)
    w = (rng.standard_normal((N, Cin, 1, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    gam = (1 + 0.1 * rng.standard_normal(N)).astype(np.float32)
    (raw, nrm), _ = R.run_k1(cfg, x, w, b, ups, cus=cus, gamma=gam)
    y0, _ = R.run_k1(conv4.UPSAMPLE[0], x, w, b, ups, cus=cus)
    assert np.array_equal(raw, y0) and not np.isnan(nrm).any()
    ref = R.reference_norm_silu(y0, gam)
    err = np.abs(nrm - ref)
    assert err.max() <= 2.0 ** -7 * max(1.0, np.abs(ref).max()) and err.mean() <= 1e-3, (float(err.max()), float(err.mean()))
