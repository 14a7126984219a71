"""Run the generated conv4 kernels (scail_amd/asmgen/conv4.py) in the CPU emulator on small convolutions.  TEST INFRASTRUCTURE."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from scail_amd.asmgen import conv4, sched  # noqa: E402
from tools import asm_emu as E  # noqa: E402
from tools.attn4_emu_run import from_bf16_bits, to_bf16_bits  # noqa: E402


def pack_weight(w, kpad=None):
    """(N, Cin, kt, 3, 3) -> (N, Kpad) with k = ((dt 3 + dh) 3 + dw) Cin + c   (scail_amd.ops.prep_conv_weight)."""
    N, Cin = w.shape[:2]
    k = w.shape[2] * 9 * Cin
    kpad = kpad or (k + 63) // 64 * 64
    out = np.zeros((N, kpad), dtype=np.float32)
    out[:, :k] = w.transpose(0, 2, 3, 4, 1).reshape(N, k)
    return out


def run(cfg: conv4.Cfg, x, w, bias=None, resid=None, pt=2, To=None, ot_mul=1, ot_off=0, y_frames=None, ldc=None, lazy=True, wgs=None, cus=256, gamma=None):
    """x (Ti, H, W, Cin), w (N, Cin, 3, 3, 3), bias (N) | None, resid (frames, H, W, N) | None; fp32 in, bf16 operands."""
    Ti, H, W, Cin = x.shape
    N = w.shape[0]
    To = To if To is not None else Ti + pt - 2
    y_frames = y_frames or (To * ot_mul + ot_off)
    ldc = ldc or N
    wp = pack_weight(w)
    mem = E.Memory(size=1 << 27)
    px = mem.alloc("x", to_bf16_bits(x))
    pw = mem.alloc("w", to_bf16_bits(wp))
    pb = mem.alloc("bias", bias.astype(np.float32)) if bias is not None else 0
    py = mem.alloc("y", np.full((y_frames, H, W, ldc), 0x7FC0, dtype=np.uint16))
    pr = mem.alloc("resid", to_bf16_bits(resid)) if resid is not None else 0
    pg = py2 = 0
    if gamma is not None and cfg.epi == 4:                     # epi 4: the `resid` argument carries gamma (fp32 [N])
        pr = mem.alloc("gamma", gamma.astype(np.float32))
    elif gamma is not None:                                    # epi 5 / 6: gamma has its own kernel argument; epi 5 writes a second output
        pg = mem.alloc("gamma", gamma.astype(np.float32))
        if cfg.epi == 5:
            py2 = mem.alloc("y2", np.full((y_frames, H, W, ldc), 0x7FC0, dtype=np.uint16))
    prog = conv4.Gen(cfg).program()
    args = conv4.pack_args(px, pw, pb, py, pr, Ti, To, H, W, Cin, N, wp.shape[1], pt, ot_mul, ot_off, ldc, resid.shape[-1] if resid is not None else 0, cus,
                           gamma=pg, y2=py2)
    stats = None
    for wg in (range(conv4.grid_blocks(To, H, W, N, cus)) if wgs is None else wgs):
        emu = E.Emu(prog, mem, n_waves=4, lds_bytes=conv4.LDS_BYTES, lazy=lazy)
        emu.launch(args, block_id=(wg, 0, 0))
        stats = emu.waves[0].stats
    y = from_bf16_bits(mem.read_back("y")).reshape(y_frames, H, W, ldc)
    if py2:
        return (y, from_bf16_bits(mem.read_back("y2")).reshape(y_frames, H, W, ldc)), stats
    return y, stats


def reference(x, w, bias=None, resid=None, pt=2, To=None):
    rt = lambda a: from_bf16_bits(to_bf16_bits(a)).astype(np.float64)
    Ti, H, W, Cin = x.shape
    N = w.shape[0]
    To = To if To is not None else Ti + pt - 2
    xp = np.zeros((Ti + pt + 2, H + 2, W + 2, Cin))
    xp[pt:pt + Ti, 1:-1, 1:-1] = rt(x)
    wr = rt(w)
    y = np.zeros((To, H, W, N))
    for dt in range(3):
        for dh in range(3):
            for dw in range(3):
                y += np.einsum("thwc,nc->thwn", xp[dt:dt + To, dh:dh + H, dw:dw + W], wr[:, :, dt, dh, dw])
    if bias is not None:
        y = y + bias.astype(np.float64)
    if resid is not None:
        y = y + rt(resid)
    return y


def reference_norm_silu(y, gamma):
    """rms_silu_kernel on the bf16-rounded convolution output: x * sqrt(C) / max(||x||, 1e-12) * gamma, SiLU (reference wan_vae.py:39-54 + SiLU)."""
    v = from_bf16_bits(to_bf16_bits(y.astype(np.float32))).astype(np.float64)
    nrm = np.maximum(np.sqrt((v * v).sum(-1, keepdims=True)), 1e-12)
    t = v * (np.sqrt(v.shape[-1]) / nrm) * gamma.astype(np.float64)
    return t / (1.0 + np.exp(-t))


def check_static(cfg):
    g = conv4.Gen(cfg)
    body = g.slice_body()
    return (sched.check_hazards(body + body) + sched.check_hazards(g.tile_start() + body) + sched.check_hazards(g.entry() + g.tile_setup() + g.epilogue() + g.tile_start())
            + sched.check_hazards(body + g.tile_end() + g.tile_setup()))


def run_k1(cfg: conv4.Cfg, x, w, bias=None, ups=0, ldc=None, lazy=True, wgs=None, cus=256, gamma=None):
    """the kt = 1 kernels: x (T, Hi, Wi, Cin), w (N, Cin, 1, 3, 3); output (T, Hi << ups, Wi << ups, N)."""
    T, Hi, Wi, Cin = x.shape
    N = w.shape[0]
    H, W = Hi << ups, Wi << ups
    ldc = ldc or N
    wp = pack_weight(w)
    mem = E.Memory(size=1 << 27)
    px = mem.alloc("x", to_bf16_bits(x))
    pw = mem.alloc("w", to_bf16_bits(wp))
    pb = mem.alloc("bias", bias.astype(np.float32)) if bias is not None else 0
    py = mem.alloc("y", np.full((T, H, W, ldc), 0x7FC0, dtype=np.uint16))
    pg = py2 = 0
    if gamma is not None:                                      # epi 7: raw + normalised copy
        pg = mem.alloc("gamma", gamma.astype(np.float32))
        py2 = mem.alloc("y2", np.full((T, H, W, ldc), 0x7FC0, dtype=np.uint16))
    prog = conv4.Gen(cfg).program()
    args = conv4.pack_args(px, pw, pb, py, 0, T, T, H, W, Cin, N, wp.shape[1], ups, 1, 0, ldc, 0, cus, gamma=pg, y2=py2)
    stats = None
    for wg in (range(conv4.grid_blocks(T, H, W, N, cus)) if wgs is None else wgs):
        emu = E.Emu(prog, mem, n_waves=4, lds_bytes=conv4.LDS_BYTES, lazy=lazy)
        emu.launch(args, block_id=(wg, 0, 0))
        stats = emu.waves[0].stats
    y = from_bf16_bits(mem.read_back("y")).reshape(T, H, W, ldc)
    if py2:
        return (y, from_bf16_bits(mem.read_back("y2")).reshape(T, H, W, ldc)), stats
    return y, stats


def reference_k1(x, w, bias=None, ups=0):
    """per frame: nearest 2x upsample (ups) then the 3x3 'same' convolution (reference Resample, wan_vae.py:76-85)."""
    rt = lambda a: from_bf16_bits(to_bf16_bits(a)).astype(np.float64)
    xr = rt(x)
    if ups:
        xr = xr.repeat(2, axis=1).repeat(2, axis=2)
    T, H, W, Cin = xr.shape
    N = w.shape[0]
    xp = np.zeros((T, H + 2, W + 2, Cin))
    xp[:, 1:-1, 1:-1] = xr
    wr = rt(w)
    y = np.zeros((T, H, W, N))
    for dh in range(3):
        for dw in range(3):
            y += np.einsum("thwc,nc->thwn", xp[:, dh:dh + H, dw:dw + W], wr[:, :, 0, dh, dw])
    if bias is not None:
        y = y + bias.astype(np.float64)
    return y


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for cfg in conv4.DEFAULTS:
        Ti, H, W, Cin, N = 3, 20, 18, 64, 96
        x = rng.standard_normal((Ti, H, W, Cin)).astype(np.float32)
        w = (rng.standard_normal((N, Cin, 3, 3, 3)) / np.sqrt(27 * Cin)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        resid = rng.standard_normal((Ti, H, W, N)).astype(np.float32) if cfg.epi == 3 else None
        print(cfg.name, "static", check_static(cfg)[:3])
        y, st = run(cfg, x, w, bias, resid)
        ref = reference(x, w, bias, resid)
        print("   max abs err", np.abs(y - ref).max(), "ref absmax", np.abs(ref).max(), st)
    for cfg in conv4.UPSAMPLE:
        for ups, (T, Hi, Wi, Cin, N) in ((1, (2, 8, 8, 64, 96)), (0, (3, 18, 20, 96, 96))):
            x = rng.standard_normal((T, Hi, Wi, Cin)).astype(np.float32)
            w = (rng.standard_normal((N, Cin, 1, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
            bias = rng.standard_normal(N).astype(np.float32)
            print(cfg.name, "ups", ups, "static", check_static(cfg)[:3])
            y, st = run_k1(cfg, x, w, bias, ups)
            ref = reference_k1(x, w, bias, ups)
            print("   max abs err", np.abs(y - ref).max(), "ref absmax", np.abs(ref).max(), "nan", int(np.isnan(y).sum()), st)
