"""Run the generated gemm4 kernels (scail_amd/asmgen/gemm4.py) in the CPU emulator on small GEMMs.  TEST INFRASTRUCTURE."""
from __future__ import annotations

import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from scail_amd.asmgen import gemm4, sched  # noqa: E402
from tools import asm_emu as E  # noqa: E402
from tools.attn4_emu_run import from_bf16_bits, to_bf16_bits  # noqa: E402


def gelu_tanh(x):
    return 0.5 * x * (1.0 + np.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))


def run(cfg: gemm4.Cfg, x, w, bias=None, resid=None, gate=None, rows_per_batch=0, lda=None, lazy=True, grid=8):
    """x (M, K), w (N, K), bias (N) | None, resid (M, N) | None, gate (nb, N) | None  (fp32 in; bf16 operands)."""
    M, K = x.shape
    N = w.shape[0]
    lda = lda or K
    mem = E.Memory(size=1 << 26)
    xs = np.zeros((M, lda), dtype=np.uint16)
    xs[:, :K] = to_bf16_bits(x)
    px = mem.alloc("x", xs)
    wb = to_bf16_bits(w)
    pw = mem.alloc("w", gemm4.pack_w(wb, N, K) if getattr(cfg, "wpacked", False) else wb)
    pb = mem.alloc("bias", bias.astype(np.float32)) if bias is not None else 0
    py = mem.alloc("y", np.zeros((M, N), dtype=np.uint16))
    pr = mem.alloc("resid", to_bf16_bits(resid)) if resid is not None else 0
    pg = mem.alloc("gate", gate.astype(np.float32)) if gate is not None else 0
    table = np.array(gemm4.tile_table(M, N), dtype=np.uint32)
    pt = mem.alloc("table", table)
    prog = gemm4.Gen(cfg).program()
    # persistent kernels: fewer workgroups than tiles (a multiple of 8, like the launcher's), each walks entries wg, wg + grid, ...
    grid = min(len(table), grid) if getattr(cfg, "persist", False) else len(table)
    args = gemm4.pack_args(px, pw, pb, py, pr, pg, pt, lda, N, N, N if gate is not None else 0, M, N, K, rows_per_batch, grid, len(table))
    stats = None
    for wg in range(grid):
        emu = E.Emu(prog, mem, n_waves=4, lds_bytes=163840, lazy=lazy)
        emu.launch(args, block_id=(wg, 0, 0))
        stats = emu.waves[0].stats
    return from_bf16_bits(mem.read_back("y")), stats


def reference(cfg, x, w, bias=None, resid=None, gate=None, rows_per_batch=0):
    rt = lambda a: from_bf16_bits(to_bf16_bits(a)).astype(np.float64)
    y = rt(x) @ rt(w).T
    if bias is not None:
        y = y + bias.astype(np.float64)
    if cfg.epi == 1:
        y = gelu_tanh(y)
    if cfg.epi in (3, 4):
        if cfg.epi == 3:
            b = (np.arange(x.shape[0]) // rows_per_batch) if rows_per_batch else np.zeros(x.shape[0], dtype=int)
            y = y * gate.astype(np.float64)[b]
        y = rt(resid) + y
    return y


def check_static(cfg):
    g = gemm4.Gen(cfg)
    if getattr(cfg, "persist", False):
        return sched.check_hazards(g.loop() + g.loop()) + sched.check_hazards([i for i in g.program()])
    lp = g.loop()
    return sched.check_hazards(lp + lp) + sched.check_hazards(g.prologue() + lp)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for cfg in gemm4.DEFAULTS:
        M, N, K = 400, 512, 192
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        resid = rng.standard_normal((M, N)).astype(np.float32)
        gate = rng.standard_normal((2, N)).astype(np.float32)
        kw = dict(bias=bias)
        if cfg.epi in (3, 4):
            kw["resid"] = resid
        if cfg.epi == 3:
            kw.update(gate=gate, rows_per_batch=208)
        print(cfg.name, "static", check_static(cfg)[:3])
        y, st = run(cfg, x, w, **kw)
        ref = reference(cfg, x, w, **kw)
        print("   max abs err", np.abs(y - ref).max(), "ref absmax", np.abs(ref).max(), st)
