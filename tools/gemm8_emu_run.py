"""Run the generated gemm8 kernels (scail_amd/asmgen/gemm8.py) in the CPU emulator.  TEST INFRASTRUCTURE."""
from __future__ import annotations

import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from scail_amd.asmgen import gemm8, sched  # noqa: E402
from tools import asm_emu as E  # noqa: E402
from tools.attn4_emu_run import from_bf16_bits, to_bf16_bits  # noqa: E402
from tools.gemm4_emu_run import reference  # noqa: E402,F401


def run(cfg, x, w, bias=None, resid=None, gate=None, rows_per_batch=0, lda=None, lazy=True):
    M, K = x.shape
    N = w.shape[0]
    lda = lda or K
    mem = E.Memory(size=1 << 26)
    xs = np.zeros((M, lda), dtype=np.uint16)
    xs[:, :K] = to_bf16_bits(x)
    px, pw = mem.alloc("x", xs), mem.alloc("w", to_bf16_bits(w))
    pb = mem.alloc("bias", bias.astype(np.float32)) if bias is not None else 0
    py = mem.alloc("y", np.zeros((M, N), dtype=np.uint16))
    pr = mem.alloc("resid", to_bf16_bits(resid)) if resid is not None else 0
    pg = mem.alloc("gate", gate.astype(np.float32)) if gate is not None else 0
    table = np.array(gemm8.tile_table(M, N), dtype=np.uint32)
    pt = mem.alloc("table", table)
    prog = gemm8.Gen(cfg).program()
    args = gemm8.pack_args(px, pw, pb, py, pr, pg, pt, lda, N, N, N if gate is not None else 0, M, N, K, rows_per_batch)
    for wg in range(len(table)):
        emu = E.Emu(prog, mem, n_waves=8, lds_bytes=131072, lazy=lazy)
        emu.launch(args, block_id=(wg, 0, 0))
    return from_bf16_bits(mem.read_back("y"))


def check_static(cfg):
    g = gemm8.Gen(cfg)
    lp = g.loop()
    return sched.check_hazards(lp + lp) + sched.check_hazards(g.prologue() + lp)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for cfg in gemm8.DEFAULTS:
        for (M, N, K, lazy) in ((400, 512, 192, True), (256, 256, 256, False)):
            x = rng.standard_normal((M, K)).astype(np.float32)
            w = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
            kw = dict(bias=rng.standard_normal(N).astype(np.float32))
            if cfg.epi in (3, 4):
                kw["resid"] = rng.standard_normal((M, N)).astype(np.float32)
            if cfg.epi == 3:
                kw.update(gate=rng.standard_normal((2, N)).astype(np.float32), rows_per_batch=208)
            y = run(cfg, x, w, lazy=lazy, **kw)
            ref = reference(cfg, x, w, **kw)
            print(cfg.name, M, N, K, "lazy" if lazy else "eager", "max abs err %.4f" % np.abs(y - ref).max(), "ref absmax %.2f" % np.abs(ref).max(),
                  "static", len(check_static(cfg)), flush=True)
