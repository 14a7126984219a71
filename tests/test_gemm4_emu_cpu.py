"""CPU emulation (tools/asm_emu.py) of the generated GEMM kernels (scail_amd/asmgen/gemm4.py): the four shipped kernels
(scail_gemm4_e0/1/3/4, csrc/gemm4.s) and variants of the measurement build: the generator produces hazard-free code whose
results equal the fp64 product of the bf16-rounded operands, under the emulator's lazy (latest-allowed) completion of LDS / memory
operations -- the mode that exposes a missing or too-weak s_waitcnt."""
import math

import numpy as np
import pytest

from scail_amd.asmgen import gemm4
from tools import gemm4_emu_run as R


def _case(cfg, M, N, K, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    kw = dict(bias=rng.standard_normal(N).astype(np.float32))
    if cfg.epi in (3, 4):
        kw["resid"] = rng.standard_normal((M, N)).astype(np.float32)
    if cfg.epi == 3:
        kw.update(gate=rng.standard_normal((2, N)).astype(np.float32), rows_per_batch=(M + 1) // 2)
    return x, w, kw


def _variant(name):
    return [c for c in gemm4.DEFAULTS + gemm4.variant_cfgs() if c.name == name][0]


@pytest.mark.parametrize("name,shape", [("scail_gemm4_e0", (300, 256, 192)),              # SHIPPED: LDS-DMA two tiles deep, 16x16x32 MFMAs; ragged last m-tile
                                        ("scail_gemm4_e1", (256, 256, 128)),              # + GELU-tanh
                                        ("scail_gemm4_e3", (264, 256, 320)),              # gate * (acc + bias) + residual, odd tile count
                                        ("scail_gemm4_e4", (520, 512, 128)),              # residual, two m-tiles and n-tiles + ragged third
                                        ("scail_gemm4p_e0", (300, 512, 192)),             # measurement build: tile-major packed W experiment, 2 n-tiles
                                        ("scail_gemm4p_e3", (264, 256, 448)),             # ... 7 k-tiles
                                        ("scail_gemm4_e0_reg", (300, 256, 192)),          # measurement build: register staging, 32x32x16
                                        ("scail_gemm4_e0_dma2", (300, 256, 320)),         # ... LDS-DMA two tiles deep, 32x32x16
                                        ("scail_gemm4_e0_spread", (256, 256, 256)),
                                        ("scail_gemm4_e3_pst", (1300, 512, 192)),         # measurement build: persistent workgroups, 12 tiles on 8 workgroups
                                        ("scail_gemm4_e0_pst", (520, 256, 128))])         # ... 3 tiles, idle workgroups
def test_gemm4_emulated(name, shape):
    cfg = _variant(name)
    assert R.check_static(cfg) == [], "hazard distances of the generated loop / prologue"
    x, w, kw = _case(cfg, *shape)
    y, _ = R.run(cfg, x, w, lazy=True, **kw)
    ref = R.reference(cfg, x, w, **kw)
    err = np.abs(y - ref)
    assert err.max() <= 2.0 ** -7 * max(1.0, np.abs(ref).max()), float(err.max())     # one bf16 rounding of the output
