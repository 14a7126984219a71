"""Run the generated attn4 kernel (scail_amd/asmgen/attn4.py) in the CPU emulator (tools/asm_emu.py) on a small attention
problem.  TEST INFRASTRUCTURE (used by tests/test_attn4_emu_cpu.py and for debugging the generator)."""
from __future__ import annotations

import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from scail_amd.asmgen import attn4, sched  # noqa: E402
from tools import asm_emu as E  # noqa: E402


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    return E.bf16_round(x.astype(np.float32)).astype(np.uint16)


def from_bf16_bits(h: np.ndarray) -> np.ndarray:
    return E.bf16_to_f32(h.astype(np.uint32))


def transpose_v(v_bits: np.ndarray, heads: int) -> np.ndarray:
    """scail_transpose_v: (B, Lk, H*128) -> (B, H, 128, ceil64(Lk)), keys permuted inside each 16-group (bits 2 <-> 3), zero pad."""
    B, Lk, D = v_bits.shape
    Lkp = (Lk + 63) // 64 * 64
    out = np.zeros((B, heads, 128, Lkp), dtype=np.uint16)
    p = np.arange(Lkp)
    key = (p & ~12) | (((p >> 3) & 1) << 2) | (((p >> 2) & 1) << 3)
    valid = key < Lk
    vv = v_bits.reshape(B, Lk, heads, 128)
    out[:, :, :, p[valid]] = np.transpose(vv[:, key[valid]], (0, 2, 3, 1))
    return out


def reference(q, k, v, heads):
    """fp64 softmax(q k^T / sqrt(128)) v on the bf16-rounded operands: q (B, Lq, H*128), k / v (B, Lk, H*128)."""
    B, Lq, D = q.shape
    out = np.zeros((B, Lq, D))
    for b in range(B):
        for h in range(heads):
            sl = slice(h * 128, (h + 1) * 128)
            s = q[b, :, sl].astype(np.float64) @ k[b, :, sl].astype(np.float64).T / math.sqrt(128.0)
            s -= s.max(axis=1, keepdims=True)
            p = np.exp(s)
            out[b, :, sl] = (p / p.sum(axis=1, keepdims=True)) @ v[b, :, sl].astype(np.float64)
    return out


def run(cfg: attn4.Cfg, q: np.ndarray, ksegs, vsegs, heads: int, lazy: bool = True, thr_log2: float = 8.0, program=None, mode=None,
        raw_scale: bool = False, launches=None, count_restarts: bool = True):
    """q (B, Lq, H*128) fp32; ksegs / vsegs: lists (one per segment) of (B, Lk, H*128) fp32.  Returns O (B, Lq, H*128) fp32
    and the emulator statistics of the last workgroup.  raw_scale (qscale kernels): q goes in UNSCALED with sl2 = scale * log2(e)
    as the kernel argument (the prologue multiplies the fragments), instead of pre-multiplied with sl2 = 0."""
    B, Lq, D = q.shape
    n_seg = len(ksegs)
    Lk = ksegs[0].shape[1]
    assert Lk % 64 == 0 or getattr(cfg, "ragged", False)
    Lkp = (Lk + 63) // 64 * 64
    mem = E.Memory(size=1 << 26)
    sl2 = (1.0 / math.sqrt(128.0)) * 1.4426950408889634
    fold = getattr(cfg, "fold", False)
    # fold: the caller hands over q already multiplied by scale * log2(e) (one rounding to bf16, as scail_rmsnorm_rope_scaled does)
    qb = to_bf16_bits(q * np.float32(sl2)) if (fold and not raw_scale) else to_bf16_bits(q)
    kb = np.stack([to_bf16_bits(x) for x in ksegs])                       # (S, B, Lk, D)
    vt = np.stack([transpose_v(to_bf16_bits(x), heads) for x in vsegs])    # (S, B, H, 128, Lkp)
    pq = mem.alloc("q", qb)
    pk = mem.alloc("k", kb)
    pvt = mem.alloc("vt", vt)
    po = mem.alloc("o", np.zeros((B, Lq, D), dtype=np.uint16))
    pctr = mem.alloc("restarts", np.zeros(4, dtype=np.uint32))          # the optional restart counter of the kernel arguments
    thr = thr_log2 if fold else thr_log2 / sl2           # fold kernels see scores in log2 units
    if fold and not raw_scale:
        sl2 = 0.0 if getattr(cfg, "qscale", False) else 1.0       # qscale kernels: 0 = q is in log2 units already
    # one launch over all items, or the launches of a split attention: [(cfg, item0, n_items, mode), ...] (scail_flash_attn_bf16's mixed
    # launch: whole rounds of 256-row tiles, then 192-row tiles for the remaining rows)
    plan = launches if launches is not None else [(cfg, 0, None, mode)]
    stats = None
    for lcfg, item0, n_items, lmode in plan:
        prog = program if (program is not None and lcfg is cfg) else attn4.Gen(lcfg).program()
        m = attn4.xcd_mode(B, heads) if lmode is None else lmode
        total = ((Lq + lcfg.rows - 1) // lcfg.rows) * heads * B
        n = total - item0 if n_items is None else n_items
        args = attn4.pack_args(pq, pk, pvt, po, Lq * D, D, B * Lk * D, Lk * D, D, B * heads * 128 * Lkp, heads * 128 * Lkp, Lq * D, D,
                               heads, Lq, Lk, Lkp, n_seg, sl2, thr, n_batch=B, mode=m, rows=lcfg.rows, item0=item0, n_items=n,
                               restarts=pctr if count_restarts else 0)
        for wid in range(attn4.grid_for(n, m)):
            emu = E.Emu(prog, mem, n_waves=4, lds_bytes=lcfg.lds_bytes, lazy=lazy)
            emu.launch(args, block_id=(wid, 0, 0))
            if stats is None or emu.waves[0].stats.get("mfma", 0) > 0:       # xcd_mode 2 pads the grid with workgroups that exit at once
                stats = emu.waves[0].stats
    stats = dict(stats or {})
    stats["restarts"] = int(mem.read_back("restarts")[0])              # workgroups that ran again (0 without count_restarts)
    return from_bf16_bits(mem.read_back("o")), stats


def run_x2(cfg: attn4.Cfg, q: np.ndarray, k1, v1, k2, v2, heads: int, n_wgs: int = 2, lazy: bool = True, thr_log2: float = 8.0, raw_scale: bool = False,
           program=None):
    """The two-key-set cross attention kernel (Cfg.x2): q (B, Lq, H*128) fp32; k1 / v1 (B, Lk1, H*128), k2 / v2 (B2, Lk2, H*128) with B2 in
    {1, B} (a shared CLIP set has batch stride 0).  ``n_wgs`` persistent workgroups walk over the items.  Returns O and the statistics."""
    B, Lq, D = q.shape
    Lk1, Lk2 = k1.shape[1], k2.shape[1]
    Lkp1, Lkp2 = (Lk1 + 63) // 64 * 64, (Lk2 + 63) // 64 * 64
    mem = E.Memory(size=1 << 26)
    sl2 = (1.0 / math.sqrt(128.0)) * 1.4426950408889634
    qb = to_bf16_bits(q) if raw_scale else to_bf16_bits(q * np.float32(sl2))
    pq = mem.alloc("q", qb)
    pk1 = mem.alloc("k1", to_bf16_bits(k1))
    pv1 = mem.alloc("vt1", transpose_v(to_bf16_bits(v1), heads))
    pk2 = mem.alloc("k2", to_bf16_bits(k2))
    pv2 = mem.alloc("vt2", transpose_v(to_bf16_bits(v2), heads))
    po = mem.alloc("o", np.zeros((B, Lq, D), dtype=np.uint16))
    prog = program if program is not None else attn4.Gen(cfg).program()
    bs = lambda x, n: 0 if x.shape[0] == 1 and B > 1 else n
    args = attn4.pack_args_x2(pq, pk1, pv1, pk2, pv2, po, Lq * D, D, bs(k1, Lk1 * D), D, bs(k1, heads * 128 * Lkp1), bs(k2, Lk2 * D),
                              bs(k2, heads * 128 * Lkp2), Lq * D, D, heads, Lq, Lk1, Lkp1, Lk2, Lkp2, sl2 if raw_scale else 0.0, thr_log2,
                              n_batch=B, n_wgs=n_wgs)
    stats = []
    for wid in range(n_wgs):
        emu = E.Emu(prog, mem, n_waves=4, lds_bytes=cfg.lds_bytes, lazy=lazy)
        emu.launch(args, block_id=(wid, 0, 0))
        stats.append(emu.waves[0].stats)
    return from_bf16_bits(mem.read_back("o")), stats


def reference_x2(q, k1, v1, k2, v2, heads):
    """bf16(bf16(softmax(q k1^T / sqrt d) v1) + softmax(q k2^T / sqrt d) v2) in fp64 on the given (already rounded) operands."""
    rt = lambda x: from_bf16_bits(to_bf16_bits(x.astype(np.float32))).astype(np.float64)
    B = q.shape[0]
    bc = lambda x: np.broadcast_to(x, (B,) + x.shape[1:])
    o1 = reference(q, bc(k1), bc(v1), heads)
    o2 = reference(q, bc(k2), bc(v2), heads)
    return rt(o1) + o2


def check_static(cfg: attn4.Cfg):
    """hazard re-check of every scheduled block (loop bodies circularly)."""
    g = attn4.Gen(cfg)
    errs = []
    for p in range(cfg.unroll):
        body = g.iter_block(p, tail=False) + g.iter_end(p, "hot")
        nxt = g.iter_block((p + 1) % cfg.unroll, tail=False)
        errs += sched.check_hazards(body + nxt)
        errs += sched.check_hazards(body + g.iter_block((p + 1) % cfg.unroll, tail=True))
        if getattr(cfg, "opt", False):      # the optimistic hot loop: into itself and into the (max-tracking) remainder chain
            fast = g.iter_block(p, tail=False, nomax=True) + g.iter_end(p, "hotf", nomax=True)
            errs += sched.check_hazards(fast + g.iter_block((p + 1) % cfg.unroll, tail=False, nomax=True))
            errs += sched.check_hazards(fast + g.iter_block((p + 1) % cfg.unroll, tail=False, careful=True))
    errs += sched.check_hazards(g.prologue() + g.segment_start() + g.iter_block(0, tail=False))
    if hasattr(g, "segment_end_and_epilogue") and cfg.mi == 16:      # the restart decision lives here (round 6: lane read of a fresh VALU result)
        errs += sched.check_hazards(g.segment_end_and_epilogue())
    return errs


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    cfg = attn4.Cfg(rd=int(os.environ.get("RD", "4")), cap=int(os.environ.get("CAP", "5")))
    B, H, Lq, Lk = 1, 1, 256, int(os.environ.get("LK", "512"))
    q = rng.standard_normal((B, Lq, H * 128)).astype(np.float32)
    k = rng.standard_normal((B, Lk, H * 128)).astype(np.float32)
    v = rng.standard_normal((B, Lk, H * 128)).astype(np.float32)
    print("static hazards:", check_static(cfg)[:5])
    o, st = run(cfg, q, [k], [v], H, lazy=os.environ.get("LAZY", "1") == "1")
    ref = reference(from_bf16_bits(to_bf16_bits(q)), from_bf16_bits(to_bf16_bits(k)), from_bf16_bits(to_bf16_bits(v)), H)
    print("max abs err", np.abs(o - ref).max(), "stats", st)
