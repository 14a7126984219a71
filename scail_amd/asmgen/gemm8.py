"""gemm8 -- hand-scheduled 8-wave bf16 GEMM  y = epi(x W^T + b)  for gfx950 (measurement build only: scail_amd/build.py writes its assembly to build_abl/gemm8.s).

Same problem, argument block, tile-order table and epilogues as gemm4.py (see there for the reference call sites); different
occupancy: 8 waves = TWO per SIMD, each with 128 arch VGPRs + 128 AGPRs.  Measured on MI355X: with ONE wave per SIMD (gemm4) every
global load, LDS write or fragment read that finds its queue busy stalls the wave's MFMA stream (MFMA-only ceiling 2 078 TFLOP/s,
1 170-1 230 with the data movement); with two waves per SIMD the partner keeps the matrix pipe busy meanwhile.

  * tile 256 x 256 x 64, waves 2 (m) x 4 (n), wave tile 128 x 64 = 4 x 2 blocks of 32 x 32: 128 accumulators in a[0:127]
  * staging through registers (ONE set of 8 x 16-byte pieces per lane: 4 x rows + 4 W rows of 128 B), written to the LDS slot
    after the tile's barrier and re-loaded at once (guide T14: "write after the barrier, re-issue the same registers")
  * ONE s_barrier per k-tile, before the last k-step (every wave has read its last fragments of the slot by then); the first
    fragments of the next tile are read while the last k-step computes
  * fragments per 16-wide k-step: 2 W + 4 x `ds_read_b128` for 8 MFMAs, double-buffered; LDS image XOR-swizzled on the write address
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

from . import isa, sched
from .gemm4 import KERNARG_SIZE, pack_args, tile_table  # noqa: F401  (same argument block and tile-order table)
from .gemm4 import (S_KARG, S_WG, S_X, S_W, S_BIAS, S_Y, S_RES, S_GATE, S_TAB, S_LDA, S_LDC, S_LDR, S_GS, S_M, S_N, S_K, S_RPB,
                    S_XRSRC, S_WRSRC, S_KOFF, S_KMAX, S_T, S_KT, S_WAVE, S_WM, S_WN, S_M0T, S_N0T, ST, S_SAVE)
from .isa import A, S, V, I32, F32, Neg, VCC, EXEC, Instr


@dataclass
class Cfg:
    epi: int = 0
    cap: int = 3
    name: str = "scail_gemm8_e0"
    abl: str = ""
    wr_from: float = 24.0
    wr_step: float = 0.5
    ld_from: float = 26.0
    ld_step: float = 0.75


def ACC(nb, mb): return A((nb * 4 + mb) * 16, 16)
def FW(buf, nb): return V(buf * 24 + nb * 4, 4)
def FX(buf, mb): return V(buf * 24 + 8 + mb * 4, 4)
def STG(op, i): return V(64 + op * 16 + i * 4, 4)


XADDR = [[V(48 + s * 4 + ks) for ks in range(4)] for s in range(2)]
WADDR = [[V(56 + s * 4 + ks) for ks in range(4)] for s in range(2)]
XSRC = [V(96 + i) for i in range(4)]
WSRC = [V(100 + i) for i in range(4)]
WRADDR = [[V(104 + s * 2 + par) for par in range(2)] for s in range(2)]
LANE = V(108)
T_ = [V(109 + i) for i in range(19)]          # T_[1], T_[2] = lane geometry (kept through the loop)


class Gen:
    def __init__(self, cfg: Cfg):
        self.cfg = cfg

    def mfmas(self, ks: int) -> List[Instr]:
        buf = ks & 1
        return [isa.mfma(ACC(nb, mb), FW(buf, nb), FX(buf, mb), ACC(nb, mb)) for nb in range(2) for mb in range(4)]

    def frag_reads(self, slot: int, ks: int, t0: float, step: float) -> List[Instr]:
        buf = ks & 1
        out = [isa.ds_read_b128(FW(buf, 0), WADDR[slot][ks], 0), isa.ds_read_b128(FX(buf, 0), XADDR[slot][ks], 0),
               isa.ds_read_b128(FX(buf, 1), XADDR[slot][ks], 4096), isa.ds_read_b128(FW(buf, 1), WADDR[slot][ks], 4096),
               isa.ds_read_b128(FX(buf, 2), XADDR[slot][ks], 8192), isa.ds_read_b128(FX(buf, 3), XADDR[slot][ks], 12288)]
        for k, i in enumerate(out):
            i.target_gap = t0 + step * k
        return out

    def load_tile(self, t0: float, step: float) -> List[Instr]:
        out = []
        k = 0
        for op, offs, rsrc in ((0, XSRC, S_XRSRC), (1, WSRC, S_WRSRC)):
            for i in range(4):
                out.append(isa.buffer_load(4, STG(op, i), offs[i], rsrc, S_KOFF, 0, target_gap=t0 + step * k))
                k += 1
        out.append(isa.sop("s_add_u32", S_KOFF, S_KOFF, I32(128), target_gap=t0 + step * k))
        out.append(isa.sop("s_min_u32", S_KOFF, S_KOFF, S_KMAX, target_gap=t0 + step * k + 0.1))
        return out

    def write_tile(self, slot: int, t0: float, step: float) -> List[Instr]:
        out = []
        k = 0
        for op in range(2):
            for i in range(4):
                out.append(isa.ds_write(16, WRADDR[slot][i & 1], STG(op, i), op * 32768 + 1024 * i, target_gap=t0 + step * k))
                k += 1
        return out

    def body(self, p: int) -> List[Instr]:
        c = self.cfg
        abl = c.abl.split(",")
        blk: List[Instr] = []
        reads_p: List[Instr] = []
        for ks in range(3):
            r = self.frag_reads(p, ks + 1, 8.0 * ks + 0.5, 1.0) if "lds" not in abl else []
            reads_p += r
            blk += r + self.mfmas(ks)
        w1, w2, bar = isa.waitcnt(lgkmcnt=0, target_gap=23.3), isa.waitcnt(vmcnt=0, target_gap=23.4), isa.barrier(target_gap=23.5)
        w1.after, bar.after = list(reads_p), list(reads_p) + [w1, w2]
        sync = [w1, w2] + ([bar] if "bar" not in abl else [])
        wr = self.write_tile(p, c.wr_from, c.wr_step) if "dma" not in abl else []
        ld = self.load_tile(c.ld_from, c.ld_step) if "dma" not in abl else []
        nxt = self.frag_reads(p ^ 1, 0, 28.0, 0.6) if "lds" not in abl else []
        for i in wr + ld:
            i.after = list(sync)
        for i in nxt:
            i.after = list(sync) + list(wr)          # the next tile's first reads are the last LDS operations of the body
        blk += sync + wr + ld + nxt + self.mfmas(3)
        return sched.schedule(blk, cap=c.cap, lookahead=1.0)

    def first_reads(self) -> List[Instr]:
        if "lds" in self.cfg.abl.split(","):
            return []
        tail: List[Instr] = []
        sched.insert_lgkm_waits(self.body(1), carry_in=[], carry_out=tail)
        order = [tuple(i.writes()) for i in tail[-6:]]
        reads = {tuple(i.writes()): i for i in self.frag_reads(0, 0, 0, 0)}
        assert sorted(order) == sorted(reads), "the last 6 LDS operations of a body must be the next tile's first fragment reads"
        return [reads[k] for k in order]

    def addr64_madd(self, ptr, a, b, shift):
        lo, hi = ST[0], ST[1]
        st = S(ST[2].idx, 2)
        return [isa.sop("s_mul_i32", lo, a, b), isa.sop("s_mul_hi_u32", hi, a, b), isa.sop("s_mov_b32", st.sub(0), lo),
                isa.sop("s_mov_b32", st.sub(1), hi), isa.sop("s_lshl_b64", st, st, I32(shift)),
                isa.sop("s_add_u32", ptr.sub(0), ptr.sub(0), st.sub(0)), isa.sop("s_addc_u32", ptr.sub(1), ptr.sub(1), st.sub(1))]

    def prologue(self) -> List[Instr]:
        c = self.cfg
        o: List[Instr] = [isa.label(c.name)]
        o += [isa.s_load(8, S(8, 8), S_KARG, 0), isa.s_load(4, S(16, 4), S_KARG, 32), isa.s_load(2, S_TAB, S_KARG, 48),
              isa.s_load(8, S(24, 8), S_KARG, 56), isa.s_load(4, S(32, 4), S_KARG, 88),
              isa.vop("v_and_b32", LANE, I32(63), V(0)), isa.vop("v_lshrrev_b32", T_[0], I32(6), V(0)),
              isa.waitcnt(lgkmcnt=0), isa.vop("v_readfirstlane_b32", S_WAVE, T_[0])]
        ent = ST[4]
        o += [isa.sop("s_lshl_b32", ST[5], S_WG, I32(2)), isa.sop("s_add_u32", S_TAB.sub(0), S_TAB.sub(0), ST[5]),
              isa.sop("s_addc_u32", S_TAB.sub(1), S_TAB.sub(1), I32(0)), isa.s_load(1, ent, S_TAB, 0), isa.waitcnt(lgkmcnt=0),
              isa.sop("s_cmp_eq_u32", None, ent, I32(0xFFFFFFFF)), isa.branch("s_cbranch_scc1", "L_exit"),
              isa.sop("s_and_b32", ST[5], ent, I32(0xFFFF)), isa.sop("s_lshr_b32", ST[6], ent, I32(16)),
              isa.sop("s_lshl_b32", S_M0T, ST[5], I32(8)), isa.sop("s_lshl_b32", S_N0T, ST[6], I32(8)),
              isa.sop("s_lshr_b32", S_WM, S_WAVE, I32(2)), isa.sop("s_and_b32", S_WN, S_WAVE, I32(3))]
        o += self.addr64_madd(S_X, S_M0T, S_LDA.sub(0), 1) + self.addr64_madd(S_W, S_N0T, S_K, 1)
        for rs, ptr in ((S_XRSRC, S_X), (S_WRSRC, S_W)):
            o += [isa.sop("s_mov_b32", rs.sub(0), ptr.sub(0)), isa.sop("s_and_b32", rs.sub(1), ptr.sub(1), I32(0xFFFF)),
                  isa.sop("s_mov_b32", rs.sub(2), I32(0xFFFFFFFF)), isa.sop("s_mov_b32", rs.sub(3), I32(0x00020000))]
        ldab, kb2 = ST[6], ST[7]
        o += [isa.sop("s_lshl_b32", ldab, S_LDA.sub(0), I32(1)), isa.sop("s_lshl_b32", kb2, S_K, I32(1)),
              isa.sop("s_lshr_b32", S_KT, S_K, I32(6)), isa.sop("s_sub_u32", ST[8], S_KT, I32(1)), isa.sop("s_lshl_b32", S_KMAX, ST[8], I32(7)),
              isa.sop("s_mov_b32", S_KOFF, I32(0)), isa.sop("s_mov_b32", S_T, I32(0))]
        ql, g, t = T_[1], T_[2], T_
        o += [isa.vop("v_and_b32", ql, I32(31), LANE), isa.vop("v_lshrrev_b32", g, I32(5), LANE)]
        # fragment read addresses: row r (128 B), chunk (2 ks + g) ^ ((r >> 1) & 7); x rows 128 wm + 32 mb + ql, W rows 64 wn + 32 nb + ql
        o += [isa.vop("v_lshrrev_b32", t[3], I32(1), ql), isa.vop("v_and_b32", t[3], I32(7), t[3]), isa.vop("v_lshlrev_b32", t[4], I32(7), ql),
              isa.vop("v_lshlrev_b32", t[5], I32(14), S_WM), isa.vop("v_add_u32", t[5], t[5], t[4]),
              isa.vop("v_lshlrev_b32", t[6], I32(13), S_WN), isa.vop("v_add_u32", t[6], t[6], t[4]), isa.vop("v_add_u32", t[6], I32(32768), t[6])]
        for ks in range(4):
            o += [isa.vop("v_or_b32", t[7], I32(2 * ks), g), isa.vop("v_xor_b32", t[7], t[7], t[3]),
                  isa.vop("v_lshl_add_u32", XADDR[0][ks], t[7], I32(4), t[5]), isa.vop("v_lshl_add_u32", WADDR[0][ks], t[7], I32(4), t[6]),
                  isa.vop("v_add_u32", XADDR[1][ks], I32(65536), XADDR[0][ks]), isa.vop("v_add_u32", WADDR[1][ks], I32(65536), WADDR[0][ks])]
        # staging: piece i of this wave = tile rows 32 w + 8 i + (lane >> 3), 16-byte chunk lane & 7 (coalesced 128-byte rows)
        mlast = ST[9]
        o += [isa.sop("s_sub_u32", mlast, S_M, S_M0T), isa.sop("s_sub_u32", mlast, mlast, I32(1)),
              isa.vop("v_lshrrev_b32", t[3], I32(3), LANE), isa.vop("v_and_b32", t[4], I32(7), LANE), isa.vop("v_lshlrev_b32", t[5], I32(5), S_WAVE)]
        for i in range(4):
            o += [isa.vop("v_add_u32", t[6], I32(8 * i), t[3]), isa.vop("v_add_u32", t[6], t[6], t[5]),
                  isa.vop("v_min_u32", t[8], t[6], mlast), isa.vop("v_mul_lo_u32", t[8], t[8], ldab), isa.vop("v_lshl_add_u32", XSRC[i], t[4], I32(4), t[8]),
                  isa.vop("v_mul_lo_u32", t[9], t[6], kb2), isa.vop("v_lshl_add_u32", WSRC[i], t[4], I32(4), t[9])]
        # LDS write address: 4096 w + (lane >> 3) * 128 + ((lane & 7) ^ (4 * parity + (lane >> 4))) * 16   (+ op * 32 KB + 1 KB * i as immediate)
        o += [isa.vop("v_lshrrev_b32", t[6], I32(4), LANE), isa.vop("v_lshlrev_b32", t[7], I32(7), t[3]),
              isa.vop("v_lshlrev_b32", t[8], I32(12), S_WAVE), isa.vop("v_add_u32", t[7], t[7], t[8])]
        for par in range(2):
            o += [isa.vop("v_add_u32", t[8], I32(4 * par), t[6]), isa.vop("v_xor_b32", t[8], t[4], t[8]),
                  isa.vop("v_lshl_add_u32", WRADDR[0][par], t[8], I32(4), t[7]), isa.vop("v_add_u32", WRADDR[1][par], I32(65536), WRADDR[0][par])]
        for i in range(128):
            o.append(isa.vop("v_accvgpr_write_b32", A(i), I32(0)))
        # pipeline fill: tiles 0, 1 in LDS, tile 2 in flight in the staging registers, first fragments of tile 0
        o += self.load_tile(0, 0) + [isa.waitcnt(vmcnt=0)] + self.write_tile(0, 0, 0)
        o += self.load_tile(0, 0) + [isa.waitcnt(vmcnt=0)] + self.write_tile(1, 0, 0)
        o += self.load_tile(0, 0) + [isa.waitcnt(lgkmcnt=0), isa.barrier()]
        o = sched.pad_hazards(sched.insert_lgkm_waits(o))
        return o + self.first_reads()

    def loop(self) -> List[Instr]:
        first = self.first_reads()
        sig = lambda q: [tuple(i.writes()) for i in q]
        c0: List[Instr] = []
        c1: List[Instr] = []
        b0 = sched.insert_lgkm_waits(self.body(0), carry_in=first, carry_out=c0)
        b1 = sched.insert_lgkm_waits(self.body(1), carry_in=first, carry_out=c1)
        if "lds" not in self.cfg.abl.split(","):
            assert sig(c0[-6:]) == sig(first) and sig(c1[-6:]) == sig(first)
        o: List[Instr] = [isa.label("L_loop")]
        o += b0 + [isa.sop("s_add_u32", S_T, S_T, I32(1)), isa.sop("s_cmp_lt_u32", None, S_T, S_KT), isa.branch("s_cbranch_scc0", "L_done")]
        o += b1 + [isa.sop("s_add_u32", S_T, S_T, I32(1)), isa.sop("s_cmp_lt_u32", None, S_T, S_KT), isa.branch("s_cbranch_scc1", "L_loop")]
        o += [isa.label("L_done"), isa.waitcnt(vmcnt=0), isa.waitcnt(lgkmcnt=0), isa.nop(15), isa.nop(15)]
        return o

    def epilogue(self) -> List[Instr]:
        """y[m][n .. n+3]: rows m = m0 + 128 wm + 32 mb + (lane & 31), columns n = n0 + 64 wn + 32 nb + 8 rr + 4 g."""
        c = self.cfg
        e: List[Instr] = []
        ql, g, t = T_[1], T_[2], T_
        nw = ST[4]
        e += [isa.sop("s_lshl_b32", nw, S_WN, I32(6)), isa.sop("s_add_u32", nw, nw, S_N0T)]
        e += self.addr64_madd(S_Y, nw, I32(1), 1)
        BQ = [[V(nb * 16 + rr * 4, 4) for rr in range(4)] for nb in range(2)]          # v0..31
        for i in range(32):
            e.append(isa.vop("v_mov_b32", V(i), I32(0)))
        e += [isa.sop("s_cmp_eq_u64", None, S_BIAS, I32(0)), isa.branch("s_cbranch_scc1", "L_nobias")]
        e += self.addr64_madd(S_BIAS, nw, I32(1), 2)
        e += [isa.vop("v_lshlrev_b32", t[3], I32(4), g)]
        for nb in range(2):
            for rr in range(4):
                e.append(isa.global_load(4, BQ[nb][rr], t[3], (32 * nb + 8 * rr) * 4, saddr=S_BIAS))
        e += [isa.waitcnt(vmcnt=0), isa.label("L_nobias")]
        mw, ldcb = ST[5], ST[6]
        e += [isa.sop("s_lshl_b32", mw, S_WM, I32(7)), isa.sop("s_add_u32", mw, mw, S_M0T), isa.sop("s_lshl_b32", ldcb, S_LDC.sub(0), I32(1))]
        RCP, KC0, KC1 = V(32), V(33), V(34)
        if c.epi in (3, 4):
            e += self.addr64_madd(S_RES, nw, I32(1), 1)
            e += [isa.sop("s_lshl_b32", ST[7], S_LDR.sub(0), I32(1))]
        if c.epi == 3:
            e += self.addr64_madd(S_GATE, nw, I32(1), 2)
            e += [isa.vop("v_cvt_f32_u32", RCP, S_RPB), isa.vop("v_rcp_f32", RCP, RCP),
                  isa.sop("s_cmp_eq_u32", None, S_RPB, I32(0)), isa.sop("s_cselect_b32", ST[9], I32(0), I32(0xFFFFFFFF)),
                  isa.sop("s_lshl_b32", ST[12], S_GS.sub(0), I32(2))]
        if c.epi == 1:
            K0, K1, sc = 0.7978845608028654, 0.044715, 2.0 * 1.4426950408889634
            e += [isa.vop("v_mov_b32", KC0, F32(K0 * sc)), isa.vop("v_mov_b32", KC1, F32(K0 * K1 * sc))]
        for mb in range(4):
            row, yoff, roff, goff, bq = V(35), V(36), V(37), V(38), V(39)
            e += [isa.vop("v_add_u32", row, mw, ql)]
            if mb:
                e += [isa.vop("v_add_u32", row, I32(32 * mb), row)]
            e += [isa.vop("v_mul_lo_u32", yoff, row, ldcb), isa.vop("v_lshl_add_u32", yoff, g, I32(3), yoff)]
            if c.epi in (3, 4):
                e += [isa.vop("v_mul_lo_u32", roff, row, ST[7]), isa.vop("v_lshl_add_u32", roff, g, I32(3), roff)]
            if c.epi == 3:
                e += [isa.vop("v_cvt_f32_u32", bq, row), isa.vop("v_add_f32", bq, F32(0.5), bq), isa.vop("v_mul_f32", bq, bq, RCP),
                      isa.vop("v_cvt_u32_f32", bq, bq), isa.vop("v_and_b32", bq, ST[9], bq),
                      isa.vop("v_mul_lo_u32", goff, bq, ST[12]), isa.vop("v_lshl_add_u32", goff, g, I32(4), goff)]
            e += [isa.v_cmp("v_cmp_lt_u32", row, S_M),
                  Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)]
            k = 0
            for nb in range(2):
                for rr in range(4):
                    base = 40 + 16 * (k % 2)
                    k += 1
                    f = [V(base + i) for i in range(4)]
                    w, rp, r_, u2, gq = V(base + 4, 2), V(base + 6, 2), V(base + 8), [V(base + 9), V(base + 10)], V(base + 12, 4)
                    noff = 32 * nb + 8 * rr
                    if c.epi in (3, 4):
                        e.append(isa.global_load(2, rp, roff, noff * 2, saddr=S_RES, extra_reads=[EXEC]))
                    if c.epi == 3:
                        e.append(isa.global_load(4, gq, goff, noff * 4, saddr=S_GATE, extra_reads=[EXEC]))
                    for i in range(4):
                        e += [isa.vop("v_accvgpr_read_b32", f[i], ACC(nb, mb).sub(4 * rr + i)), isa.vop("v_add_f32", f[i], f[i], BQ[nb][rr].sub(i))]
                    if c.epi == 1:
                        for i in range(4):
                            u = u2[i & 1]
                            e += [isa.vop("v_mul_f32", u, f[i], f[i]), isa.vop("v_fma_f32", u, u, KC1, KC0), isa.vop("v_mul_f32", u, u, f[i]),
                                  isa.vop("v_exp_f32", u, u), isa.vop("v_add_f32", u, F32(1.0), u), isa.vop("v_rcp_f32", u, u),
                                  isa.vop("v_fma_f32", f[i], Neg(f[i]), u, f[i])]
                    if c.epi in (3, 4):
                        e.append(isa.waitcnt(vmcnt=0))
                        if c.epi == 3:
                            for i in range(4):
                                e.append(isa.vop("v_mul_f32", f[i], f[i], gq.sub(i)))
                        for i in range(4):
                            src = rp.sub(i >> 1)
                            e += [isa.vop("v_lshlrev_b32", r_, I32(16), src) if (i & 1) == 0 else isa.vop("v_and_b32", r_, I32(0xFFFF0000), src),
                                  isa.vop("v_add_f32", f[i], f[i], r_)]
                    e += [isa.vop("v_cvt_pk_bf16_f32", w.sub(0), f[0], f[1]), isa.vop("v_cvt_pk_bf16_f32", w.sub(1), f[2], f[3]),
                          isa.global_store(2, yoff, w, noff * 2, saddr=S_Y, extra_reads=[EXEC])]
            e += [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]
        e += [isa.label("L_exit"), isa.waitcnt(vmcnt=0), Instr("s_endpgm", cls=isa.BRANCH)]
        return sched.pad_hazards(e)

    def program(self) -> List[Instr]:
        prog = self.prologue() + self.loop() + self.epilogue()
        pre = f"L_{self.cfg.name}"
        for i in prog:
            if i.label and i.label.startswith("L_"):
                new = pre + i.label[1:]
                if getattr(i, "text", None):
                    i.text = i.text.replace(i.label, new)
                i.label = new
        return prog


HEAD = """// GENERATED by scail_amd/asmgen/gemm8.py -- do not edit; regenerate with `python -m scail_amd.asmgen.gemm8`.
// Hand-scheduled 8-wave bf16 GEMM for gfx950 (256 x 256 x 64 tile, two waves per SIMD, 128 accumulators per lane in a[0:127]).
\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
"""


def kernel_text(c: Cfg) -> str:
    body = isa.render(Gen(c).program())
    return f"""// ---- kernel {c.name}: epilogue {c.epi} ----
\t.text
\t.protected\t{c.name}
\t.globl\t{c.name}
\t.p2align\t8
\t.type\t{c.name},@function
{body}.L{c.name}_end:
\t.size\t{c.name}, .L{c.name}_end-{c.name}
\t.section\t.rodata,"a",@progbits
\t.p2align\t6, 0x0
\t.amdhsa_kernel {c.name}
\t\t.amdhsa_group_segment_fixed_size 131072
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {KERNARG_SIZE}
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 1
\t\t.amdhsa_system_sgpr_workgroup_id_z 1
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr 256
\t\t.amdhsa_next_free_sgpr 96
\t\t.amdhsa_accum_offset 128
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_round_mode_16_64 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
"""


def metadata(cfgs) -> str:
    ks = "".join(f"""  - .agpr_count:     128
    .args:
      - .offset:         0
        .size:           {KERNARG_SIZE}
        .value_kind:     by_value
    .group_segment_fixed_size: 131072
    .kernarg_segment_align: 8
    .kernarg_segment_size: {KERNARG_SIZE}
    .max_flat_workgroup_size: 512
    .name:           {c.name}
    .private_segment_fixed_size: 0
    .sgpr_count:     102
    .sgpr_spill_count: 0
    .symbol:         {c.name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     256
    .vgpr_spill_count: 0
    .wavefront_size: 64
""" for c in cfgs)
    return f"""\t.amdgpu_metadata
---
amdhsa.kernels:
{ks}amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
\t.end_amdgpu_metadata
"""


def assembly(cfgs) -> str:
    return HEAD + "".join(kernel_text(c) for c in cfgs) + metadata(cfgs)


DEFAULTS = [Cfg(epi=e, name=f"scail_gemm8_e{e}") for e in (0, 1, 3, 4)]


def variant_cfgs():
    out = [Cfg(epi=0, cap=2, name="scail_gemm8_e0_c2"), Cfg(epi=0, cap=5, name="scail_gemm8_e0_c5"),
           Cfg(epi=0, wr_step=1.0, ld_from=28.0, name="scail_gemm8_e0_wr1"),
           Cfg(epi=0, ld_from=30.0, ld_step=0.3, name="scail_gemm8_e0_ldlate")]
    for abl in ("dma", "lds", "bar", "dma,lds"):
        out.append(Cfg(epi=0, abl=abl, name="scail_gemm8_e0_abl_" + abl.replace(",", "_")))
    return out


def main():
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(os.path.dirname(here)), "build_abl", "gemm8.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    text = assembly(DEFAULTS)
    if "--check" in sys.argv:
        sys.exit(0 if open(out).read() == text else 1)
    if not os.path.exists(out) or open(out).read() != text:
        open(out, "w").write(text)
    print(out, len(text.splitlines()), "lines")


if __name__ == "__main__":
    main()
