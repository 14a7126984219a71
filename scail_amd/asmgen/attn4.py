"""attn4 -- hand-scheduled flash attention for the SCAIL DiT self-attention on gfx950 (generator of csrc/attn4.s).

Reference call site: F.scaled_dot_product_attention in sat/transformer_defaults.py:67-72 reached from
dit_video_crossattn_sc_xc.py:1058-1105 (no mask, scale 1/sqrt(128)); same C entry point as the 8-wave kernel of
csrc/attn.hip (scail_flash_attn_bf16), which selects the kernel generated here for Lk >= 512 without accumulate.

Two kernel families come out of this generator:
  * ``M16F`` = scail_attn4_m16f, THE SHIPPED KERNEL (Cfg(mi=16, fold, lsum, ragged, qscale, opt, pv_qb)): 16x16x32 MFMAs, scores in
    log2 units (q arrives multiplied by scale * log2 e, or the prologue multiplies the Q fragments for raw-scale callers), the
    reference maximum folded into the accumulator init of the first QK^T MFMA, row sums on the matrix pipe, any key count, an
    optimistic hot loop that tracks no maximum and is verified on the row sums (see Cfg.opt), P.V in query-block-major order;
  * ``DEFAULT`` = scail_attn4, the 32x32x16 kernel of round 2 (scale per score, lazy running maximum in every iteration): measurement
    build and emulator tests only.
The description below is the common structure (register map of the 32x32x16 form; the 16x16x32 maps are the ``*16`` helpers).

Shape of the kernel (MI355X guide: "4-wave, one-wave-per-SIMD" attention structure):
  * workgroup = 256 query rows = 4 waves x 64 rows; ONE wave per SIMD with the whole 512-register file:
      AGPR  a[0:127]   O^T accumulators  [4 d-blocks][2 row-blocks] x 16
            a[128:191] Q fragments       [2 row-blocks][8 k-steps] x 4   (loaded once)
            a[192:255] K fragments of the next key tile [2 key-blocks][8 k-steps] x 4
      VGPR  v[0:63] / v[64:127]  two score tiles S (64 keys x 64 rows): the tile being exponentiated and the tile the QK^T
            MFMAs are writing; the bf16 P fragments overwrite the score registers they came from
            v[128:191] V^T fragments [4 d-blocks][4 k-steps] x 4,  v[192:] addresses, softmax state, temporaries
  * MFMA formulation identical to attn.hip: S^T = K Q^T (lane = query row, registers = keys), O^T += V^T P^T with the SAME
    lane's score registers as B operand (V^T stored with key bits 2<->3 swapped inside 16-key groups: scail_transpose_v).
  * K / V^T tiles of 64 keys arrive by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, XOR-swizzled on the
    SOURCE address, same involution on the fragment reads) into rings of RD slots, K RD+1 tiles ahead and V^T RD-1; one
    s_barrier per tile; counted vmcnt leaves (RD-2) tiles of DMA in flight across it.
  * software pipeline per tile t (64 MFMAs):  phase A: QK^T(t+1)  ||  exp / sum / bf16-pack of S(t)  ||  V^T(t) fragment reads
                                              phase B: P(t) V(t)  ||  row max of S(t+1)             ||  K(t+2) fragment reads
    The non-MFMA instructions are placed into the gaps between MFMAs by sched.schedule (<= CAP per gap).
  * online softmax with a LAZY running max: the O / l rescale (an out-of-line subroutine) runs only when some row's tile
    maximum exceeds the running maximum by more than ``thr`` (kernel argument; raw-score units for DEFAULT, log2 units for the fold
    kernels): exact arithmetic otherwise, and on random or real data it fires once per block.  M16F's hot loop drops even that check
    (Cfg.opt) and keeps it in the remainder iterations and in the second pass of a workgroup whose optimistic pass overflowed.

Limits (the C entry falls back to the 8-wave kernel otherwise): head_dim 128, no accumulate, Lk >= 512 (any count for M16F, whole
64-key tiles for DEFAULT); Lq * q_rs, Lk * k_rs and 128 * Lkp below 2^31 elements (32-bit byte offsets inside one (batch, head) slice);
(Lq / 256)^2 * heads * batch below 2^31 (reciprocal-multiplication decode of the workgroup id).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import List

from . import isa, sched
from .isa import A, S, V, I32, F32, Neg, VCC, EXEC, M0, Instr

KERNARG_SIZE = 168
# q k vt o | q_bs q_rs k_ss k_bs k_rs vt_ss vt_bs o_bs o_rs | heads Lq Lk Lkp n_seg | sl2 thr | nqb magic_nqb magic_heads xcd_mode |
# items_per_xcd (xcd_mode 2) n_items item0 (xcd_modes 0 / 2: this launch covers items [item0, item0 + n_items) of the pair-major list) |
# restarts: optional device pointer to a u32 counter (0 = none), incremented once per WORKGROUP that leaves the optimistic pass and runs
# again with the lazy-maximum loop (read in the restart path only: nothing in the hot loop)
KERNARG_FMT = "<4Q9q5iffiIIiiiiQ"
KARG_RESTARTS = 160


# x2 kernels: appended to the base block -- k2 vt2 | k2_bs vt2_bs | Lk2 Lkp2 | n_wgs pad
X2_EXTRA_FMT = "<2Q2q3i4x"
X2_KERNARG_SIZE = KERNARG_SIZE + struct.calcsize(X2_EXTRA_FMT)


def pack_args_x2(q, k1, vt1, k2, vt2, o, q_bs, q_rs, k1_bs, k_rs, vt1_bs, k2_bs, vt2_bs, o_bs, o_rs, heads, Lq, Lk1, Lkp1, Lk2, Lkp2, sl2, thr,
                 n_batch, n_wgs, rows=256) -> bytes:
    base = pack_args(q, k1, vt1, o, q_bs, q_rs, 0, k1_bs, k_rs, 0, vt1_bs, o_bs, o_rs, heads, Lq, Lk1, Lkp1, 1, sl2, thr, n_batch=n_batch, mode=0, rows=rows)
    b = base + struct.pack(X2_EXTRA_FMT, k2, vt2, k2_bs, vt2_bs, Lk2, Lkp2, n_wgs)
    assert len(b) == X2_KERNARG_SIZE
    return b


def magic31(d: int) -> int:
    """x // d == (2 x * magic31(d)) >> 32 for the small x the kernel divides (x * d < 2^31)."""
    return -(-(1 << 31) // d)


def grid_blocks(n_batch: int, heads: int, Lq: int, rows: int = 256, mode: int = None) -> int:
    items = ((Lq + rows - 1) // rows) * heads * n_batch
    mode = xcd_mode(n_batch, heads) if mode is None else mode
    return 8 * ((items + 7) // 8) if mode == 2 else items


def grid_for(n_items: int, mode: int) -> int:
    """workgroups of a launch over n_items consecutive items (a launch may cover a part of the item list: pack_args item0 / n_items)"""
    return 8 * ((n_items + 7) // 8) if mode == 2 else n_items


def xcd_mode(n_batch: int, heads: int) -> int:
    """1: workgroup id -> (XCD = id % 8 works on (batch, head) pairs = XCD mod 8), so the 32 CUs of an XCD stream the SAME K / V^T
    through their L2 (the hardware places consecutive workgroup ids on consecutive XCDs); needs pairs % 8 == 0.
    2 (any pair count): the (pair, query block) items in pair-major order are cut into 8 equal runs, XCD x walks run x
    (item = x * ceil(items / 8) + id / 8; ids past the end of a run exit at once), so an XCD streams one pair's K / V^T at a time
    (two where a run crosses a pair boundary) instead of all 8 XCDs streaming the same pair as the plain decode (0) does."""
    return 1 if (n_batch * heads) % 8 == 0 else 2


def pack_args(q, k, vt, o, q_bs, q_rs, k_ss, k_bs, k_rs, vt_ss, vt_bs, o_bs, o_rs, heads, Lq, Lk, Lkp, n_seg, sl2, thr, n_batch=1,
              mode=None, rows: int = 256, item0: int = 0, n_items: int = None, restarts: int = 0) -> bytes:
    """item0 / n_items (xcd_modes 0 and 2): the launch covers items [item0, item0 + n_items) of the pair-major (pair, query block) list --
    how scail_flash_attn_bf16 splits one attention into a 256-row launch of whole rounds and a 192-row launch for the rest."""
    nqb = (Lq + rows - 1) // rows
    mode = xcd_mode(n_batch, heads) if mode is None else mode
    items = nqb * heads * n_batch - item0 if n_items is None else n_items
    assert item0 == 0 or mode != 1
    b = struct.pack(KERNARG_FMT, q, k, vt, o, q_bs, q_rs, k_ss, k_bs, k_rs, vt_ss, vt_bs, o_bs, o_rs, heads, Lq, Lk, Lkp, n_seg, sl2, thr,
                    nqb, magic31(nqb), magic31(heads), mode, (items + 7) // 8, items, item0, restarts)
    assert len(b) == KERNARG_SIZE
    return b


@dataclass
class Cfg:
    rd: int = 4            # ring depth (2 or 4); unroll = max(2, rd)
    cap: int = 5           # fillers per MFMA gap
    lookahead: float = 1.0
    name: str = "scail_attn4"
    dma_k_at: float = 1.0  # MFMA gap of the first K piece, pieces dma_step gaps apart; V^T pieces from dma_v_at
    dma_v_at: float = 9.0
    dma_step: float = 2.0
    sm_end: float = 54.0   # the exp / sum / pack stream is spread over gaps [0, sm_end]
    sm_group: int = 1      # 1: element by element (dependent neighbours); 8 / 16: stage by stage over 8 / 16 elements
    max_chains: int = 1    # independent partial-maximum chains per row block in the row max
    abl: str = ""          # TIMING ABLATIONS (wrong results; ablation build only): "dma" / "lds" / "valu" / "bar" / "max" removed
    mi: int = 32           # MFMA shape: 32 = v_mfma_f32_32x32x16_bf16 (64 per tile), 16 = v_mfma_f32_16x16x32_bf16 (128 per tile)
    ragged: bool = False   # (mi = 16) any key count: the last tile of a segment may hold 1..64 valid keys.  Its out-of-range K rows are
                           # fetched from 64 rows earlier (in bounds), their scores are set to -FLT_MAX before the row maximum; both only in
                           # the remainder iterations of a segment (the hot loop never touches the last tile)
    lsum: bool = False     # (fold) row sums on the matrix pipe: a 9th "d block" whose V^T fragment is a constant row of ones accumulates
                           # sum_k P[k][q] beside O (8 extra MFMAs per tile replace 64 v_add_f32)
    fold: bool = False     # (mi = 16) q arrives multiplied by scale * log2(e): the running maximum is folded into the accumulator
                           # init of the first QK^T MFMA (S' = S - M), so p = exp2(S') needs no scale / shift instruction
    qscale: bool = False   # (fold) a caller with a RAW scale (kernel argument sl2 != 0) gets its Q fragments multiplied by sl2 = scale *
                           # log2(e) once in the prologue (bf16 -> fp32 -> x sl2 -> bf16, 448 instructions per workgroup); sl2 == 0 = q
                           # arrives in log2 units already (scail_rmsnorm_rope_scaled) and the block is skipped
    opt: bool = False      # (fold + lsum) OPTIMISTIC hot loop: the reference point M of exp2(S - M) is fixed after the first tile
                           # (tile maximum + ``head`` log2 units of headroom) and the hot loop neither tracks the row maximum nor
                           # branches -- exact in floating point as long as nothing overflows (bf16 P and the fp32 accumulators keep
                           # their relative precision at any magnitude).  The epilogue checks the row sums l (accumulated on the
                           # matrix pipe): if any row of the workgroup has l >= 2^60 or NaN the WHOLE workgroup runs again with the
                           # lazy-maximum hot loop (mode 1), which is correct for any input
    head: float = 40.0     # (opt) headroom: exp2 overflows only when a score exceeds the first tile's row maximum by > 127 + head
    pv_qb: bool = False    # (mi = 16) P.V MFMAs in query-block-major order (qb, db) instead of (db, qb): the P quad of query block qb is
                           # first needed 8 qb MFMAs into its k-step instead of within the first four, so the exp / pack stream can
                           # run evenly over the whole tile (sm_end ~ 58) without the forced clusters in front of a k-step
    v_at: float = 2.0      # V^T(t) fragment reads: first at this gap unit, v_step apart (needed by the P.V MFMAs of their k-step)
    v_step: float = 1.5
    k_at: float = 18.0     # K(t+2) fragment reads: first at this gap unit (after the QK^T MFMAs that still read the old fragments), k_step apart
    k_step: float = 2.0
    late_extra: float = -1.0   # >= 0: sched.schedule(late_extra=...) -- one filler beyond ``cap`` in a gap when the stream is that many gaps late
    nq: int = 4            # (mi = 16, fold, lsum) 16-row query blocks per wave: 4 = 256-row workgroups; 3 = 192-row workgroups (102 MFMAs
                           # per tile instead of 136 beside the same K / V^T traffic): the launch shape of a sequence-parallel rank, where
                           # ceil(workgroups / CUs) x tile cost is lower for the shorter tile (scail_flash_attn_bf16 picks per launch)
    pksum: bool = False    # (fold, not lsum) row sums on the VALU with v_pk_add_f32 (two fp32 adds per instruction): 32 instructions per tile where lsum
                           # spends 8 of the tile's 136 MFMAs -- EXPERIMENT of round 5 (measurement build)
    x2: bool = False       # (M16F family) CROSS ATTENTION OVER TWO KEY SETS, persistent workgroups: o = bf16(bf16(softmax(q K1^T) V1) + softmax(q K2^T) V2)
                           # (dit_video_crossattn_sc_xc.py:1107-1203: text + CLIP image tokens).  One workgroup per CU walks over the (pair, query
                           # block) items id, id + n_wgs, ...; per item the Q fragments are loaded once and the key pipeline runs twice (set 0, set 1:
                           # own key count, own softmax state, own optimistic pass / restart); set 0's normalised output is stored as bf16 at its
                           # final place and read back (same lane, same address) by set 1's epilogue, which adds and stores.  Kernel arguments:
                           # the base block (set 0 in the k / vt fields, xcd_mode 0, n_items = all items) + X2_EXTRA
    align: int = 0         # .p2align of the hot-loop entry labels (0 = none): code placement A/B (guide: hand-asm streams are
                           # sensitive to a uniform shift of the instruction stream)

    @property
    def rows(self): return 64 * self.nq        # query rows of a workgroup (4 waves x nq blocks of 16)
    @property
    def unroll(self): return max(2, self.rd)
    @property
    def pk(self): return self.rd + 1       # K prefetch distance (tiles)
    @property
    def pv(self): return self.rd - 1
    @property
    def vm_keep(self): return 8 * (self.pk - 3)     # DMA instructions left in flight across the per-tile barrier
    @property
    def lds_bytes(self): return 2 * self.rd * 16384


# ---- register map -------------------------------------------------------------------------------
def O(db, rb): return A((db * 2 + rb) * 16, 16)
def Qf(rb, ks): return A(128 + (rb * 8 + ks) * 4, 4)
def Kf(kb, ks): return A(192 + (kb * 8 + ks) * 4, 4)
def Sb(i, kb, rb): return V(i * 64 + (kb * 2 + rb) * 16, 16)
def Vtf(db, ks): return V(128 + (db * 4 + ks) * 4, 4)


# mi = 16 (16 x 16 x 32 MFMAs): 16-query blocks qb 0..3, 16-key blocks kb 0..3, 16-d blocks db 0..7, 32-wide k-steps
def O16(db, qb): return A((db * 4 + qb) * 4, 4)
def Qf16(qb, ks): return A(128 + (qb * 4 + ks) * 4, 4)
def Kf16(kb, ks): return A(192 + (kb * 4 + ks) * 4, 4)
# the two key blocks of a 32-key step sit in 8 consecutive registers, so their packed bf16 P quad (the B operand of P.V) is built in place
def Sb16(i, kb, qb): return V(i * 64 + ((kb >> 1) * 4 + qb) * 8 + (kb & 1) * 4, 4)
def Pq16(i, ks, qb): return V(i * 64 + (ks * 4 + qb) * 8, 4)
def Vtf16(db, ks): return V(128 + (db * 2 + ks) * 4, 4)


KADDR16 = [[V(192 + ks * 2 + par) for par in range(2)] for ks in range(4)]       # [k-step][parity of the key block]
VADDR16 = [V(200), V(201)]
M16 = [V(212 + i) for i in range(4)]
MC16 = [V(216 + i) for i in range(4)]
L16 = [[V(220 + qb * 2 + j) for j in range(2)] for qb in range(4)]
MX16 = [V(228 + i) for i in range(4)]
ALPHA16 = [V(232 + i) for i in range(4)]
OOFF16 = [V(236 + i) for i in range(4)]
ROW16 = [V(240 + i) for i in range(4)]
TMP16 = [V(244 + i) for i in range(8)] + [V(202), V(203)]      # 10 temporaries inside the loop (the epilogue uses the dead S registers)

# fold: -M of query block qb as an accumulator-init quad, row-sum partials; M16 / MC16 / MX16 / ALPHA16 do not exist
NEGM = [V(212 + 4 * qb, 4) for qb in range(4)]
L16F = [[V(228 + qb * 2 + j) for j in range(2)] for qb in range(4)]
# lsum: accumulator quads of the ones-row product (row 0 = lanes 0-15, register 0 holds the row sums) and the constant A fragment
LACC = [V(228, 4), V(232, 4), V(236, 4), V(240, 4)]           # = L16F + OOFF16 + ROW16 (recomputed in the epilogue)
ONES = V(248, 4)
TMPL = [V(244 + i) for i in range(4)] + [V(202), V(203)]      # the 6 temporaries left inside the loop
S_CLAMP, S_FIRST = S(87), S(88)
S_TAILREL = [S(89 + i) for i in range(4)]     # ragged: valid keys of the last tile - first tile row of this wave's K piece i
S_MODE, S_HEAD = S(94), S(95)     # opt: 0 = optimistic hot loop / 1 = lazy-maximum hot loop after a restart; headroom of the first maximum
S_TAIL = S(93)       # rescale subroutine: lower bound of the maximum step (0, -inf at the very first tile), first-call flag

KADDR = [V(192 + i) for i in range(8)]
VADDR = [V(200 + i) for i in range(4)]
KDMA = [V(204 + i) for i in range(4)]
VDMA = [V(208 + i) for i in range(4)]
M_ = [V(212), V(213)]
MC = [V(214), V(215)]
L_ = [[V(216 + rb * 4 + j) for j in range(4)] for rb in range(2)]
MX = [V(224), V(225)]
ALPHA = [V(226), V(227)]
TMP = [V(228 + i) for i in range(20)]        # v228..v247
OOFF = [V(248), V(249)]
ROW = [V(250), V(251)]
LANE, VT0, VT1, VT2 = V(252), V(253), V(254), V(255)

# SGPRs
S_KARG = S(0, 2)
S_QB, S_H, S_B = S(2), S(3), S(4)
S_Q, S_K, S_VT, S_O = S(8, 2), S(10, 2), S(12, 2), S(14, 2)
S_QBS, S_QRS, S_KSS, S_KBS = S(16, 2), S(18, 2), S(20, 2), S(22, 2)
S_KRS, S_VTSS, S_VTBS, S_OBS = S(24, 2), S(26, 2), S(28, 2), S(30, 2)
S_ORS = S(32, 2)
S_HEADS, S_LQ, S_LK, S_LKP, S_NSEG, S_C, S_THR, S_NQB = S(36), S(37), S(38), S(39), S(40), S(41), S(42), S(43)
S_MAGQ, S_MAGH, S_XMODE = S(84), S(85), S(86)
S_ITEM, S_SET, S_NWG, S_NIT = S(5), S(6), S(7), S(34)     # x2: current item, key set (0 / 1), workgroups of the launch, items of the launch
S_IPX, S_NITEMS, S_ITEM0 = S(87), S(88), S(89)     # xcd_mode 2: items per XCD run; items of this launch, its first item (live in the id decode only: s87.. are S_CLAMP / S_FIRST / S_TAILREL later)
S_KRSRC, S_VRSRC = S(44, 4), S(48, 4)
S_KOFF, S_VOFF, S_KSTEP, S_VSTEP, S_KMAX, S_VMAX = S(52), S(53), S(54), S(55), S(56), S(57)
S_T, S_NT, S_SEG, S_WAVE = S(58), S(59), S(60), S(61)
S_KLDS, S_VLDS = S(62), S(63)
S_RET = S(64, 2)
ST = [S(66 + i) for i in range(14)]          # s66..s79 temporaries
S_SAVE = S(80, 2)


class Gen:
    def __init__(self, cfg: Cfg):
        self.cfg = cfg
        self.out: List[Instr] = []

    def emit(self, x):
        if isinstance(x, Instr):
            self.out.append(x)
        else:
            self.out.extend(x)

    # =============================================================================================
    # building blocks (each returns a list in PROGRAM ORDER)
    # =============================================================================================
    def qk_mfmas(self, nxt: int) -> List[Instr]:
        """S_next[kb][rb] = sum_ks K[kb][ks] x Q[rb][ks]  (key-block major: S[0][*] completes after 16 MFMAs)."""
        out = []
        for kb in range(2):
            for ks in range(8):
                for rb in range(2):
                    d = Sb(nxt, kb, rb)
                    out.append(isa.mfma(d, Kf(kb, ks), Qf(rb, ks), I32(0) if ks == 0 else d, tag="qk"))
        return out

    def pv_mfmas(self, cur: int) -> List[Instr]:
        """O[db][rb] += V^T[db][ks] x P[rb][ks];  P[rb][ks] = S_cur[ks >> 1][rb] registers 8 (ks & 1) .. +3 (packed bf16)."""
        out = []
        for ks in range(4):
            for db in range(4):
                for rb in range(2):
                    p = Sb(cur, ks >> 1, rb).sub(8 * (ks & 1), 4)
                    out.append(isa.mfma(O(db, rb), Vtf(db, ks), p, O(db, rb), tag="pv"))
        return out

    def softmax_finish(self, cur: int, t0: float, t1: float) -> List[Instr]:
        """p = exp2(s c - m c), l += p, pack pairs to bf16 in place (k-step major = the order P.V consumes them)."""
        out = []
        nofma = "fma" in self.cfg.abl.split(",")
        for ks in range(4):
            kb, r0 = ks >> 1, 8 * (ks & 1)
            if self.cfg.sm_group <= 1:
                # element by element: fma -> exp -> add back to back (every instruction waits for the one before it)
                for rb in range(2):
                    s = Sb(cur, kb, rb)
                    for j in range(8):
                        r = s.sub(r0 + j)
                        if not nofma:
                            out.append(isa.vop("v_fma_f32", r, r, S_C, Neg(MC[rb])))
                        out.append(isa.vop("v_exp_f32", r, r))
                        out.append(isa.vop("v_add_f32", L_[rb][j & 3], L_[rb][j & 3], r))
                    for i in range(4):
                        out.append(isa.vop("v_cvt_pk_bf16_f32", s.sub(r0 + i), s.sub(r0 + 2 * i), s.sub(r0 + 2 * i + 1)))
                continue
            # software-pipelined by class: with ONE wave per SIMD nothing hides a dependent VALU's latency (v_exp_f32 is a
            # long-latency transcendental), so the k-step's 16 (or 8) elements go stage by stage: all scale-shifts, all exps,
            # all row-sum adds, all packs -- every consumer sits >= sm_group instructions behind its producer
            rbs = [(0, 1)] if self.cfg.sm_group >= 16 else [(0,), (1,)]
            for grp in rbs:
                el = [(rb, Sb(cur, kb, rb), j) for rb in grp for j in range(8)]
                if not nofma:
                    for rb, s, j in el:
                        out.append(isa.vop("v_fma_f32", s.sub(r0 + j), s.sub(r0 + j), S_C, Neg(MC[rb])))
                for rb, s, j in el:
                    out.append(isa.vop("v_exp_f32", s.sub(r0 + j), s.sub(r0 + j)))
                for rb, s, j in el:
                    out.append(isa.vop("v_add_f32", L_[rb][j & 3], L_[rb][j & 3], s.sub(r0 + j)))
                for rb in grp:
                    s = Sb(cur, kb, rb)
                    for i in range(4):
                        out.append(isa.vop("v_cvt_pk_bf16_f32", s.sub(r0 + i), s.sub(r0 + 2 * i), s.sub(r0 + 2 * i + 1)))
        n = len(out)
        for k, ins in enumerate(out):
            ins.target_gap = t0 + (t1 - t0) * k / n
        return out

    def rowmax(self, nxt: int, t_kb0: float, t_kb1: float, t_fin: float) -> List[Instr]:
        """MX[rb] = max over the 64 keys of S_next (both lanes of the row), then VCC = any(MX - M > thr)."""
        out = []
        nch = self.cfg.max_chains
        for rb in range(2):
            acc = TMP[rb]
            chains = [acc] + [TMP[10 + rb * 4 + c] for c in range(1, nch)]      # partial maxima (independent dependency chains)
            for kb, tg in ((0, t_kb0), (1, t_kb1)):
                s = Sb(nxt, kb, rb)
                lists = [[s.sub(r) for r in range(16)][c::nch] for c in range(nch)]
                steps: List[List[Instr]] = []
                for c, vals in enumerate(lists):
                    ops_c: List[Instr] = []
                    if kb == 0:
                        ops_c.append(isa.vop("v_max3_f32", chains[c], vals[0], vals[1], vals[2]))
                        vals = vals[3:]
                    while len(vals) >= 2:
                        ops_c.append(isa.vop("v_max3_f32", chains[c], chains[c], vals[0], vals[1]))
                        vals = vals[2:]
                    if vals:
                        ops_c.append(isa.vop("v_max_f32", chains[c], chains[c], vals[0]))
                    steps.append(ops_c)
                grp: List[Instr] = []
                for k in range(max(len(x) for x in steps)):          # round-robin over the chains
                    grp += [x[k] for x in steps if k < len(x)]
                if kb == 1 and nch > 1:                              # fold the partial maxima
                    rest = chains[1:]
                    while len(rest) >= 2:
                        grp.append(isa.vop("v_max3_f32", acc, acc, rest[0], rest[1]))
                        rest = rest[2:]
                    if rest:
                        grp.append(isa.vop("v_max_f32", acc, acc, rest[0]))
                span = 8.0 / max(len(grp), 1)
                for k, ins in enumerate(grp):
                    ins.target_gap = tg + rb * 0.4 * span + k * span
                out.extend(grp)
        fin = []
        for rb in range(2):
            acc, cp = TMP[rb], TMP[2 + rb]
            fin.append(isa.vop("v_mov_b32", cp, acc))
            fin.append(isa.permlane32_swap(acc, cp))                                  # acc = {lo, lo}, cp = {hi, hi}
            fin.append(isa.vop("v_max_f32", MX[rb], acc, cp))
            fin.append(isa.vop("v_sub_f32", TMP[4 + rb], MX[rb], M_[rb]))
        fin.append(isa.vop("v_max_f32", TMP[4], TMP[4], TMP[5]))
        fin.append(isa.v_cmp("v_cmp_gt_f32", TMP[4], S_THR))
        for k, ins in enumerate(fin):
            ins.target_gap = t_fin + 0.5 * k
        return out + fin

    # ---- mi = 16 --------------------------------------------------------------------------------------------------------------
    def qk_mfmas16(self, nxt: int) -> List[Instr]:
        """S_next[kb][qb] = sum_ks K[kb][ks] x Q[qb][ks], key-block major (S[0][*] completes after 16 MFMAs); the A fragment stays
        for 4 consecutive MFMAs."""
        out = []
        for kb in range(4):
            for ks in range(4):
                for qb in range(self.cfg.nq):
                    d = Sb16(nxt, kb, qb)
                    c0 = NEGM[qb] if self.cfg.fold else I32(0)
                    out.append(isa.mfma16(d, Kf16(kb, ks), Qf16(qb, ks), c0 if ks == 0 else d, tag="qk"))
        return out

    def pv_mfmas16(self, cur: int) -> List[Instr]:
        """O[db][qb] += V^T[db][ks] x P[qb][ks]; P[qb][ks] = the packed quad built in place from S[2 ks][qb], S[2 ks + 1][qb]."""
        out = []
        for ks in range(2):
            nq = self.cfg.nq
            order = ([(db, qb) for qb in range(nq) for db in range(8)] if self.cfg.pv_qb else
                     [(db, qb) for db in range(8) for qb in range(nq)])
            for db, qb in order:
                out.append(isa.mfma16(O16(db, qb), Vtf16(db, ks), Pq16(cur, ks, qb), O16(db, qb), tag="pv"))
            if self.cfg.lsum:
                for qb in range(self.cfg.nq):
                    out.append(isa.mfma16(LACC[qb], ONES, Pq16(cur, ks, qb), LACC[qb], tag="pv"))
        return out

    def softmax_finish16(self, cur: int, t0: float, t1: float) -> List[Instr]:
        out = []
        nofma = "fma" in self.cfg.abl.split(",")
        for ks in range(2):
            for qb in range(self.cfg.nq):
                base = Pq16(cur, ks, qb).idx
                regs = [V(base + j) for j in range(8)]
                lsum = L16F if self.cfg.fold else L16
                for j, r in enumerate(regs):
                    if not nofma and not self.cfg.fold:
                        out.append(isa.vop("v_fma_f32", r, r, S_C, Neg(MC16[qb])))
                    out.append(isa.vop("v_exp_f32", r, r))
                    if not self.cfg.lsum and not self.cfg.pksum:
                        out.append(isa.vop("v_add_f32", lsum[qb][j & 1], lsum[qb][j & 1], r))
                    if self.cfg.pksum and (j & 1):
                        # row sums on the VALU, two per instruction: l[0:1] += p[j-1 : j]  (32 v_pk_add_f32 per tile instead of the 8 ones-row MFMAs)
                        lp = V(lsum[qb][0].idx, 2)
                        out.append(isa.vop("v_pk_add_f32", lp, lp, V(base + j - 1, 2)))
                for i in range(4):
                    out.append(isa.vop("v_cvt_pk_bf16_f32", regs[i], regs[2 * i], regs[2 * i + 1]))
        n = len(out)
        for k, ins in enumerate(out):
            ins.target_gap = t0 + (t1 - t0) * k / n
        return out

    def rowmax16(self, nxt: int, t_kb, t_fin: float) -> List[Instr]:
        """MX16[qb] = max over the 64 keys of S_next (the 4 lanes l % 16 == q of a query), then VCC = any(MX - M > thr)."""
        out = []
        for kb in range(4):
            grp = []
            T = TMPL if self.cfg.lsum else TMP16
            for qb in range(self.cfg.nq):
                s = Sb16(nxt, kb, qb)
                acc = T[qb]
                if kb == 0:
                    grp.append(isa.vop("v_max3_f32", acc, s.sub(0), s.sub(1), s.sub(2)))
                    grp.append(isa.vop("v_max_f32", acc, acc, s.sub(3)))
                else:
                    grp.append(isa.vop("v_max3_f32", acc, acc, s.sub(0), s.sub(1)))
                    grp.append(isa.vop("v_max3_f32", acc, acc, s.sub(2), s.sub(3)))
            grp = grp[0::2] + grp[1::2]                     # the four rows' chains interleaved
            span = 12.0 * self.cfg.nq / 4.0 / len(grp)
            for k, ins in enumerate(grp):
                ins.target_gap = t_kb[kb] + k * span
            out.extend(grp)
        if self.cfg.fold:
            # S' is already relative to the running maximum: any lane's partial maximum above thr triggers the (rare) subroutine,
            # which does the cross-lane part
            T = TMPL if self.cfg.lsum else TMP16
            fin = [isa.vop("v_max3_f32", T[4], T[0], T[1], T[2])] + ([isa.vop("v_max_f32", T[4], T[4], T[3])] if self.cfg.nq == 4 else []) + [
                   isa.v_cmp("v_cmp_gt_f32", T[4], S_THR)]
            for k, ins in enumerate(fin):
                ins.target_gap = t_fin + 0.5 * k
            return out + fin
        fin = []
        for qb in range(self.cfg.nq):
            acc, cp = TMP16[qb], TMP16[4 + (qb & 1)]
            fin.append(isa.vop("v_mov_b32", cp, acc))
            fin.append(isa.permlane32_swap(acc, cp))                                  # acc = {lo, lo}, cp = {hi, hi}
            fin.append(isa.vop("v_max_f32", acc, acc, cp))
            fin.append(isa.vop("v_mov_b32", cp, acc))
            fin.append(isa.permlane16_swap(acc, cp))                                  # acc = even rows twice, cp = odd rows twice
            fin.append(isa.vop("v_max_f32", MX16[qb], acc, cp))
            fin.append(isa.vop("v_sub_f32", TMP16[6 + (qb & 1)], MX16[qb], M16[qb]))
            if qb & 1:
                fin.append(isa.vop("v_max_f32", TMP16[6], TMP16[6], TMP16[7]))
                if qb == 3:
                    fin.append(isa.vop("v_max_f32", TMP16[8], TMP16[8], TMP16[6]))
                else:
                    fin.append(isa.vop("v_mov_b32", TMP16[8], TMP16[6]))
        fin.append(isa.v_cmp("v_cmp_gt_f32", TMP16[8], S_THR))
        for k, ins in enumerate(fin):
            ins.target_gap = t_fin + 0.5 * k
        return out + fin

    def mask_scores16(self, nxt: int, tile_ahead: int, t_kb) -> List[Instr]:
        """ragged: scores of keys >= the valid count of tile S_T + tile_ahead are set to -FLT_MAX (exp2 -> 0, ignored by the row
        maximum).  Key of register e of block kb in lane group g: 32 (kb >> 1) + 8 (kb & 1) + e + G, G = 16 (g >> 1) + 4 (g & 1).
        VT1 holds -FLT_MAX (a literal beside VCC would exceed the constant-bus limit of v_cndmask_b32)."""
        valid, lim = ST[1], ST[2]
        G, g = TMPL[5], VT2
        t0 = t_kb[0] - 3.0
        out = [isa.sop("s_add_u32", valid, S_T, I32(tile_ahead), target_gap=t0),
               isa.sop("s_lshl_b32", valid, valid, I32(6), target_gap=t0 + 0.1),
               isa.sop("s_sub_u32", valid, S_LK, valid, target_gap=t0 + 0.2),
               isa.vop("v_and_b32", G, I32(1), g, target_gap=t0 + 0.3), isa.vop("v_lshlrev_b32", G, I32(2), G, target_gap=t0 + 0.4),
               isa.vop("v_lshrrev_b32", VT0, I32(1), g, target_gap=t0 + 0.5), isa.vop("v_lshl_add_u32", G, VT0, I32(4), G, target_gap=t0 + 0.6)]
        for kb in range(4):
            for e in range(4):
                tg = t_kb[kb] - 1.5 + 0.3 * e
                out += [isa.sop("s_sub_u32", lim, valid, I32(32 * (kb >> 1) + 8 * (kb & 1) + e), target_gap=tg),
                        isa.v_cmp("v_cmp_gt_i32", lim, G, target_gap=tg + 0.05)]                        # vcc = key is valid
                for qb in range(self.cfg.nq):
                    r = Sb16(nxt, kb, qb).sub(e)
                    out.append(isa.v_cndmask(r, VT1, r, target_gap=tg + 0.1))
        return out

    def v_frag_reads16(self, slot: int, t0: float, step: float) -> List[Instr]:
        out = []
        k = 0
        for ks in range(2):
            for db in range(8):
                out.append(isa.ds_read_b128(Vtf16(db, ks), VADDR16[ks], slot * 16384 + db * 2048, target_gap=t0 + step * k))
                k += 1
        return out

    def k_frag_reads16(self, slot: int, t0: float, step: float) -> List[Instr]:
        out = []
        k = 0
        for kb in range(4):
            for ks in range(4):
                out.append(isa.ds_read_b128(Kf16(kb, ks), KADDR16[ks][kb & 1], slot * 16384 + (kb >> 1) * 8192, target_gap=t0 + step * k))
                k += 1
        return out

    def v_frag_reads(self, slot: int, t0: float, step: float) -> List[Instr]:
        out = []
        k = 0
        for ks in range(4):
            for db in range(4):
                out.append(isa.ds_read_b128(Vtf(db, ks), VADDR[ks], slot * 16384 + db * 4096, target_gap=t0 + step * k))
                k += 1
        return out

    def k_frag_reads(self, slot: int, t0: float, step: float) -> List[Instr]:
        out = []
        k = 0
        for kb in range(2):
            for ks in range(8):
                out.append(isa.ds_read_b128(Kf(kb, ks), KADDR[ks], slot * 16384 + kb * 8192, target_gap=t0 + step * k))
                k += 1
        return out

    def dma_tile(self, which: str, slot: int, t0: float, step: float, careful: bool = False) -> List[Instr]:
        """4 LDS-DMA pieces of this wave for one K or V^T tile, then advance (and clamp) the stream offset.
        careful (ragged K tiles): if this is the segment's last tile, lanes whose tile row lies beyond the valid keys read the row
        64 rows earlier (in bounds; the scores of those keys are masked)."""
        lds, offs, rsrc, off, stp, mx = ((S_KLDS, KDMA, S_KRSRC, S_KOFF, S_KSTEP, S_KMAX) if which == "k" else
                                         (S_VLDS, VDMA, S_VRSRC, S_VOFF, S_VSTEP, S_VMAX))
        out = [isa.sop("s_add_u32", M0, lds, I32(slot * 16384), target_gap=t0 - 0.6)]
        if careful and which == "k":
            tmp, g, lim = VT0, VT2, ST[4]
            out.append(isa.sop("s_cmp_eq_u32", None, off, mx, target_gap=t0 - 0.5))
            for i in range(4):
                tg = t0 + step * i
                out += [isa.sop("s_cselect_b32", lim, S_TAILREL[i], I32(64), target_gap=tg - 0.4),
                        isa.v_cmp("v_cmp_le_i32", lim, g, target_gap=tg - 0.3),                      # vcc = row beyond the valid keys
                        isa.vop("v_subrev_u32", tmp, stp, offs[i], target_gap=tg - 0.2),              # offset of the row 64 rows earlier
                        isa.v_cndmask(tmp, offs[i], tmp, target_gap=tg - 0.1),
                        isa.buffer_load_lds(tmp, rsrc, off, 1024 * i, target_gap=tg, tag="dma")]
            out.append(isa.sop("s_add_u32", off, off, stp, target_gap=t0 + step * 3 + 0.3))
            out.append(isa.sop("s_min_u32", off, off, mx, target_gap=t0 + step * 3 + 0.6))
            return out
        for i in range(4):
            out.append(isa.buffer_load_lds(offs[i], rsrc, off, 1024 * i, target_gap=t0 + step * i, tag="dma"))
        out.append(isa.sop("s_add_u32", off, off, stp, target_gap=t0 + step * 3 + 0.3))
        out.append(isa.sop("s_min_u32", off, off, mx, target_gap=t0 + step * 3 + 0.6))
        return out

    # ---------------------------------------------------------------------------------------------
    def iter_block(self, p: int, tail: bool, careful: bool = False, nomax: bool = False) -> List[Instr]:
        """One pipelined tile iteration at unroll position p (tile t = p mod unroll), scheduled.  Gap numbers below are in units
        of 32 matrix-pipe cycles; mi = 16 has two MFMA gaps per unit (gs = 2)."""
        c = self.cfg
        cur, nxt = p & 1, (p + 1) & 1
        rd = c.rd
        m16 = c.mi == 16
        gs = (2.0 if m16 else 1.0) * c.nq / 4.0      # nq = 3: 102 MFMAs per tile, every target scales with the spine
        blk: List[Instr] = []
        abl = c.abl.split(",")
        careful = careful and c.ragged
        if not tail and "dma" not in abl:
            blk += self.dma_tile("k", (p + c.pk) % rd, c.dma_k_at * gs, c.dma_step * gs, careful=careful)
            blk += self.dma_tile("v", (p + c.pv) % rd, c.dma_v_at * gs, c.dma_step * gs)
        if "lds" not in abl:
            blk += (self.v_frag_reads16 if m16 else self.v_frag_reads)(p % rd, (c.v_at if not tail else 0.0) * gs, (c.v_step if not tail else 1.0) * gs)
        if not tail:
            blk += self.qk_mfmas16(nxt) if m16 else self.qk_mfmas(nxt)
            if careful:
                blk += self.mask_scores16(nxt, 1, tuple(x * c.nq / 4.0 for x in (20.0, 36.0, 52.0, 68.0)))
        if "valu" not in abl:
            blk += (self.softmax_finish16 if m16 else self.softmax_finish)(cur, 0.0, (c.sm_end if not tail else 20.0) * gs)
        blk += self.pv_mfmas16(cur) if m16 else self.pv_mfmas(cur)
        if not tail:
            if "lds" not in abl:
                blk += (self.k_frag_reads16 if m16 else self.k_frag_reads)((p + 2) % rd, c.k_at * gs, c.k_step * gs)
            if "valu" not in abl and "max" not in abl and not nomax:
                blk += (self.rowmax16(nxt, tuple(x * c.nq / 4.0 for x in (20.0, 36.0, 52.0, 68.0)), 84.0 * c.nq / 4.0) if m16 else
                        self.rowmax(nxt, 21.0, 38.0, 52.0))
        seq = sched.schedule(blk, cap=c.cap, lookahead=c.lookahead, late_extra=c.late_extra if c.late_extra >= 0 else None)
        seq = sched.insert_lgkm_waits(seq)
        return seq

    def align_directive(self) -> List[Instr]:
        if not self.cfg.align:
            return []
        d = Instr("label", label=None, cls=isa.LABEL)
        d.text = f".p2align {self.cfg.align.bit_length() - 1}"
        return [d]

    def iter_end(self, p: int, kind: str, nomax: bool = False) -> List[Instr]:
        """block end of a full iteration: fragment reads landed, DMA of the tile needed next landed, workgroup barrier, then the
        (rare) lazy-rescale call (nomax: the optimistic hot loop has neither)."""
        c = self.cfg
        skip = f"L_{kind}{p}_norescale"
        if "bar" in c.abl.split(","):
            return [isa.waitcnt(lgkmcnt=0), isa.waitcnt(vmcnt=c.vm_keep)]
        if "valu" in c.abl.split(",") or "max" in c.abl.split(",") or nomax:
            return [isa.waitcnt(lgkmcnt=0), isa.waitcnt(vmcnt=c.vm_keep), isa.barrier()]
        sub = f"L_rescale{(p + 1) & 1}" if c.fold else "L_rescale"
        return [isa.waitcnt(lgkmcnt=0), isa.waitcnt(vmcnt=c.vm_keep), isa.barrier(),
                isa.branch("s_cbranch_vccz", skip), isa.s_call(S_RET, sub), isa.label(skip)]

    # =============================================================================================
    def rescale_sub16(self) -> List[Instr]:
        out = [isa.label("L_rescale"), isa.nop(15), isa.nop(15)]
        for qb in range(self.cfg.nq):
            mn, d = TMP16[0 + (qb & 1)], TMP16[2 + (qb & 1)]
            out += [isa.vop("v_max_f32", mn, M16[qb], MX16[qb]),
                    isa.vop("v_sub_f32", d, M16[qb], mn),
                    isa.vop("v_mul_f32", d, d, S_C),
                    isa.vop("v_exp_f32", ALPHA16[qb], d),
                    isa.vop("v_mov_b32", M16[qb], mn),
                    isa.vop("v_mul_f32", MC16[qb], mn, S_C)]
            for j in range(2):
                out.append(isa.vop("v_mul_f32", L16[qb][j], L16[qb][j], ALPHA16[qb]))
        k = 0
        for db in range(8):
            for qb in range(self.cfg.nq):
                for r in range(4):
                    t = TMP16[4 + (k % 6)]
                    k += 1
                    o = O16(db, qb).sub(r)
                    out += [isa.vop("v_accvgpr_read_b32", t, o), isa.vop("v_mul_f32", t, t, ALPHA16[qb]),
                            isa.vop("v_accvgpr_write_b32", o, t)]
        out.append(isa.nop(3))
        out.append(Instr("s_setpc_b64", [], [S_RET], cls=isa.BRANCH))
        return sched.pad_hazards(out[:1]) + sched.pad_hazards(out[1:])

    def rescale_sub16f(self, nxt: int) -> List[Instr]:
        """fold: raise the running maxima by the (cross-lane) maximum of the pending score tile S'(t+1) in buffer ``nxt``:
        mx = max(rowmax, CLAMP);  S' -= mx, -M -= mx, and -- unless this is the first call of the kernel (O = l = 0, mx may be a huge
        negative number) -- l, O *= 2^-mx."""
        out = [isa.label(f"L_rescale{nxt}"), isa.nop(15), isa.nop(15)]
        lsum = self.cfg.lsum
        if lsum:      # the V^T fragments of the finished tile are dead at the call sites: v128.. serve as temporaries
            mx, cp, al, t2 = [TMPL[qb] for qb in range(4)], V(128), [V(129 + qb) for qb in range(4)], V(133)
        else:
            mx, cp, al, t2 = [TMP16[qb] for qb in range(4)], TMP16[4], [TMP16[5 + qb] for qb in range(4)], TMP16[9]
        for qb in range(self.cfg.nq):
            out += [isa.vop("v_mov_b32", cp, mx[qb]), isa.permlane32_swap(mx[qb], cp), isa.vop("v_max_f32", mx[qb], mx[qb], cp),
                    isa.vop("v_mov_b32", cp, mx[qb]), isa.permlane16_swap(mx[qb], cp), isa.vop("v_max_f32", mx[qb], mx[qb], cp),
                    isa.vop("v_max_f32", mx[qb], mx[qb], S_CLAMP)]
            if self.cfg.opt:          # first call of the optimistic pass: M = tile maximum + headroom (S_HEAD is 0 on every later call)
                out.append(isa.vop("v_add_f32", mx[qb], mx[qb], S_HEAD))
            for i in range(4):
                out.append(isa.vop("v_sub_f32", NEGM[qb].sub(i), NEGM[qb].sub(i), mx[qb]))
            for kb in range(4):
                for i in range(4):
                    r = Sb16(nxt, kb, qb).sub(i)
                    out.append(isa.vop("v_sub_f32", r, r, mx[qb]))
        out += [isa.sop("s_cmp_lg_u32", None, S_FIRST, I32(0)), isa.sop("s_mov_b32", S_FIRST, I32(0)), isa.sop("s_mov_b32", S_CLAMP, I32(0))]
        if self.cfg.opt:
            out.append(isa.sop("s_mov_b32", S_HEAD, I32(0)))
        out += [isa.branch("s_cbranch_scc1", f"L_rescale{nxt}_ret")]
        for qb in range(self.cfg.nq):
            out += [isa.vop("v_exp_f32", al[qb], Neg(mx[qb]))]
        for qb in range(self.cfg.nq):
            if lsum:
                for j in range(4):
                    out.append(isa.vop("v_mul_f32", LACC[qb].sub(j), LACC[qb].sub(j), al[qb]))
            else:
                for j in range(2):
                    out.append(isa.vop("v_mul_f32", L16F[qb][j], L16F[qb][j], al[qb]))
        k = 0
        for db in range(8):
            for qb in range(self.cfg.nq):
                for r in range(4):
                    t = [cp, t2][k % 2]
                    k += 1
                    o = O16(db, qb).sub(r)
                    out += [isa.vop("v_accvgpr_read_b32", t, o), isa.vop("v_mul_f32", t, t, al[qb]),
                            isa.vop("v_accvgpr_write_b32", o, t)]
        out += [isa.label(f"L_rescale{nxt}_ret"), isa.nop(3), Instr("s_setpc_b64", [], [S_RET], cls=isa.BRANCH)]
        i0 = next(k for k, x in enumerate(out) if x.label == f"L_rescale{nxt}_ret")
        return sched.pad_hazards(out[:1]) + sched.pad_hazards(out[1:i0]) + sched.pad_hazards(out[i0:])

    def rescale_sub(self) -> List[Instr]:
        if self.cfg.fold:
            return self.rescale_sub16f(0) + self.rescale_sub16f(1)
        if self.cfg.mi == 16:
            return self.rescale_sub16()
        out = [isa.label("L_rescale"), isa.nop(15), isa.nop(15)]
        for rb in range(2):
            mn, d = TMP[6 + rb], TMP[8 + rb]
            out += [isa.vop("v_max_f32", mn, M_[rb], MX[rb]),
                    isa.vop("v_sub_f32", d, M_[rb], mn),
                    isa.vop("v_mul_f32", d, d, S_C),
                    isa.vop("v_exp_f32", ALPHA[rb], d),
                    isa.vop("v_mov_b32", M_[rb], mn),
                    isa.vop("v_mul_f32", MC[rb], mn, S_C)]
            for j in range(4):
                out.append(isa.vop("v_mul_f32", L_[rb][j], L_[rb][j], ALPHA[rb]))
        k = 0
        for db in range(4):
            for rb in range(2):
                for r in range(16):
                    t = TMP[10 + (k % 8)]
                    k += 1
                    o = O(db, rb).sub(r)
                    out += [isa.vop("v_accvgpr_read_b32", t, o), isa.vop("v_mul_f32", t, t, ALPHA[rb]),
                            isa.vop("v_accvgpr_write_b32", o, t)]
        out.append(isa.nop(3))
        out.append(Instr("s_setpc_b64", [], [S_RET], cls=isa.BRANCH))
        return sched.pad_hazards(out[:1]) + sched.pad_hazards(out[1:])

    # =============================================================================================
    def addr64_madd(self, ptr: isa.Reg, stride64: isa.Reg, mult: isa.Reg, shift: int) -> List[Instr]:
        """ptr(64) += (stride64 << shift) * mult  (mult: 32-bit SGPR)."""
        t0, t1, t2, t3 = ST[0], ST[1], ST[2], ST[3]
        st = S(ST[4].idx, 2)
        return [isa.sop("s_lshl_b64", st, stride64, I32(shift)),
                isa.sop("s_mul_i32", t0, st.sub(0), mult), isa.sop("s_mul_hi_u32", t1, st.sub(0), mult),
                isa.sop("s_mul_i32", t2, st.sub(1), mult), isa.sop("s_add_u32", t1, t1, t2),
                isa.sop("s_add_u32", ptr.sub(0), ptr.sub(0), t0), isa.sop("s_addc_u32", ptr.sub(1), ptr.sub(1), t1)]

    def wave_row_base(self) -> List[Instr]:
        """ST[8] = first query row of the workgroup (rows x query block), ST[7] = first row of this wave inside it (16 nq x wave)."""
        if self.cfg.nq == 4:
            return [isa.sop("s_lshl_b32", ST[8], S_QB, I32(8)), isa.sop("s_lshl_b32", ST[7], S_WAVE, I32(6))]
        return [isa.sop("s_mul_i32", ST[8], S_QB, I32(self.cfg.rows)), isa.sop("s_mul_i32", ST[7], S_WAVE, I32(16 * self.cfg.nq))]

    def x2_set_block(self, krsb) -> List[Instr]:
        """x2: entry of a key set's pass (S_SET = 0 right after the item's Q loads, 1 after set 0's epilogue).  Set 1 takes its K / V^T
        pointers, batch strides and key counts from the appended kernel arguments; both sets (re)derive what depends on the key
        count: V^T row stride -> the V^T DMA lane offsets, tile count, clamps of the DMA offsets, the ragged tail.  Temporaries: score
        registers (dead between passes; TMP16 overlaps the ONES fragment, which is live from the first item on)."""
        t = [V(i) for i in range(8)]
        o: List[Instr] = [isa.sop("s_mov_b32", S_SET, I32(0)), isa.label("L_set"), isa.nop(15),
                          isa.sop("s_cmp_eq_u32", None, S_SET, I32(0)), isa.branch("s_cbranch_scc1", "L_set_args")]
        X = KERNARG_SIZE
        o += [isa.s_load(2, S_K, S_KARG, X), isa.s_load(2, S_VT, S_KARG, X + 8), isa.s_load(2, S_KBS, S_KARG, X + 16),
              isa.s_load(2, S_VTBS, S_KARG, X + 24), isa.s_load(2, S(S_LK.idx, 2), S_KARG, X + 32), isa.waitcnt(lgkmcnt=0)]
        o += self.addr64_madd(S_K, S_KBS, S_B, 1) + self.addr64_madd(S_VT, S_VTBS, S_B, 1)
        o += [isa.sop("s_lshl_b32", ST[6], S_H, I32(8)),
              isa.sop("s_add_u32", S_K.sub(0), S_K.sub(0), ST[6]), isa.sop("s_addc_u32", S_K.sub(1), S_K.sub(1), I32(0)),
              isa.sop("s_lshl_b32", ST[7], S_LKP, I32(8)), isa.sop("s_mul_i32", ST[8], ST[7], S_H), isa.sop("s_mul_hi_u32", ST[9], ST[7], S_H),
              isa.sop("s_add_u32", S_VT.sub(0), S_VT.sub(0), ST[8]), isa.sop("s_addc_u32", S_VT.sub(1), S_VT.sub(1), ST[9])]
        o += [isa.label("L_set_args"), isa.nop(3)]
        lkpb = ST[11]
        o += [isa.sop("s_lshl_b32", lkpb, S_LKP, I32(1)),
              isa.sop("s_lshr_b32", S_NT, S_LKP, I32(6)), isa.sop("s_sub_u32", ST[12], S_NT, I32(1)),
              isa.sop("s_mul_i32", S_KMAX, ST[12], S_KSTEP), isa.sop("s_lshl_b32", S_VMAX, ST[12], I32(7))]
        # V^T piece i: rows 32 w + 8 i + (lane >> 3), chunk (lane & 7) ^ ((row >> 1) & 7)   (as in the prologue)
        o += [isa.vop("v_lshrrev_b32", t[0], I32(3), LANE), isa.vop("v_and_b32", t[1], I32(7), LANE),
              isa.vop("v_lshlrev_b32", t[3], I32(5), S_WAVE)]
        for i in range(4):
            o += [isa.vop("v_add_u32", t[4], I32(8 * i), t[0]), isa.vop("v_add_u32", t[6], t[4], t[3]),
                  isa.vop("v_lshrrev_b32", t[5], I32(1), t[6]), isa.vop("v_and_b32", t[5], I32(7), t[5]),
                  isa.vop("v_xor_b32", t[5], t[1], t[5]),
                  isa.vop("v_mul_lo_u32", t[6], t[6], lkpb), isa.vop("v_lshl_add_u32", t[6], t[5], I32(4), t[6]),
                  isa.vop("v_subrev_u32", VDMA[i], I32(1024 * i), t[6])]
        o += [isa.sop("s_lshl_b32", ST[0], ST[12], I32(6)),
              isa.sop("s_sub_u32", S_TAIL, S_LK, ST[0]), isa.sop("s_lshl_b32", ST[1], S_WAVE, I32(4)), isa.sop("s_sub_u32", ST[1], S_TAIL, ST[1])]
        for i in range(4):
            o.append(isa.sop("s_sub_u32", S_TAILREL[i], ST[1], I32(4 * i)))
        return o

    def prologue(self) -> List[Instr]:
        c = self.cfg
        assert c.nq == 4 or (c.nq == 3 and c.mi == 16 and c.fold and c.lsum), "nq = 3 exists for the fold / lsum 16x16x32 kernels only"
        o: List[Instr] = [isa.label(c.name)]
        if c.x2:
            # one-time part (V0 = workitem id is overwritten by the score tiles later), then the item loop: every item reloads the kernel
            # arguments (the per-item pointer arithmetic and the set switch overwrite them) -- scalar-cache hits after the first item
            assert c.fold and c.lsum and c.opt and c.ragged and c.nq == 4
            o += [isa.vop("v_and_b32", LANE, I32(63), V(0)), isa.vop("v_lshrrev_b32", VT0, I32(6), V(0)),
                  isa.s_load(1, S_NWG, S_KARG, KERNARG_SIZE + 40), isa.s_load(1, S_NIT, S_KARG, 152),
                  isa.sop("s_mov_b32", S_ITEM, S(2)), isa.nop(3), isa.vop("v_readfirstlane_b32", S_WAVE, VT0), isa.waitcnt(lgkmcnt=0),
                  isa.label("L_item"), isa.nop(15)]
        o += [isa.s_load(8, S(8, 8), S_KARG, 0), isa.s_load(8, S(16, 8), S_KARG, 32), isa.s_load(8, S(24, 8), S_KARG, 64),
              isa.s_load(2, S_ORS, S_KARG, 96), isa.s_load(8, S(36, 8), S_KARG, 104)]
        if not c.x2:
            o += [isa.vop("v_and_b32", LANE, I32(63), V(0)), isa.vop("v_lshrrev_b32", VT0, I32(6), V(0))]
        o += [isa.s_load(4, S(84, 4), S_KARG, 136), isa.s_load(2, S(S_NITEMS.idx, 2), S_KARG, 152), isa.waitcnt(lgkmcnt=0)]
        o += [isa.sop("s_mov_b32", S(2), S_ITEM)] if c.x2 else [isa.vop("v_readfirstlane_b32", S_WAVE, VT0)]
        # ---- 1-D workgroup id -> (query block, head, batch); xcd_mode 1: ids congruent mod 8 (= one XCD) share (batch, head)
        #      pairs, so the 32 CUs of an XCD stream the same K / V^T tiles through their L2 ----
        wid, xcd, j, qd, pair, tt = S(2), ST[0], ST[1], ST[2], ST[3], ST[4]
        #      xcd_mode 2 (any pair count): item = xcd * items_per_xcd + id / 8 in pair-major order, ids past a run's end exit ----
        o += [isa.sop("s_and_b32", xcd, wid, I32(7)), isa.sop("s_lshr_b32", j, wid, I32(3)),
              isa.sop("s_cmp_eq_u32", None, S_XMODE, I32(2)), isa.branch("s_cbranch_scc0", "L_id_plain"),
              isa.sop("s_cmp_lt_u32", None, j, S_IPX), isa.branch("s_cbranch_scc1", "L_id_in_run"),
              Instr("s_endpgm", cls=isa.BRANCH), isa.label("L_id_in_run"),
              isa.sop("s_mul_i32", tt, xcd, S_IPX), isa.sop("s_add_u32", j, j, tt),
              isa.sop("s_cmp_lt_u32", None, j, S_NITEMS), isa.branch("s_cbranch_scc1", "L_id_plain"),
              Instr("s_endpgm", cls=isa.BRANCH), isa.label("L_id_plain"),
              isa.sop("s_cmp_lg_u32", None, S_XMODE, I32(0)), isa.sop("s_cselect_b32", j, j, wid),
              isa.sop("s_add_u32", j, j, S_ITEM0),                                                     # first item of this launch (0 in xcd_mode 1)
              isa.sop("s_lshl_b32", tt, j, I32(1)), isa.sop("s_mul_hi_u32", qd, tt, S_MAGQ),            # qd = j / nqb
              isa.sop("s_mul_i32", tt, qd, S_NQB), isa.sop("s_sub_u32", S_QB, j, tt),                   # qb = j % nqb
              isa.sop("s_lshl_b32", tt, qd, I32(3)), isa.sop("s_add_u32", tt, tt, xcd),
              isa.sop("s_cmp_eq_u32", None, S_XMODE, I32(1)), isa.sop("s_cselect_b32", pair, tt, qd),
              isa.sop("s_lshl_b32", tt, pair, I32(1)), isa.sop("s_mul_hi_u32", S_B, tt, S_MAGH),         # b = pair / heads
              isa.sop("s_mul_i32", tt, S_B, S_HEADS), isa.sop("s_sub_u32", S_H, pair, tt)]
        # ---- per (batch, head) base pointers ----
        o += self.addr64_madd(S_Q, S_QBS, S_B, 1) + self.addr64_madd(S_K, S_KBS, S_B, 1)
        o += self.addr64_madd(S_VT, S_VTBS, S_B, 1) + self.addr64_madd(S_O, S_OBS, S_B, 1)
        hb = ST[6]
        o += [isa.sop("s_lshl_b32", hb, S_H, I32(8))]                              # h * 128 elements * 2 bytes
        for ptr in (S_Q, S_K, S_O):
            o += [isa.sop("s_add_u32", ptr.sub(0), ptr.sub(0), hb), isa.sop("s_addc_u32", ptr.sub(1), ptr.sub(1), I32(0))]
        # vt += h * 128 * Lkp * 2 bytes
        o += [isa.sop("s_lshl_b32", ST[7], S_LKP, I32(8)), isa.sop("s_mul_i32", ST[8], ST[7], S_H), isa.sop("s_mul_hi_u32", ST[9], ST[7], S_H),
              isa.sop("s_add_u32", S_VT.sub(0), S_VT.sub(0), ST[8]), isa.sop("s_addc_u32", S_VT.sub(1), S_VT.sub(1), ST[9])]
        # ---- buffer descriptors (raw, no bounds: every address is inside the slice by construction) ----
        for rs, ptr in ((S_KRSRC, S_K), (S_VRSRC, S_VT)):
            o += [isa.sop("s_mov_b32", rs.sub(0), ptr.sub(0)), isa.sop("s_and_b32", rs.sub(1), ptr.sub(1), I32(0xFFFF)),
                  isa.sop("s_mov_b32", rs.sub(2), I32(0x7FFFFFFF)), isa.sop("s_mov_b32", rs.sub(3), I32(0x00020000))]
        # ---- scalars ----
        krsb, lkpb = ST[10], ST[11]
        o += [isa.sop("s_lshl_b32", krsb, S_KRS.sub(0), I32(1)),                    # K row stride in bytes
              isa.sop("s_lshl_b32", lkpb, S_LKP, I32(1)),                          # V^T row stride in bytes
              isa.sop("s_lshl_b32", S_KSTEP, krsb, I32(6)), isa.sop("s_mov_b32", S_VSTEP, I32(128)),
              isa.sop("s_lshr_b32", S_NT, S_LKP, I32(6)), isa.sop("s_sub_u32", ST[12], S_NT, I32(1)),
              isa.sop("s_mul_i32", S_KMAX, ST[12], S_KSTEP), isa.sop("s_lshl_b32", S_VMAX, ST[12], I32(7)),
              isa.sop("s_lshl_b32", S_KLDS, S_WAVE, I32(12)), isa.sop("s_add_u32", S_VLDS, S_KLDS, I32(c.rd * 16384)),
              isa.sop("s_mov_b32", S_SEG, I32(0))]
        if c.mi == 16:
            # ---- lane geometry (mi = 16): ql = lane % 16 (fragment row / query column), g = lane / 16 (16-byte chunk inside a 32-wide k-step) ----
            ql, g = VT1, VT2
            o += [isa.vop("v_and_b32", ql, I32(15), LANE), isa.vop("v_lshrrev_b32", g, I32(4), LANE)]
            t = TMP16
            # K fragment addresses.  A-row r = ql of key block kb reads key  32 (kb >> 1) + 8 (kb & 1) + 16 (r >> 3) + (r & 7)  of the tile:
            # with this row order a lane's score registers of blocks 2 ks, 2 ks + 1 are exactly the 8 keys its P.V k-slots hold in the V^T
            # image of scail_transpose_v (key bits 2 <-> 3 swapped inside 16-key groups) -- no cross-lane movement between the two GEMMs.
            # LDS row = key (256 B), 16-byte chunk (4 ks + g) ^ f(key) with f(key) = key bits 0-2 | key bit 4 << 3 -- for the keys of a
            # fragment f = r, which makes the 16 lanes of every ds_read_b128 lane group hit 16 different 4-bank sets (with the 32x32
            # kernel's f = key & 15 half of the K fragment reads were 2-way conflicts here);  parity = kb & 1, kb >> 1 -> immediate offset
            o += [isa.vop("v_and_b32", t[0], I32(7), ql), isa.vop("v_lshrrev_b32", t[1], I32(3), ql), isa.vop("v_lshl_add_u32", t[1], t[1], I32(4), t[0])]   # 16 (r >> 3) + (r & 7)
            for par in range(2):
                o += [isa.vop("v_add_u32", t[2], I32(8 * par), t[1]),                    # key row inside the 32-key half
                      isa.vop("v_lshlrev_b32", t[2], I32(8), t[2])]
                for ks in range(4):
                    o += [isa.vop("v_or_b32", t[4], I32(4 * ks), g), isa.vop("v_xor_b32", t[4], t[4], ql),
                          isa.vop("v_lshl_add_u32", KADDR16[ks][par], t[4], I32(4), t[2])]
            # V^T fragment addresses: row 16 db + ql (128 B), chunk (4 ks + g) ^ ((ql >> 1) & 7), in the V ring; db goes into the offset
            o += [isa.vop("v_lshrrev_b32", t[0], I32(1), ql), isa.vop("v_and_b32", t[0], I32(7), t[0]),
                  isa.vop("v_lshlrev_b32", t[1], I32(7), ql), isa.vop("v_add_u32", t[1], I32(c.rd * 16384), t[1])]
            for ks in range(2):
                o += [isa.vop("v_or_b32", t[2], I32(4 * ks), g), isa.vop("v_xor_b32", t[2], t[2], t[0]),
                      isa.vop("v_lshl_add_u32", VADDR16[ks], t[2], I32(4), t[1])]
        else:
            # ---- lane geometry ----
            ql, g = VT1, VT2
            o += [isa.vop("v_and_b32", ql, I32(31), LANE), isa.vop("v_lshrrev_b32", g, I32(5), LANE)]
            t = TMP
            # K fragment addresses: row ql (256 B), 16-byte chunk (2 ks + g) ^ (ql & 15)
            o += [isa.vop("v_and_b32", t[0], I32(15), ql), isa.vop("v_lshlrev_b32", t[1], I32(8), ql)]
            for ks in range(8):
                o += [isa.vop("v_or_b32", t[2], I32(2 * ks), g), isa.vop("v_xor_b32", t[2], t[2], t[0]),
                      isa.vop("v_lshl_add_u32", KADDR[ks], t[2], I32(4), t[1])]
            # V^T fragment addresses: row ql (128 B), chunk (2 ks + g) ^ ((ql >> 1) & 7), in the V ring
            o += [isa.vop("v_lshrrev_b32", t[0], I32(1), ql), isa.vop("v_and_b32", t[0], I32(7), t[0]),
                  isa.vop("v_lshlrev_b32", t[1], I32(7), ql), isa.vop("v_add_u32", t[1], I32(c.rd * 16384), t[1])]
            for ks in range(4):
                o += [isa.vop("v_or_b32", t[2], I32(2 * ks), g), isa.vop("v_xor_b32", t[2], t[2], t[0]),
                      isa.vop("v_lshl_add_u32", VADDR[ks], t[2], I32(4), t[1])]
        # LDS-DMA source offsets.  K piece i of this wave: tile rows 16 w + 4 i + (lane >> 4), chunk (lane & 15) ^ (row & 15)
        o += [isa.vop("v_lshrrev_b32", t[0], I32(4), LANE), isa.vop("v_and_b32", t[1], I32(15), LANE),
              isa.vop("v_lshlrev_b32", t[3], I32(4), S_WAVE)]       # 16 w
        for i in range(4):
            o += [isa.vop("v_add_u32", t[4], I32(4 * i), t[0])]                     # row & 15
            if c.mi == 16:        # swizzle key f(row) = row bits 0-2 | row bit 4 << 3 (bit 4 of the tile row = wave & 1)
                o += [isa.vop("v_and_b32", t[5], I32(7), t[4]), isa.sop("s_and_b32", ST[0], S_WAVE, I32(1)),
                      isa.vop("v_lshl_or_b32", t[5], ST[0], I32(3), t[5]), isa.vop("v_xor_b32", t[5], t[1], t[5])]
            else:
                o += [isa.vop("v_xor_b32", t[5], t[1], t[4])]
            o += [isa.vop("v_add_u32", t[6], t[4], t[3]),
                  isa.vop("v_mul_lo_u32", t[6], t[6], krsb), isa.vop("v_lshl_add_u32", t[6], t[5], I32(4), t[6]),
                  isa.vop("v_subrev_u32", KDMA[i], I32(1024 * i), t[6])]
        # V^T piece i: rows 32 w + 8 i + (lane >> 3), chunk (lane & 7) ^ ((row >> 1) & 7)
        o += [isa.vop("v_lshrrev_b32", t[0], I32(3), LANE), isa.vop("v_and_b32", t[1], I32(7), LANE),
              isa.vop("v_lshlrev_b32", t[3], I32(5), S_WAVE)]       # 32 w
        for i in range(4):
            o += [isa.vop("v_add_u32", t[4], I32(8 * i), t[0]), isa.vop("v_add_u32", t[6], t[4], t[3]),
                  isa.vop("v_lshrrev_b32", t[5], I32(1), t[6]), isa.vop("v_and_b32", t[5], I32(7), t[5]),
                  isa.vop("v_xor_b32", t[5], t[1], t[5]),
                  isa.vop("v_mul_lo_u32", t[6], t[6], lkpb), isa.vop("v_lshl_add_u32", t[6], t[5], I32(4), t[6]),
                  isa.vop("v_subrev_u32", VDMA[i], I32(1024 * i), t[6])]
        if c.mi == 16:
            # query rows of this lane: row = 256 qb_wg + 64 w + 16 qb + ql ; Q loads straight into the accumulator file:
            # B operand of S^T = K Q^T: lane holds Q[row][32 ks + 8 g .. +7]
            qrsb, orsb, lqm1 = ST[12], ST[13], ST[9]
            o += [isa.sop("s_lshl_b32", qrsb, S_QRS.sub(0), I32(1)), isa.sop("s_lshl_b32", orsb, S_ORS.sub(0), I32(1)),
                  isa.sop("s_sub_u32", lqm1, S_LQ, I32(1))] + self.wave_row_base() + [
                  isa.sop("s_add_u32", ST[8], ST[8], ST[7])]
            qa = [TMP16[4 + qb] for qb in range(4)]
            for qb in range(self.cfg.nq):
                o += [isa.vop("v_add_u32", ROW16[qb], ST[8], ql)]
                if qb:
                    o += [isa.vop("v_add_u32", ROW16[qb], I32(16 * qb), ROW16[qb])]
                o += [isa.vop("v_min_u32", t[0], ROW16[qb], lqm1), isa.vop("v_mul_lo_u32", t[0], t[0], qrsb),
                      isa.vop("v_lshl_add_u32", qa[qb], g, I32(4), t[0]),
                      isa.vop("v_mul_lo_u32", t[3], ROW16[qb], orsb), isa.vop("v_lshl_add_u32", OOFF16[qb], g, I32(3), t[3])]
            for qb in range(self.cfg.nq):
                for ks in range(4):
                    o.append(isa.global_load(4, Qf16(qb, ks), qa[qb], 64 * ks, saddr=S_Q))
            if c.fold:
                if c.qscale:
                    # raw-scale callers (sl2 != 0): Q fragments x sl2 = scale * log2(e), one extra rounding to bf16 (q' = bf16(sl2 * bf16(q)));
                    # callers whose q is in log2 units already pass sl2 == 0 and skip the block (and its wait for the Q loads)
                    o += [isa.sop("s_cmp_eq_u32", None, S_C, I32(0)), isa.branch("s_cbranch_scc1", "L_qdone"), isa.waitcnt(vmcnt=0)]
                    k = 0
                    for qb in range(self.cfg.nq):
                        for ks in range(4):
                            for i in range(4):
                                a = Qf16(qb, ks).sub(i)
                                lo, hi = TMP16[2 * (k % 4)], TMP16[2 * (k % 4) + 1]
                                k += 1
                                o += [isa.vop("v_accvgpr_read_b32", hi, a), isa.vop("v_lshlrev_b32", lo, I32(16), hi),
                                      isa.vop("v_and_b32", hi, I32(0xFFFF0000), hi), isa.vop("v_mul_f32", lo, lo, S_C),
                                      isa.vop("v_mul_f32", hi, hi, S_C), isa.vop("v_cvt_pk_bf16_f32", lo, lo, hi),
                                      isa.vop("v_accvgpr_write_b32", a, lo)]
                    o += [isa.label("L_qdone"), isa.nop(7)]
                # ---- constants of the whole kernel ----
                if c.ragged:
                    # valid keys of a segment's last tile (1..64), relative to the first tile row of this wave's K piece i; lane term of
                    # the key index of a score register (mask_scores16) in VT1 -- ql is not needed past this point
                    o += [isa.sop("s_lshr_b32", ST[0], S_LKP, I32(6)), isa.sop("s_sub_u32", ST[0], ST[0], I32(1)), isa.sop("s_lshl_b32", ST[0], ST[0], I32(6)),
                          isa.sop("s_sub_u32", S_TAIL, S_LK, ST[0]), isa.sop("s_lshl_b32", ST[1], S_WAVE, I32(4)), isa.sop("s_sub_u32", ST[1], S_TAIL, ST[1])]
                    for i in range(4):
                        o.append(isa.sop("s_sub_u32", S_TAILREL[i], ST[1], I32(4 * i)))
                if c.lsum:
                    # row sums on the matrix pipe: A fragment = a row of ones in row 0 (lanes with lane % 16 == 0), zeros elsewhere
                    o += [isa.v_cmp("v_cmp_eq_u32", ql, I32(0)), isa.vop("v_mov_b32", ONES.sub(0), I32(0x3F803F80))]
                    o += [isa.v_cndmask(ONES.sub(0), I32(0), ONES.sub(0))]
                    for i in range(1, 4):
                        o.append(isa.vop("v_mov_b32", ONES.sub(i), ONES.sub(0)))
                if c.ragged:
                    assert c.lsum or c.pksum, "ragged reuses VT0 / VT1, which only the lsum epilogue leaves free"
                    o.append(isa.vop("v_mov_b32", VT1, F32(-3.0e38)))             # the masked score (ql is not needed past this point)
                if c.x2:
                    o += self.x2_set_block(krsb)
                if c.opt:
                    assert c.lsum or c.pksum, "the optimistic pass is verified on the row sums"
                    o += [isa.sop("s_mov_b32", S_MODE, I32(0)), isa.sop("s_mov_b32", S_HEAD, F32(c.head))]
                    # ---- (re)start of a pass over the keys: the epilogue jumps back here with S_MODE = 1 when the optimistic pass
                    #      overflowed; the K / V^T descriptors may have walked over the key segments ----
                    o += [isa.label("L_restart"), isa.nop(15)]
                    for rs, ptr in ((S_KRSRC, S_K), (S_VRSRC, S_VT)):
                        o += [isa.sop("s_mov_b32", rs.sub(0), ptr.sub(0)), isa.sop("s_and_b32", rs.sub(1), ptr.sub(1), I32(0xFFFF))]
                    o += [isa.sop("s_mov_b32", S_SEG, I32(0))]
                # ---- softmax state of a pass ----
                for i in range(128):
                    o.append(isa.vop("v_accvgpr_write_b32", A(i), I32(0)))
                # running maximum starts at 0; the first tile's subroutine call (unconditional, CLAMP = -inf) sets it to the tile's maximum
                for qb in range(self.cfg.nq):
                    for i in range(4):
                        o.append(isa.vop("v_mov_b32", NEGM[qb].sub(i), I32(0)))
                    if not c.lsum:
                        for j in range(2):
                            o.append(isa.vop("v_mov_b32", L16F[qb][j], I32(0)))
                o += [isa.sop("s_mov_b32", S_CLAMP, I32(0xFF800000)), isa.sop("s_mov_b32", S_FIRST, I32(1))]
                if c.lsum:
                    for qb in range(self.cfg.nq):
                        for i in range(4):
                            o.append(isa.vop("v_mov_b32", LACC[qb].sub(i), I32(0)))
            else:
                for i in range(128):
                    o.append(isa.vop("v_accvgpr_write_b32", A(i), I32(0)))
                for qb in range(self.cfg.nq):
                    o += [isa.vop("v_mov_b32", M16[qb], F32(-1e30)), isa.vop("v_mul_f32", MC16[qb], M16[qb], S_C)]
                    for j in range(2):
                        o.append(isa.vop("v_mov_b32", L16[qb][j], I32(0)))
        else:
            # query rows of this lane: row = 256 qb + 64 w + 32 rb + ql ; Q loads straight into the accumulator file
            qrsb, orsb, lqm1 = ST[12], ST[13], ST[9]
            o += [isa.sop("s_lshl_b32", qrsb, S_QRS.sub(0), I32(1)), isa.sop("s_lshl_b32", orsb, S_ORS.sub(0), I32(1)),
                  isa.sop("s_sub_u32", lqm1, S_LQ, I32(1)),
                  isa.sop("s_lshl_b32", ST[8], S_QB, I32(8)), isa.sop("s_lshl_b32", ST[7], S_WAVE, I32(6)),
                  isa.sop("s_add_u32", ST[8], ST[8], ST[7])]
            for rb in range(2):
                o += [isa.vop("v_add_u32", ROW[rb], ST[8], ql)]
                if rb:
                    o += [isa.vop("v_add_u32", ROW[rb], I32(32), ROW[rb])]
                o += [isa.vop("v_min_u32", t[0], ROW[rb], lqm1), isa.vop("v_mul_lo_u32", t[0], t[0], qrsb),
                      isa.vop("v_lshl_add_u32", t[1 + rb], g, I32(4), t[0]),
                      isa.vop("v_mul_lo_u32", t[3], ROW[rb], orsb), isa.vop("v_lshl_add_u32", OOFF[rb], g, I32(3), t[3])]
            for rb in range(2):
                for ks in range(8):
                    o.append(isa.global_load(4, Qf(rb, ks), t[1 + rb], 32 * ks, saddr=S_Q))
            # softmax state
            for i in range(128):
                o.append(isa.vop("v_accvgpr_write_b32", A(i), I32(0)))
            for rb in range(2):
                o += [isa.vop("v_mov_b32", M_[rb], F32(-1e30)), isa.vop("v_mul_f32", MC[rb], M_[rb], S_C)]
                for j in range(4):
                    o.append(isa.vop("v_mov_b32", L_[rb][j], I32(0)))
        return sched.pad_hazards(o)

    def segment_start(self) -> List[Instr]:
        """(Re)start the tile pipeline on the current key segment: first DMAs, scores and row max of tile 0."""
        c = self.cfg
        o: List[Instr] = [isa.label("L_seg_start"), isa.sop("s_mov_b32", S_KOFF, I32(0)), isa.sop("s_mov_b32", S_VOFF, I32(0)),
                          isa.sop("s_mov_b32", S_T, I32(0))]
        for j in range(c.rd):
            o += self.dma_tile("k", j % c.rd, 0, 0, careful=c.ragged)
        for j in range(c.pv):
            o += self.dma_tile("v", j % c.rd, 0, 0)
        m16 = c.mi == 16
        kfr = self.k_frag_reads16 if m16 else self.k_frag_reads
        o += [isa.waitcnt(vmcnt=0), isa.barrier()]
        o += kfr(0, 0, 0)
        o += [isa.waitcnt(lgkmcnt=0), isa.barrier()]
        o += self.dma_tile("k", c.rd % c.rd, 0, 0, careful=c.ragged)   # K(rd) into slot 0, whose fragments are in registers now
        o += self.qk_mfmas16(0) if m16 else self.qk_mfmas(0)
        if c.ragged:
            o += [isa.nop(15)] + self.mask_scores16(0, 0, (0, 0, 0, 0))
        o += kfr(1 % c.rd, 0, 0)
        o += [isa.nop(15)]
        o += self.rowmax16(0, (0, 0, 0, 0), 0) if m16 else self.rowmax(0, 0, 0, 0)
        o += [isa.waitcnt(lgkmcnt=0), isa.waitcnt(vmcnt=c.vm_keep if c.rd > 2 else 0), isa.barrier()]
        if c.fold:
            # the very first tile of the kernel always goes through the subroutine (it establishes the maxima, whatever their sign)
            o += [isa.sop("s_cmp_lg_u32", None, S_FIRST, I32(0)), isa.branch("s_cbranch_scc1", "L_seg_rescale"),
                  isa.branch("s_cbranch_vccz", "L_seg_norescale"), isa.label("L_seg_rescale"), isa.s_call(S_RET, "L_rescale0"),
                  isa.label("L_seg_norescale")]
        else:
            o += [isa.branch("s_cbranch_vccz", "L_seg_norescale"), isa.s_call(S_RET, "L_rescale"), isa.label("L_seg_norescale")]
        return sched.pad_hazards(sched.insert_lgkm_waits(o))

    def loops(self) -> List[Instr]:
        c = self.cfg
        U = c.unroll
        o: List[Instr] = []
        r = ST[0]
        # dispatch: R = T - t remaining tiles (t multiple of U);  R >= U + 2 -> U full iterations in the hot loop
        # (ragged: R >= U + pk + 1, so that neither a K DMA nor a QK^T of the hot loop touches the segment's last tile)
        o += [isa.label("L_dispatch"), isa.sop("s_sub_u32", r, S_NT, S_T), isa.sop("s_cmp_ge_u32", None, r, I32(U + c.pk + 1 if c.ragged else U + 2)),
              isa.branch("s_cbranch_scc0", "L_rem0")]
        if c.opt:       # mode 0: the optimistic hot loop (no row maximum, no branch); mode 1 (after a restart): the lazy-maximum loop
            o += [isa.sop("s_cmp_eq_u32", None, S_MODE, I32(0)), isa.branch("s_cbranch_scc1", "L_hotf")]
        o += self.align_directive() + [isa.label("L_hot")]
        for p in range(U):
            o += self.iter_block(p, tail=False) + self.iter_end(p, "hot")
        o += [isa.sop("s_add_u32", S_T, S_T, I32(U)), isa.branch("s_branch", "L_dispatch")]
        if c.opt:
            o += self.align_directive() + [isa.label("L_hotf")]
            for p in range(U):
                o += self.iter_block(p, tail=False, nomax=True) + self.iter_end(p, "hotf", nomax=True)
            o += [isa.sop("s_add_u32", S_T, S_T, I32(U)), isa.branch("s_branch", "L_dispatch")]
        # remainder chain: positions 0 .. U-1, each either the last tile (-> tail) or one more full iteration
        for p in range(U):
            o += [isa.label(f"L_rem{p}"), isa.sop("s_sub_u32", r, S_NT, S_T), isa.sop("s_cmp_eq_u32", None, r, I32(1)),
                  isa.branch("s_cbranch_scc1", f"L_tail{p}")]
            o += self.iter_block(p, tail=False, careful=True) + self.iter_end(p, "rem")
            o += [isa.sop("s_add_u32", S_T, S_T, I32(1))]
        o += [isa.branch("s_branch", "L_rem0" if c.ragged else "L_tail0")]      # ragged: up to U + pk remainder iterations -> the chain cycles
        for p in range(U):
            o += [isa.label(f"L_tail{p}")] + self.iter_block(p, tail=True)
            o += [isa.nop(15), isa.nop(15), isa.waitcnt(vmcnt=0), isa.waitcnt(lgkmcnt=0), isa.barrier(), isa.branch("s_branch", "L_seg_end")]
        return o

    def segment_end_and_epilogue(self) -> List[Instr]:
        o: List[Instr] = [isa.label("L_seg_end"), isa.sop("s_add_u32", S_SEG, S_SEG, I32(1)), isa.sop("s_cmp_lt_u32", None, S_SEG, S_NSEG),
                          isa.branch("s_cbranch_scc0", "L_epilogue")]
        # next segment: advance both descriptors by the segment strides (bytes, 64-bit)
        st = S(ST[4].idx, 2)
        for rs, ss in ((S_KRSRC, S_KSS), (S_VRSRC, S_VTSS)):
            o += [isa.sop("s_lshl_b64", st, ss, I32(1)), isa.sop("s_add_u32", rs.sub(0), rs.sub(0), st.sub(0)),
                  isa.sop("s_addc_u32", rs.sub(1), rs.sub(1), st.sub(1))]
        o += [isa.branch("s_branch", "L_seg_start")]
        if self.cfg.mi == 16:
            # ---- epilogue (mi = 16): O / l -> bf16; the lane owns query rows 16 qb + ql and columns d = 16 db + 4 g + e.  The S registers
            #      are dead here and serve as temporaries ----
            e = [isa.label("L_epilogue")]
            inv = [V(qb) for qb in range(4)]
            row16, ooff16 = ROW16, OOFF16
            if self.cfg.lsum:
                # the row / offset registers were given to the row-sum accumulators: rebuild them (row = 256 qb_wg + 64 w + 16 qb + lane % 16)
                row16, ooff16 = [V(16 + qb) for qb in range(4)], [V(20 + qb) for qb in range(4)]
                ql, g, t3, orsb = V(24), V(25), V(26), ST[13]
                e += [isa.sop("s_lshl_b32", orsb, S_ORS.sub(0), I32(1))] + self.wave_row_base() + [
                      isa.sop("s_add_u32", ST[8], ST[8], ST[7]),
                      isa.vop("v_and_b32", ql, I32(15), LANE), isa.vop("v_lshrrev_b32", g, I32(4), LANE)]
                for qb in range(self.cfg.nq):
                    e += [isa.vop("v_add_u32", row16[qb], ST[8], ql)]
                    if qb:
                        e += [isa.vop("v_add_u32", row16[qb], I32(16 * qb), row16[qb])]
                    e += [isa.vop("v_mul_lo_u32", t3, row16[qb], orsb), isa.vop("v_lshl_add_u32", ooff16[qb], g, I32(3), t3)]
            opt = self.cfg.opt
            bad = S(ST[0].idx, 2)
            x2 = self.cfg.x2
            stash = [[V(64 + (qb * 8 + db) * 2, 2) for db in range(8)] for qb in range(4)]      # x2, set 1: bf16(O of set 0) as stored (score tile 1 is dead)
            if x2:
                # set 1: request set 0's stored output now (same lane, same addresses; sc0 sc1 = from L2, where the store went), the row-sum
                # reduction and the restart check below run under the latency.  A restarted pass requests it again (harmless).
                e += [isa.sop("s_cmp_eq_u32", None, S_SET, I32(0)), isa.branch("s_cbranch_scc1", "L_x2_noload"), isa.nop(3)]
                for qb in range(self.cfg.nq):
                    e += [isa.v_cmp("v_cmp_lt_u32", row16[qb], S_LQ), Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)]
                    for db in range(8):
                        e.append(isa.global_load(2, stash[qb][db], ooff16[qb], 32 * db, saddr=S_O, sc=True, extra_reads=[EXEC]))
                    e += [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]
                e += [isa.label("L_x2_noload"), isa.nop(3)]
            if opt:
                e += [Instr("s_mov_b64", [bad], [I32(0)], cls=isa.SALU)]
            for qb in range(self.cfg.nq):
                a, b = V(4), V(5)
                lsum = L16F if self.cfg.fold else L16
                first = ([isa.vop("v_mov_b32", a, LACC[qb].sub(0))] if self.cfg.lsum        # lanes 0-15 hold the sums, the other rows of the block are 0
                         else [isa.vop("v_add_f32", a, lsum[qb][0], lsum[qb][1])])
                e += first + [isa.vop("v_mov_b32", b, a),
                      isa.permlane32_swap(a, b), isa.vop("v_add_f32", a, a, b), isa.vop("v_mov_b32", b, a),
                      isa.permlane16_swap(a, b), isa.vop("v_add_f32", a, a, b), isa.vop("v_rcp_f32", inv[qb], a)]
                if opt:     # verification of the optimistic pass: every row sum finite and < 2^60 (NaN compares "not greater" too)
                    e += [isa.v_cmp("v_cmp_ngt_f32", F32(2.0 ** 60), a), isa.nop(3), isa.sop("s_or_b64", bad, bad, VCC)]
            if opt:
                # The four waves share the K / V^T rings and the per-tile barriers, so the decision to run again must be uniform over
                # the workgroup: flags through LDS (the rings are idle here: the tail drained every DMA and ended with a barrier).
                flag, addr, fl4 = V(6), V(7), V(8, 4)
                e += [isa.sop("s_cmp_lg_u32", None, S_MODE, I32(0)), isa.branch("s_cbranch_scc1", "L_store"),
                      isa.sop("s_cmp_lg_u64", None, bad, I32(0)), isa.sop("s_cselect_b32", ST[2], I32(1), I32(0)),
                      isa.sop("s_lshl_b32", ST[3], S_WAVE, I32(2)),
                      isa.vop("v_mov_b32", flag, ST[2]), isa.vop("v_mov_b32", addr, ST[3]),
                      isa.ds_write(4, addr, flag), isa.waitcnt(lgkmcnt=0), isa.barrier(),
                      isa.vop("v_mov_b32", addr, I32(0)), isa.ds_read_b128(fl4, addr), isa.waitcnt(lgkmcnt=0),
                      isa.vop("v_or3_b32", flag, fl4.sub(0), fl4.sub(1), fl4.sub(2)), isa.vop("v_or_b32", flag, flag, fl4.sub(3)),
                      isa.vop("v_readfirstlane_b32", ST[2], flag),
                      isa.barrier(),                     # every wave has read the flags before a restarting workgroup's DMA reuses the ring
                      isa.nop(3), isa.sop("s_cmp_eq_u32", None, ST[2], I32(0)), isa.branch("s_cbranch_scc1", "L_store"),
                      # restart counter (optional kernel argument): wave 0, one lane, one atomic per restarting workgroup; drained
                      # before the pass starts again (the ring's counted vmcnt waits must not see it)
                      isa.sop("s_cmp_lg_u32", None, S_WAVE, I32(0)), isa.branch("s_cbranch_scc1", "L_nocount"),
                      isa.s_load(2, S(ST[4].idx, 2), S_KARG, KARG_RESTARTS), isa.waitcnt(lgkmcnt=0),
                      isa.sop("s_cmp_eq_u64", None, S(ST[4].idx, 2), I32(0)), isa.branch("s_cbranch_scc1", "L_nocount"),
                      isa.vop("v_mov_b32", addr, I32(0)), isa.vop("v_mov_b32", flag, I32(1)),
                      Instr("s_mov_b64", [S_SAVE], [EXEC], cls=isa.SALU), Instr("s_mov_b64", [EXEC], [I32(1)], cls=isa.SALU),
                      isa.global_atomic_add(addr, flag, S(ST[4].idx, 2), extra_reads=[EXEC]), isa.waitcnt(vmcnt=0),
                      Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU),
                      isa.label("L_nocount"), isa.nop(3),
                      isa.sop("s_mov_b32", S_MODE, I32(1)), isa.branch("s_branch", "L_restart"),
                      isa.label("L_store"), isa.nop(7)]

            def store_pass(add_stash: bool) -> List[Instr]:
                r: List[Instr] = []
                for qb in range(self.cfg.nq):
                    r += [isa.v_cmp("v_cmp_lt_u32", row16[qb], S_LQ), Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)]
                    for db in range(8):
                        base = 32 + 6 * (db % 2)
                        f = [V(base + i) for i in range(4)]
                        w = V(base + 4, 2)
                        for i in range(4):
                            r += [isa.vop("v_accvgpr_read_b32", f[i], O16(db, qb).sub(i)), isa.vop("v_mul_f32", f[i], f[i], inv[qb])]
                        if add_stash:       # + bf16(O of set 0): element 2 j of the pair is the low half
                            st = stash[qb][db]
                            lo, hi = V(44), V(45)
                            for j in range(2):
                                r += [isa.vop("v_lshlrev_b32", lo, I32(16), st.sub(j)), isa.vop("v_and_b32", hi, I32(0xFFFF0000), st.sub(j)),
                                      isa.vop("v_add_f32", f[2 * j], f[2 * j], lo), isa.vop("v_add_f32", f[2 * j + 1], f[2 * j + 1], hi)]
                        r += [isa.vop("v_cvt_pk_bf16_f32", w.sub(0), f[0], f[1]), isa.vop("v_cvt_pk_bf16_f32", w.sub(1), f[2], f[3]),
                              isa.global_store(2, ooff16[qb], w, 32 * db, saddr=S_O, extra_reads=[EXEC])]
                    r += [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]
                return r

            if not x2:
                e += store_pass(False)
                e += [isa.waitcnt(vmcnt=0), Instr("s_endpgm", cls=isa.BRANCH)]
                return o + sched.pad_hazards(e)
            # x2.  Set 0: store bf16(O / l) at its final place and run the pass of set 1.  Set 1: add what set 0 stored, store, next item.
            e += [isa.sop("s_cmp_eq_u32", None, S_SET, I32(0)), isa.branch("s_cbranch_scc0", "L_x2_final")]
            e += store_pass(False)
            e += [isa.sop("s_mov_b32", S_SET, I32(1)), isa.branch("s_branch", "L_set")]
            e += [isa.label("L_x2_final"), isa.nop(7), isa.waitcnt(vmcnt=0)]
            e += store_pass(True)
            e += [isa.sop("s_add_u32", S_ITEM, S_ITEM, S_NWG), isa.sop("s_cmp_lt_u32", None, S_ITEM, S_NIT),
                  isa.branch("s_cbranch_scc1", "L_item"), isa.waitcnt(vmcnt=0), Instr("s_endpgm", cls=isa.BRANCH)]
            i0 = next(k for k, x in enumerate(e) if x.label == "L_x2_final")
            return o + sched.pad_hazards(e[:i0]) + sched.pad_hazards(e[i0:])
        # ---- epilogue: O / l -> bf16, rows of this lane: d = 32 db + 8 rr + 4 g + e ----
        e: List[Instr] = [isa.label("L_epilogue")]
        inv = [TMP[0], TMP[1]]
        for rb in range(2):
            a, b = TMP[2], TMP[3]
            e += [isa.vop("v_add_f32", L_[rb][0], L_[rb][0], L_[rb][1]), isa.vop("v_add_f32", L_[rb][2], L_[rb][2], L_[rb][3]),
                  isa.vop("v_add_f32", a, L_[rb][0], L_[rb][2]), isa.vop("v_mov_b32", b, a),
                  isa.permlane32_swap(a, b), isa.vop("v_add_f32", a, a, b), isa.vop("v_rcp_f32", inv[rb], a)]
        for rb in range(2):
            e += [isa.v_cmp("v_cmp_lt_u32", ROW[rb], S_LQ), Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)]
            k = 0
            for db in range(4):
                for rr in range(4):
                    base = 4 + 6 * (k % 2)
                    k += 1
                    f = [TMP[base + i] for i in range(4)]
                    w = V(TMP[base + 4].idx, 2)
                    assert w.idx % 2 == 0
                    for i in range(4):
                        e += [isa.vop("v_accvgpr_read_b32", f[i], O(db, rb).sub(4 * rr + i)), isa.vop("v_mul_f32", f[i], f[i], inv[rb])]
                    e += [isa.vop("v_cvt_pk_bf16_f32", w.sub(0), f[0], f[1]), isa.vop("v_cvt_pk_bf16_f32", w.sub(1), f[2], f[3]),
                          isa.global_store(2, OOFF[rb], w, 64 * db + 16 * rr, saddr=S_O, extra_reads=[EXEC])]
            e += [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]
        e += [isa.waitcnt(vmcnt=0), Instr("s_endpgm", cls=isa.BRANCH)]
        return o + sched.pad_hazards(e)

    # =============================================================================================
    def program(self) -> List[Instr]:
        prog = self.prologue() + self.segment_start() + self.loops() + self.segment_end_and_epilogue() + self.rescale_sub()
        pre = f"L_{self.cfg.name}"                     # labels are per kernel (several variants may share one .s file)
        for i in prog:
            if i.label and i.label.startswith("L_"):
                new = pre + i.label[1:]
                if getattr(i, "text", None):
                    i.text = i.text.replace(i.label, new)
                i.label = new
        return prog


HEAD = """// GENERATED by scail_amd/asmgen/attn4.py -- do not edit; regenerate with `python -m scail_amd.asmgen.attn4`.
// Hand-scheduled 4-wave flash attention for gfx950; see the generator's docstring for the register map and the pipeline.
\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
"""


def kernel_text(c: Cfg) -> str:
    body = isa.render(Gen(c).program())
    return f"""// ---- kernel {c.name}: ring depth {c.rd}, <= {c.cap} fillers per MFMA gap, lookahead {c.lookahead} ----
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
\t\t.amdhsa_group_segment_fixed_size {c.lds_bytes}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {X2_KERNARG_SIZE if c.x2 else KERNARG_SIZE}
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 1
\t\t.amdhsa_system_sgpr_workgroup_id_z 1
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr 512
\t\t.amdhsa_next_free_sgpr 96
\t\t.amdhsa_accum_offset 256
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
    ks = "".join(f"""  - .agpr_count:     256
    .args:
      - .offset:         0
        .size:           {X2_KERNARG_SIZE if c.x2 else KERNARG_SIZE}
        .value_kind:     by_value
    .group_segment_fixed_size: {c.lds_bytes}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {X2_KERNARG_SIZE if c.x2 else KERNARG_SIZE}
    .max_flat_workgroup_size: 256
    .name:           {c.name}
    .private_segment_fixed_size: 0
    .sgpr_count:     102
    .sgpr_spill_count: 0
    .symbol:         {c.name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     512
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


DEFAULT = Cfg(rd=4, cap=5, name="scail_attn4")


# the shipped kernel: M16F (16x16x32 MFMAs, scores in log2 units -- q pre-multiplied by scale * log2 e, or multiplied in the prologue for
# raw-scale callers --, running maximum folded into the accumulator init, optimistic hot loop).  DEFAULT (32x32x16 MFMAs, scale per
# score; round 2's raw-scale kernel) lives on in the measurement build and in the emulator tests
M16F = Cfg(name="scail_attn4_m16f", mi=16, fold=True, lsum=True, ragged=True, cap=1, sm_end=44.0, lookahead=2.0, qscale=True, opt=True, pv_qb=True)
# the same kernel with 3 query blocks per wave = 192-row workgroups: the launch shape of a sequence-parallel rank (Ulysses at 8 ranks: 5 heads x
# 191 tiles of 256 rows = 3.73 rounds over 256 CUs -> 4; 255 tiles of 192 rows = 4.98 rounds of a shorter tile)
M16F_Q3 = Cfg(name="scail_attn4_m16f_q3", mi=16, fold=True, lsum=True, ragged=True, cap=1, sm_end=44.0, lookahead=2.0, qscale=True, opt=True, pv_qb=True, nq=3)
# cross attention over the text and the CLIP key set (Cfg.x2): the M16F pipeline with persistent workgroups
X2 = Cfg(name="scail_attn4_x2", mi=16, fold=True, lsum=True, ragged=True, cap=1, sm_end=44.0, lookahead=2.0, qscale=True, opt=True, pv_qb=True, x2=True)
SHIPPED = [M16F, M16F_Q3, X2]


def variant_cfgs():
    """A/B variants for GPU tuning runs (ablation build only); the kernel name encodes the knobs, tools/attn4_tune.py lists them."""
    out = [DEFAULT]
    for rd, cap in ((4, 4), (4, 6), (2, 5)):
        out.append(Cfg(rd=rd, cap=cap, name=f"scail_attn4_r{rd}c{cap}"))
    # placement of the 8 LDS-DMA pieces inside the 64-gap body
    out.append(Cfg(name="scail_attn4_dmalate", dma_k_at=49.0, dma_v_at=56.0, dma_step=1.7))
    out.append(Cfg(name="scail_attn4_dmamid", dma_k_at=26.0, dma_v_at=42.0, dma_step=4.0))
    out.append(Cfg(name="scail_attn4_dmaspread", dma_k_at=2.0, dma_v_at=34.0, dma_step=8.0))
    out.append(Cfg(name="scail_attn4_sm48", sm_end=48.0))
    out.append(Cfg(name="scail_attn4_sm60", sm_end=60.0))
    # dependency distance of the softmax VALU stream (one wave per SIMD: nothing else hides a dependent instruction's latency)
    for g, m in ((8, 1), (16, 1), (16, 2), (16, 4), (8, 4)):
        out.append(Cfg(name=f"scail_attn4_g{g}m{m}", sm_group=g, max_chains=m))
    out.append(Cfg(name="scail_attn4_g16m4c6", sm_group=16, max_chains=4, cap=6))
    out.append(Cfg(name="scail_attn4_g16m4c4", sm_group=16, max_chains=4, cap=4))
    # timing ablations (WRONG RESULTS): what each instruction class costs beside the 64 MFMAs of a tile
    for abl in ("dma", "lds", "valu", "bar", "max", "dma,lds", "dma,lds,valu", "dma,lds,valu,bar", "fma"):
        out.append(Cfg(name="scail_attn4_abl_" + abl.replace(",", "_"), abl=abl))
    # 16 x 16 x 32 MFMAs (128 per tile, 16-cycle gaps)
    for cap in (2, 3):
        out.append(Cfg(name=f"scail_attn4_m16c{cap}", cap=cap, mi=16))
    out.append(Cfg(name="scail_attn4_m16c2la2", cap=2, mi=16, lookahead=2.0))
    for abl in ("dma", "lds", "valu", "dma,lds,valu"):
        out.append(Cfg(name="scail_attn4_m16_abl_" + abl.replace(",", "_"), abl=abl, mi=16, cap=2))
    F = dict(mi=16, fold=True, cap=2, lookahead=2.0)
    out.append(Cfg(name="scail_attn4_m16g", lsum=True, **F))
    out.append(Cfg(name="scail_attn4_m16g_sm44", lsum=True, sm_end=44.0, **F))
    out.append(Cfg(name="scail_attn4_m16g_sm36", lsum=True, sm_end=36.0, **F))
    out.append(Cfg(name="scail_attn4_m16g_c1", lsum=True, **{**F, "cap": 1}))
    G1 = dict(mi=16, fold=True, lsum=True, cap=1)
    out.append(Cfg(name="scail_attn4_m16g_c1sm44", sm_end=44.0, lookahead=2.0, **G1))
    out.append(Cfg(name="scail_attn4_m16g_c1sm60", sm_end=60.0, lookahead=2.0, **G1))
    out.append(Cfg(name="scail_attn4_m16g_c1la1", lookahead=1.0, **G1))
    out.append(Cfg(name="scail_attn4_m16g_c1la4", lookahead=4.0, **G1))
    out.append(Cfg(name="scail_attn4_m16g_c1dmamid", dma_k_at=26.0, dma_v_at=42.0, dma_step=4.0, lookahead=2.0, **G1))
    out.append(Cfg(name="scail_attn4_m16g_c1dmaspread", dma_k_at=2.0, dma_v_at=34.0, dma_step=8.0, lookahead=2.0, **G1))
    out.append(Cfg(name="scail_attn4_m16f_c1", mi=16, fold=True, cap=1, lookahead=2.0))
    out.append(Cfg(name="scail_attn4_m16f_nolsum", mi=16, fold=True, cap=2, lookahead=2.0))      # round-2 first fold version (row sums on the VALU)
    out.append(Cfg(name="scail_attn4_m16f_c3", mi=16, fold=True, cap=3))
    out.append(Cfg(name="scail_attn4_m16f_c3la2", mi=16, fold=True, cap=3, lookahead=2.0))
    out.append(Cfg(name="scail_attn4_m16f_la1", mi=16, fold=True, cap=2, lookahead=1.0))
    out.append(Cfg(name="scail_attn4_m16f_la4", mi=16, fold=True, cap=2, lookahead=4.0))
    for sm in (40.0, 44.0, 48.0, 60.0):
        out.append(Cfg(name=f"scail_attn4_m16f_sm{int(sm)}", sm_end=sm, **F))
    out.append(Cfg(name="scail_attn4_m16f_dmaspread", dma_k_at=2.0, dma_v_at=34.0, dma_step=8.0, **F))
    out.append(Cfg(name="scail_attn4_m16f_dmamid", dma_k_at=26.0, dma_v_at=42.0, dma_step=4.0, **F))
    for abl in ("dma", "lds", "valu", "max", "bar", "dma,lds,valu"):
        out.append(Cfg(name="scail_attn4_m16f_abl_" + abl.replace(",", "_"), abl=abl, **F))
    # round 3: the shipped kernel's knobs (optimistic hot loop) -- A/B against the round-2 loop, code placement, softmax stream extent
    P = dict(mi=16, fold=True, lsum=True, ragged=True, cap=1, sm_end=44.0, lookahead=2.0, qscale=True)
    out.append(Cfg(name="scail_attn4_m16f_noopt", **P))                                  # round 2's hot loop (+ the prologue scaling)
    out.append(Cfg(name="scail_attn4_m16f_opt_db", opt=True, **P))                       # optimistic loop, P.V in (db, qb) order
    out.append(Cfg(name="scail_attn4_m16f_opt_a6", opt=True, align=64, **P))
    out.append(Cfg(name="scail_attn4_m16f_opt_a8", opt=True, align=256, **P))
    for sm in (36.0, 48.0, 50.0):
        out.append(Cfg(name=f"scail_attn4_m16f_opt_sm{int(sm)}", opt=True, **{**P, "sm_end": sm}))
    for sm in (44.0, 52.0, 60.0):
        out.append(Cfg(name=f"scail_attn4_m16f_opt_qb_sm{int(sm)}", opt=True, pv_qb=True, **{**P, "sm_end": sm}))
    out.append(Cfg(name="scail_attn4_m16f_opt_c2", opt=True, **{**P, "cap": 2}))
    out.append(Cfg(name="scail_attn4_m16f_opt_la1", opt=True, **{**P, "lookahead": 1.0}))
    out.append(Cfg(name="scail_attn4_m16f_opt_la4", opt=True, **{**P, "lookahead": 4.0}))
    out.append(Cfg(name="scail_attn4_m16f_opt_dmamid", opt=True, dma_k_at=26.0, dma_v_at=42.0, dma_step=4.0, **P))
    out.append(Cfg(name="scail_attn4_m16f_opt_dmaspread", opt=True, dma_k_at=2.0, dma_v_at=34.0, dma_step=8.0, **P))
    for abl in ("dma", "lds", "valu", "bar", "dma,lds,valu"):
        out.append(Cfg(name="scail_attn4_m16f_opt_abl_" + abl.replace(",", "_"), opt=True, abl=abl, **P))
    # evenly spread streams: P.V in query-block-major order, exp / pack over the whole tile, fragment reads and DMA pieces at a fixed pitch
    U = dict(pv_qb=True, v_at=2.0, v_step=3.0, k_at=20.0, k_step=3.0, dma_k_at=1.0, dma_v_at=33.0, dma_step=8.0)
    PU = {**P, **U}
    out.append(Cfg(name="scail_attn4_m16f_u_c1", opt=True, **{**PU, "sm_end": 58.0}))
    out.append(Cfg(name="scail_attn4_m16f_u_c1le2", opt=True, late_extra=2.0, **{**PU, "sm_end": 58.0}))
    out.append(Cfg(name="scail_attn4_m16f_u_c1le4", opt=True, late_extra=4.0, **{**PU, "sm_end": 58.0}))
    out.append(Cfg(name="scail_attn4_m16f_u_c1le2sm50", opt=True, late_extra=2.0, **{**PU, "sm_end": 50.0}))
    out.append(Cfg(name="scail_attn4_m16f_u_c2", opt=True, **{**PU, "sm_end": 58.0, "cap": 2, "lookahead": 0.6}))
    out.append(Cfg(name="scail_attn4_m16f_u_c2la2", opt=True, **{**PU, "sm_end": 58.0, "cap": 2}))
    out.append(Cfg(name="scail_attn4_m16f_le2", opt=True, late_extra=2.0, **P))             # shipped targets + backlog relief
    out.append(Cfg(name="scail_attn4_m16f_qb_le2", opt=True, late_extra=2.0, pv_qb=True, **P))
    # second sweep around the shipped schedule (query-block-major P.V, exp / pack stream over gaps 0..88)
    Q = {**P, "opt": True, "pv_qb": True}
    for sm in (40.0, 48.0):
        out.append(Cfg(name=f"scail_attn4_m16f_q_sm{int(sm)}", **{**Q, "sm_end": sm}))
    out.append(Cfg(name="scail_attn4_m16f_q_le4", late_extra=4.0, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_le8", late_extra=8.0, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_v20", v_step=2.0, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_v30", v_step=3.0, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_k16", k_at=16.0, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_k24s25", k_at=24.0, k_step=2.5, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_k20s30", k_at=20.0, k_step=3.0, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_dmamid", dma_k_at=26.0, dma_v_at=42.0, dma_step=4.0, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_dmaspread", dma_k_at=1.0, dma_v_at=33.0, dma_step=8.0, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_dmalate", dma_k_at=49.0, dma_v_at=56.0, dma_step=1.7, **Q))
    out.append(Cfg(name="scail_attn4_m16f_q_la1", **{**Q, "lookahead": 1.0}))
    out.append(Cfg(name="scail_attn4_m16f_q_la4", **{**Q, "lookahead": 4.0}))
    out.append(Cfg(name="scail_attn4_m16f_q_u_le4", late_extra=4.0, v_step=3.0, k_at=20.0, k_step=3.0, dma_k_at=1.0, dma_v_at=33.0, dma_step=8.0, **Q))
    # round 5: row sums by v_pk_add_f32 instead of the ones-row MFMAs (128 MFMAs per tile), the shipped kernel's other knobs
    PK = {**P, "opt": True, "pv_qb": True, "lsum": False, "pksum": True}
    out.append(Cfg(name="scail_attn4_m16f_pk", **PK))
    out.append(Cfg(name="scail_attn4_m16f_pk_c2", **{**PK, "cap": 2}))
    out.append(Cfg(name="scail_attn4_m16f_pk_sm52", **{**PK, "sm_end": 52.0}))
    out.append(Cfg(name="scail_attn4_m16f_pk_c2sm52", **{**PK, "cap": 2, "sm_end": 52.0}))
    out.append(Cfg(name="scail_attn4_m16f_pk_le2", late_extra=2.0, **PK))
    out.append(Cfg(name="scail_attn4_m16_abl_fma", abl="fma", mi=16, cap=2, lookahead=2.0))
    out.append(Cfg(name="scail_attn4_m16_abl_fma_c3", abl="fma", mi=16, cap=3))
    out.append(Cfg(name="scail_attn4_m16_abl_fma_sm44", abl="fma", mi=16, cap=2, sm_end=44.0, lookahead=2.0))
    out.append(Cfg(name="scail_attn4_m16_abl_fma_max", abl="fma,max", mi=16, cap=2, lookahead=2.0))
    out.append(Cfg(name="scail_attn4_abl_fma_c4", abl="fma", cap=4))
    out.append(Cfg(name="scail_attn4_abl_fma_sm44", abl="fma", sm_end=44.0))
    return out


def main():
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "csrc", "attn4.s")
    if "--variants" in sys.argv:
        dst = sys.argv[sys.argv.index("--variants") + 1]
        open(dst, "w").write(assembly(SHIPPED + variant_cfgs()))
        print(dst)
        return
    text = assembly(SHIPPED)
    if "--check" in sys.argv:
        sys.exit(0 if open(out).read() == text else 1)
    if not os.path.exists(out) or open(out).read() != text:
        open(out, "w").write(text)
    print(out, len(text.splitlines()), "lines")


if __name__ == "__main__":
    main()
