"""gemm4 -- hand-scheduled bf16 GEMM  y = epi(x W^T + b)  for the big per-token projections of the SCAIL DiT on gfx950
(generator of csrc/gemm4.s).  Replaces F.linear in ColumnParallelLinear / RowParallelLinear (sat/mpu/layers.py:237, 435) and the
gate / residual / GELU ops around it (dit_video_crossattn_sc_xc.py:1036, 1042, 1050; sat/transformer_defaults.py:163-176); same C
entry point as the kernels of csrc/gemm.hip (scail_gemm_bf16), which selects this kernel for M >= 2048, N % 256 == 0, K % 64 == 0.

Shape (the structure of the vendor's own MI355X GEMM: 256 x 256 x 64 macro tile, 4 waves, one per SIMD):
  * workgroup = one 256 (m) x 256 (n) tile of y, 4 waves as 2 (m) x 2 (n), wave tile 128 x 128 = 4 x 4 blocks of 32 x 32;
    the 256 fp32 accumulators per lane fill the whole accumulator file a[0:255]; the arch VGPRs hold only fragments and addresses
  * MFMA issued "transposed" like gemm.hip: A = W fragment (rows n), B = x fragment (columns m) -> a lane's accumulator registers
    run along n: 4 consecutive n per register quad -> 8-byte bf16 stores, float4 bias / gate loads
  * x and W k-tiles (256 rows x 64 k = 32 KB each) are staged THROUGH REGISTERS: with one wave per SIMD the arch VGPRs are free
    (the accumulators live in the AGPR file), and a `buffer_load_dwordx4` + `ds_write_b128` pair costs the wave ~20 issue cycles
    per KiB where an LDS-DMA piece (`buffer_load ... lds`) blocks it for 60-180 (measured here: the LDS-DMA version of this kernel
    ran 1163 TFLOP/s, 1548 with the DMA removed).  Two staging register sets: the loads of tile t+3 are issued at the top of
    iteration t, the set holding tile t+2 is written to LDS late in iteration t -> ~1.75 iterations of flight per load.
    LDS rows are whole 128-byte lines, XOR-swizzled on the WRITE address (chunk ^ ((row >> 1) & 7)), two 64 KB slots; fragments
    are read per 16-wide k-step into a double-buffered register set; per k-tile ONE s_barrier, placed before the LAST k-step: at
    that point every wave has read its last fragments of the tile, so the slot is refilled (tile t+2) while k-step 3 computes and
    the first fragments of tile t+1 are read -- no exposed LDS latency at tile boundaries
  * tile order = a host-built table (XCD-aware + grouped: the 32 tiles resident on an XCD share 4 + 8 operand panels in its L2),
    read with one s_load per workgroup: no integer division in the kernel
  * epilogues as separate kernels: bias (0), bias + GELU-tanh (1), residual + gate * (. + bias) (3), residual + (. + bias) (4);
    bias may be NULL

Limits (the C entry keeps gemm.hip's kernels otherwise): N % 256 == 0, K % 64 == 0, K >= 128, M * ld < 2^31 elements.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import List

from . import isa, sched
from .isa import A, S, V, I32, F32, Neg, VCC, EXEC, M0, Instr

KERNARG_SIZE = 112
KERNARG_FMT = "<7Q4q4i2i"      # x w bias y resid gate table | lda ldc ldr gs | M N K rows_per_batch | grid workgroups, table entries


def pack_args(x, w, bias, y, resid, gate, table, lda, ldc, ldr, gs, M, N, K, rpb, grid=0, entries=0) -> bytes:
    b = struct.pack(KERNARG_FMT, x, w, bias, y, resid, gate, table, lda, ldc, ldr, gs, M, N, K, rpb, grid, entries)
    assert len(b) == KERNARG_SIZE
    return b


def tile_table(M: int, N: int, group_m: int = 4):
    """Tile order (list of (m_tile << 16 | n_tile... stored as m_tile | n_tile << 16): workgroup id b runs on XCD b % 8; each XCD
    gets a contiguous range of the GROUPED order (group_m m-tiles x all n-tiles column by column), so the 32 tiles in flight on an
    XCD cover ~4 m-panels x 8 n-panels."""
    tm, tn = (M + 255) // 256, N // 256
    order = []
    for g0 in range(0, tm, group_m):
        rows = range(g0, min(g0 + group_m, tm))
        for n in range(tn):
            for m in rows:
                order.append(m | (n << 16))
    T = len(order)
    per = (T + 7) // 8
    table = []
    for b in range(per * 8):
        xcd, j = b & 7, b >> 3
        lin = xcd * per + j
        table.append(order[lin] if lin < T else 0xFFFFFFFF)
    return table


@dataclass
class Cfg:
    epi: int = 0           # 0 bias, 1 bias + GELU-tanh, 3 resid + gate * (acc + bias), 4 resid + (acc + bias)
    cap: int = 3           # fillers per MFMA gap
    lookahead: float = 1.0 # scheduler: how far before its target gap a filler may be placed
    stage: str = "reg"     # "reg": global -> VGPR -> ds_write (default); "dma": LDS-DMA (buffer_load ... lds), kept for the A/B;
                           # "dma2": LDS-DMA, whole-tile fragment registers read half a tile ahead, two barriers (body_dma2)
    b1_at: float = 9.5     # dma2: gap of the barrier that releases the slot (after the reads of k-steps 2, 3)
    b2_at: float = 31.5    # dma2: gap of the barrier that publishes tile t+1
    rd2_step: float = 0.5  # dma2: spacing of the 16 fragment reads of a half tile
    wpacked: bool = False  # EXPERIMENT (measured: no gain, 1425 vs 1435 TFLOP/s): W arrives TILE-MAJOR (pack_w below): block (n tile,
                           # k tile) = the 32 KB LDS image of that tile, so each LDS-DMA piece of W reads 1 KB of contiguous memory.  The
                           # "wpack" / "xpack" TIMING ablations that suggested +6 % re-read the same 32 KB every k-tile: they measure L2
                           # hits, not contiguity.  Kept as an emulator-tested variant of the measurement build only
    nt_store: bool = False # epilogue stores with the non-temporal hint (y is written once and read by a later kernel)
    stg: bool = False      # (mi = 16) epilogue through LDS: a row block (16 rows x 128 columns of the wave) is written in accumulator layout and read
                           # back in memory layout, so a global store / residual load instruction moves 4 rows x 256 contiguous bytes (whole cache
                           # lines) instead of 16 rows x 32 bytes; staging: 4 x 4352 bytes behind the two k-tile slots
    mi: int = 32           # MFMA shape: 32 = v_mfma_f32_32x32x16_bf16 (64 per tile), 16 = v_mfma_f32_16x16x32_bf16 (128 per tile; dma2 only)
    dma_from: float = 48.0  # first gap of the 16 LDS-DMA pieces / LDS writes of tile t+2 (after the barrier at 47.5)
    dma_step: float = 1.0
    ld_from: float = 1.0   # first gap of the 16 global loads of tile t+3 (register staging)
    ld_step: float = 2.0
    rd_step: float = 1.5   # spacing (gaps) of the 8 fragment reads of a k-step
    stagger: float = 0.0   # > 0: one loop copy per wave, wave w issues its global loads `stagger * w` gaps later
    ne: int = 4            # stage "spread": pieces written right after the barrier (the other 16 - ne spread over k-steps 0-2)
    late_from: float = 1.0
    late_step: float = 3.7
    persist: bool = False  # (dma2, mi = 16) PERSISTENT workgroups: workgroup b walks table entries b, b + grid, b + 2 grid ... and issues the
                           # first two DMA tiles of its NEXT tile before the epilogue of the current one (see program_persistent)
    name: str = "scail_gemm4_e0"
    abl: str = ""


def ACC(nb, mb): return A((nb * 4 + mb) * 16, 16)
def FW(buf, nb): return V(buf * 32 + nb * 4, 4)            # W fragments (MFMA A operand)
def FX(buf, mb): return V(buf * 32 + 16 + mb * 4, 4)       # x fragments (MFMA B operand)
# stage "dma2": the fragments of a WHOLE k-tile stay in registers (4 k-steps x 32): sets 0, 1 = v0..63, sets 2, 3 = v96..159
# (the staging registers of the register path, unused by the LDS-DMA paths)
def FW4(ks, nb): return V((ks * 32 if ks < 2 else 96 + (ks - 2) * 32) + nb * 4, 4)
def FX4(ks, mb): return V((ks * 32 if ks < 2 else 96 + (ks - 2) * 32) + 16 + mb * 4, 4)


# mi = 16: 8 x 8 blocks of 16 x 16 per wave (4 accumulators each), fragments of a k32-step = 8 + 8 quads: set 0 = v0..63, set 1 = v96..159
def ACC16(nb, mb): return A((nb * 8 + mb) * 4, 4)
def FW16(s, nb): return V((0 if s == 0 else 96) + nb * 4, 4)
def FX16(s, mb): return V((0 if s == 0 else 96) + 32 + mb * 4, 4)


XADDR = [[V(64 + s * 4 + ks) for ks in range(4)] for s in range(2)]
WADDR = [[V(72 + s * 4 + ks) for ks in range(4)] for s in range(2)]
XDMA = [V(80 + i) for i in range(8)]         # per-piece source offsets (bytes inside the operand panel)
WDMA = [V(88 + i) for i in range(8)]
def STG(q, op, i): return V(96 + q * 64 + op * 32 + i * 4, 4)     # staging sets: q in {0, 1}, operand 0 = x / 1 = W, piece i
WRADDR = [[V(224 + s_ * 2 + par) for par in range(2)] for s_ in range(2)]     # LDS write address [slot][piece parity]
LANE = V(228)
T_ = [V(229 + i) for i in range(27)]         # v229..255 temporaries (prologue / epilogue); T_[1], T_[2] = lane geometry, kept

S_KARG = S(0, 2)
S_WG = S(2)
S_X, S_W, S_BIAS, S_Y = S(8, 2), S(10, 2), S(12, 2), S(14, 2)
S_RES, S_GATE = S(16, 2), S(18, 2)
S_TAB = S(20, 2)
S_LDA, S_LDC, S_LDR, S_GS = S(24, 2), S(26, 2), S(28, 2), S(30, 2)
S_M, S_N, S_K, S_RPB = S(32), S(33), S(34), S(35)
S_XRSRC, S_WRSRC = S(36, 4), S(40, 4)
S_KOFF, S_KMAX, S_T, S_KT = S(44), S(45), S(46), S(47)
S_WAVE, S_WM, S_WN = S(48), S(49), S(50)
S_M0T, S_N0T = S(51), S(52)                  # tile origin (rows / columns)
S_XLDS, S_WLDS = S(53), S(54)                # LDS byte offset of this wave's DMA region inside a slot
ST = [S(56 + i) for i in range(16)]          # s56..s71 temporaries
S_SAVE = S(72, 2)
S_WOFF, S_WMAX = S(74), S(75)                # wpacked: byte offset of the current k-tile block inside the packed W panel, its last value
# persistent kernels: working copies of the operand pointers (the originals stay in s8..s19), the entry walked, next tile
S_XC, S_WC, S_YC, S_BC, S_RC, S_GC = S(76, 2), S(78, 2), S(80, 2), S(82, 2), S(84, 2), S(86, 2)
S_E, S_NVALID, S_G, S_ENT, S_M0N, S_N0N = S(88), S(89), S(90), S(91), S(92), S(93)
S_TABC = S(94, 2)


class Gen:
    def __init__(self, cfg: Cfg):
        self.cfg = cfg

    # ---------------------------------------------------------------------------------------------
    def fw(self, ks, nb): return FW4(ks, nb) if self.cfg.stage == "dma2" else FW(ks & 1, nb)
    def fx(self, ks, mb): return FX4(ks, mb) if self.cfg.stage == "dma2" else FX(ks & 1, mb)

    def mfmas16(self, s: int) -> List[Instr]:
        return [isa.mfma16(ACC16(nb, mb), FW16(s, nb), FX16(s, mb), ACC16(nb, mb), tag=f"k{s}") for nb in range(8) for mb in range(8)]

    def frag_reads16(self, slot: int, s: int, t0: float, step: float) -> List[Instr]:
        out = []
        for i in range(8):
            out.append(isa.ds_read_b128(FW16(s, i), WADDR[slot][s], 2048 * i, target_gap=t0 + step * (2 * i)))
            out.append(isa.ds_read_b128(FX16(s, i), XADDR[slot][s], 2048 * i, target_gap=t0 + step * (2 * i + 1)))
        return out

    def mfmas(self, ks: int) -> List[Instr]:
        return [isa.mfma(ACC(nb, mb), self.fw(ks, nb), self.fx(ks, mb), ACC(nb, mb), tag=f"k{ks}") for nb in range(4) for mb in range(4)]

    def frag_reads(self, slot: int, ks: int, t0: float, step: float) -> List[Instr]:
        out = []
        for i in range(4):
            out.append(isa.ds_read_b128(self.fw(ks, i), WADDR[slot][ks], 4096 * i, target_gap=t0 + step * (2 * i)))
            out.append(isa.ds_read_b128(self.fx(ks, i), XADDR[slot][ks], 4096 * i, target_gap=t0 + step * (2 * i + 1)))
        return out

    def dma_tile(self, slot: int, t0: float, step: float, advance: bool = True) -> List[Instr]:
        """16 LDS-DMA pieces of this wave (8 x rows, 8 W rows: 64 rows of each operand tile) for the k-tile at S_KOFF."""
        out = []
        k = 0
        wp = self.cfg.wpacked
        for lds, offs, rsrc, koff in ((S_XLDS, XDMA, S_XRSRC, S_KOFF), (S_WLDS, WDMA, S_WRSRC, S_WOFF if wp else S_KOFF)):
            for half in range(2):
                out.append(isa.sop("s_add_u32", M0, lds, I32(slot * 65536 + half * 4096), target_gap=t0 + step * k - 0.5))
                for i in range(4):
                    out.append(isa.buffer_load_lds(offs[half * 4 + i], rsrc, koff, 1024 * i, target_gap=t0 + step * k, tag="dma"))
                    k += 1
        if advance:
            out.append(isa.sop("s_add_u32", S_KOFF, S_KOFF, I32(128), target_gap=t0 + step * k))
            out.append(isa.sop("s_min_u32", S_KOFF, S_KOFF, S_KMAX, target_gap=t0 + step * k + 0.1))
            if wp:
                out.append(isa.sop("s_add_u32", S_WOFF, S_WOFF, I32(32768), target_gap=t0 + step * k + 0.2))
                out.append(isa.sop("s_min_u32", S_WOFF, S_WOFF, S_WMAX, target_gap=t0 + step * k + 0.3))
        return out

    def load_tile(self, q: int, t0: float, step: float) -> List[Instr]:
        """16 global loads of this wave (8 pieces of 8 rows x 128 B per operand) of the k-tile at S_KOFF into staging set q."""
        out = []
        k = 0
        for op, offs, rsrc in ((0, XDMA, S_XRSRC), (1, WDMA, S_WRSRC)):
            for i in range(8):
                out.append(isa.buffer_load(4, STG(q, op, i), offs[i], rsrc, S_KOFF, 0, target_gap=t0 + step * k, tag="ld"))
                k += 1
        out.append(isa.sop("s_add_u32", S_KOFF, S_KOFF, I32(128), target_gap=t0 + step * k))
        out.append(isa.sop("s_min_u32", S_KOFF, S_KOFF, S_KMAX, target_gap=t0 + step * k + 0.1))
        return out

    def write_tile(self, q: int, slot: int, t0: float, step: float) -> List[Instr]:
        """staging set q -> LDS slot: piece i of operand op lands at slot + op * 32 KB + 8 KB * wave + 1 KB * i (+ swizzled lane part)."""
        out = []
        k = 0
        for op in range(2):
            for i in range(8):
                out.append(isa.ds_write(16, WRADDR[slot][i & 1], STG(q, op, i), op * 32768 + 1024 * i, target_gap=t0 + step * k, tag="st"))
                k += 1
        return out

    # ---- stage == "spread": every piece does  wait -> ds_write (tile T) -> global load (tile T + 2) into the same registers,
    #      the 16 pieces evenly spread over the tile period (12 before the barrier into the OTHER slot, 4 after it into this slot):
    #      no burst of LDS writes, two full iterations of flight per load, one counted `vmcnt(31)` per piece ----------------------
    PIECES = [(op, i) for op in range(2) for i in range(8)]

    def _piece(self, k: int, q: int, slot: int, gap: float, prev, with_wait=True, with_write=True):
        op, i = self.PIECES[k]
        out = []
        w = isa.waitcnt(vmcnt=31, target_gap=gap)
        w.after = list(prev)
        wr = isa.ds_write(16, WRADDR[slot][i & 1], STG(q, op, i), op * 32768 + 1024 * i, target_gap=gap + 0.05)
        wr.after = [w]
        ld = isa.buffer_load(4, STG(q, op, i), (XDMA if op == 0 else WDMA)[i], S_XRSRC if op == 0 else S_WRSRC, S_KOFF, 0, target_gap=gap + 0.1)
        ld.after = [wr]
        return [w, wr, ld]

    def body_spread(self, p: int, wave: int = 0) -> List[Instr]:
        c = self.cfg
        blk: List[Instr] = []
        reads_p: List[Instr] = []
        for ks in range(3):
            r = self.frag_reads(p, ks + 1, 16.0 * ks + 1.0, c.rd_step)
            reads_p += r
            blk += r + self.mfmas(ks)
        late: List[Instr] = []
        prev: List[Instr] = []
        for n, k in enumerate(range(c.ne, 16)):                     # tile t+1 (set p^1) -> slot p^1, then tile t+3 into the set
            tr = self._piece(k, p ^ 1, p ^ 1, c.late_from + c.late_step * n + c.stagger * wave, prev)
            late += tr
            prev = [tr[-1]]
        adv = [isa.sop("s_add_u32", S_KOFF, S_KOFF, I32(128), target_gap=47.1), isa.sop("s_min_u32", S_KOFF, S_KOFF, S_KMAX, target_gap=47.2)]
        adv[0].after = list(prev)
        w1, bar = isa.waitcnt(lgkmcnt=0, target_gap=47.3), isa.barrier(target_gap=47.5)
        w1.after, bar.after = list(reads_p) + [x for x in late if x.cls == isa.DS_WRITE], [w1]
        early: List[Instr] = []
        prev = [bar, adv[1]]
        for n, k in enumerate(range(c.ne)):                         # tile t+2 (set p) -> slot p, then tile t+4 into the set
            tr = self._piece(k, p, p, 48.0 + (15.0 / max(c.ne, 1)) * n + 0.2 * wave, prev)
            early += tr
            prev = [tr[-1]]
        nxt = self.frag_reads(p ^ 1, 0, 57.0, 0.8)
        for i in nxt:
            i.after = [bar] + [x for x in early if x.cls == isa.DS_WRITE]
        blk += late + adv + [w1, bar] + early + nxt + self.mfmas(3)
        return sched.schedule(blk, cap=c.cap, lookahead=1.0)

    # ---- stage == "dma2": LDS-DMA staging two tiles deep inside the two 64 KB slots ---------------------------------------------
    #   registers hold the fragments of a whole k-tile (4 sets); they are read HALF A TILE ahead of their MFMAs:
    #     gaps  0-31  MFMA k-steps 0, 1 of tile t   ||  read fragments of k-steps 2, 3 of tile t (slot p)
    #     barrier B1 (every wave has read the last fragments of tile t: slot p is free)
    #     gaps ~10-42 the 16 LDS-DMA pieces of tile t+2 -> slot p, one per two gaps
    #     barrier B2 (each wave: vmcnt(pieces of THIS iteration issued so far) -> its pieces of tile t+1, issued one iteration
    #                 ago, have landed; after the barrier everybody's have)
    #     gaps 32-63  MFMA k-steps 2, 3 of tile t   ||  read fragments of k-steps 0, 1 of tile t+1 (slot p^1)
    #   -> a DMA piece has 0.8-1.3 tile periods of flight, a fragment read half a tile, and no LDS write instruction exists.
    def body_dma2(self, p: int) -> List[Instr]:
        c = self.cfg
        abl = c.abl.split(",")
        gs = 2.0 if c.mi == 16 else 1.0           # gaps per 32 matrix-pipe cycles
        b1_at, b2_at = c.b1_at * gs, c.b2_at * gs + (0.5 if c.mi == 16 else 0.0)
        if c.mi == 16:
            rd_a = [] if "lds" in abl else self.frag_reads16(p, 1, 0.0, c.rd2_step * gs)
        else:
            rd_a = [] if "lds" in abl else self.frag_reads(p, 2, 0.0, c.rd2_step) + self.frag_reads(p, 3, 8 * c.rd2_step, c.rd2_step)
        w1, b1 = isa.waitcnt(lgkmcnt=0, target_gap=b1_at - 0.2), isa.barrier(target_gap=b1_at)
        w1.after, b1.after = list(rd_a), list(rd_a) + [w1]
        dma = [] if "dma" in abl else self.dma_tile(p, b1_at + 0.5, c.dma_step * 2.0 * gs)
        for i in dma:
            i.after = [b1]
        w2, b2 = isa.waitcnt(vmcnt=0, target_gap=b2_at - 0.2), isa.barrier(target_gap=b2_at)
        b2.after = [w2, b1]
        w2.after = [b1]
        if c.mi == 16:
            rd_b = [] if "lds" in abl else self.frag_reads16(p ^ 1, 0, b2_at + 0.5, c.rd2_step * gs)
        else:
            rd_b = [] if "lds" in abl else self.frag_reads(p ^ 1, 0, b2_at + 0.5, c.rd2_step) + self.frag_reads(p ^ 1, 1, b2_at + 0.5 + 8 * c.rd2_step, c.rd2_step)
        for i in rd_b:
            i.after = [b2]
        sync1 = [w1] + ([b1] if "bar" not in abl else [])
        sync2 = [w2] + ([b2] if "bar" not in abl else [])
        if c.mi == 16:
            blk = rd_a + sync1 + dma + self.mfmas16(0) + sync2 + rd_b + self.mfmas16(1)
        else:
            blk = rd_a + self.mfmas(0) + sync1 + dma + self.mfmas(1) + sync2 + rd_b + self.mfmas(2) + self.mfmas(3)
        seq = sched.schedule(blk, cap=c.cap, lookahead=c.lookahead)
        n_before = sum(1 for i in seq[:seq.index(w2)] if getattr(i, "tag", "") == "dma")
        w2.vmcnt, w2.mods = n_before, f"vmcnt({n_before})"
        return seq

    def body(self, p: int, wave: int = 0) -> List[Instr]:
        """One k-tile (64 MFMAs) on slot p: k-steps 0-2, barrier, refill of slot p with tile t+2 || k-step 3 || first fragments
        of tile t+1 from slot p^1.  ``wave`` (with cfg.stagger): the loop exists once per wave, each copy issuing its 16 global
        loads in a different part of the tile period -- the four waves of a workgroup run in lock step (one barrier per tile), and
        with one wave per SIMD a load that finds the CU's memory pipeline busy stalls the wave's MFMA stream."""
        c = self.cfg
        if c.stage == "spread":
            return self.body_spread(p, wave)
        if c.stage == "dma2":
            return self.body_dma2(p)
        abl = c.abl.split(",")
        blk: List[Instr] = []
        reads_p: List[Instr] = []
        for ks in range(3):
            r = self.frag_reads(p, ks + 1, 16.0 * ks + 1.0, c.rd_step / 1.0 * 1.0) if "lds" not in abl else []
            reads_p += r
            blk += r
            blk += self.mfmas(ks)
        ld0 = c.ld_from + c.stagger * wave
        loads = self.load_tile(p ^ 1, ld0, c.ld_step) if (c.stage == "reg" and "dma" not in abl and "nold" not in abl) else []
        blk = loads + blk
        w1 = isa.waitcnt(lgkmcnt=0, target_gap=47.3)
        w2 = isa.waitcnt(vmcnt=16 if c.stage == "reg" else 0, target_gap=47.4)
        bar = isa.barrier(target_gap=47.5)
        w1.after, w2.after, bar.after = list(reads_p), (list(loads) if not c.stagger else []), list(reads_p) + [w1, w2]
        sync = [w1, w2] + ([bar] if "bar" not in abl else [])
        blk += sync
        if c.stage == "reg":
            dma = self.write_tile(p, p, c.dma_from, c.dma_step) if ("dma" not in abl and "nost" not in abl) else []
        else:
            dma = self.dma_tile(p, c.dma_from, c.dma_step) if "dma" not in abl else []
        nxt = self.frag_reads(p ^ 1, 0, 49.0, 1.5) if "lds" not in abl else []
        for i in dma + nxt:
            i.after = list(sync)
        if c.stage == "reg":
            for i in nxt:                 # the next tile's first fragment reads are the LAST LDS operations of the body (see loop())
                i.after = list(sync) + list(dma)
                i.target_gap += 8.0
        blk += dma + nxt
        blk += self.mfmas(3)
        seq = sched.schedule(blk, cap=c.cap, lookahead=1.0)      # LDS waits are added by loop(): fragment reads cross body boundaries
        if c.stage == "reg" and w2 in seq:
            # the staging set written after this wait was loaded one iteration earlier: everything older than THIS iteration's
            # loads issued so far must have landed (loads retire in order)
            n_before = sum(1 for i in seq[:seq.index(w2)] if i.cls == isa.VMEM_LOAD)
            w2.vmcnt, w2.mods = n_before, f"vmcnt({n_before})"
        return seq

    # ---------------------------------------------------------------------------------------------
    def addr64_madd(self, ptr: isa.Reg, a, b, shift: int) -> List[Instr]:
        """ptr(64) += (a * b) << shift   (a, b: 32-bit SGPRs / immediates, unsigned)."""
        lo, hi = ST[0], ST[1]
        st = S(ST[2].idx, 2)
        return [isa.sop("s_mul_i32", lo, a, b), isa.sop("s_mul_hi_u32", hi, a, b),
                isa.sop("s_mov_b32", st.sub(0), lo), isa.sop("s_mov_b32", st.sub(1), hi), isa.sop("s_lshl_b64", st, st, I32(shift)),
                isa.sop("s_add_u32", ptr.sub(0), ptr.sub(0), st.sub(0)), isa.sop("s_addc_u32", ptr.sub(1), ptr.sub(1), st.sub(1))]

    def prologue(self) -> List[Instr]:
        c = self.cfg
        o: List[Instr] = [isa.label(c.name)]
        o += [isa.s_load(8, S(8, 8), S_KARG, 0), isa.s_load(4, S(16, 4), S_KARG, 32), isa.s_load(2, S_TAB, S_KARG, 48),
              isa.s_load(8, S(24, 8), S_KARG, 56), isa.s_load(4, S(32, 4), S_KARG, 88),
              isa.vop("v_and_b32", LANE, I32(63), V(0)), isa.vop("v_lshrrev_b32", T_[0], I32(6), V(0)),
              isa.waitcnt(lgkmcnt=0), isa.vop("v_readfirstlane_b32", S_WAVE, T_[0])]
        # tile of this workgroup from the host-built order table
        ent = ST[4]
        o += [isa.sop("s_lshl_b32", ST[5], S_WG, I32(2)), isa.sop("s_add_u32", S_TAB.sub(0), S_TAB.sub(0), ST[5]),
              isa.sop("s_addc_u32", S_TAB.sub(1), S_TAB.sub(1), I32(0)), isa.s_load(1, ent, S_TAB, 0), isa.waitcnt(lgkmcnt=0),
              isa.sop("s_cmp_eq_u32", None, ent, I32(0xFFFFFFFF)), isa.branch("s_cbranch_scc1", "L_exit"),
              isa.sop("s_and_b32", ST[5], ent, I32(0xFFFF)), isa.sop("s_lshr_b32", ST[6], ent, I32(16)),
              isa.sop("s_lshl_b32", S_M0T, ST[5], I32(8)), isa.sop("s_lshl_b32", S_N0T, ST[6], I32(8)),
              isa.sop("s_lshr_b32", S_WM, S_WAVE, I32(1)), isa.sop("s_and_b32", S_WN, S_WAVE, I32(1))]
        # descriptors: x rows from m0 (base += m0 * lda * 2), W rows from n0 (base += n0 * K * 2)
        o += self.addr64_madd(S_X, S_M0T, S_LDA.sub(0), 1) + self.addr64_madd(S_W, S_N0T, S_K, 1)
        for rs, ptr in ((S_XRSRC, S_X), (S_WRSRC, S_W)):
            o += [isa.sop("s_mov_b32", rs.sub(0), ptr.sub(0)), isa.sop("s_and_b32", rs.sub(1), ptr.sub(1), I32(0xFFFF)),
                  isa.sop("s_mov_b32", rs.sub(2), I32(0xFFFFFFFF)), isa.sop("s_mov_b32", rs.sub(3), I32(0x00020000))]
        ldab, kb2 = ST[6], ST[7]
        o += [isa.sop("s_lshl_b32", ldab, S_LDA.sub(0), I32(1)), isa.sop("s_lshl_b32", kb2, S_K, I32(1)),
              isa.sop("s_lshr_b32", S_KT, S_K, I32(6)), isa.sop("s_sub_u32", ST[8], S_KT, I32(1)), isa.sop("s_lshl_b32", S_KMAX, ST[8], I32(7)),
              isa.sop("s_mov_b32", S_KOFF, I32(0)), isa.sop("s_mov_b32", S_T, I32(0)),
              isa.sop("s_mov_b32", S_WOFF, I32(0)), isa.sop("s_lshl_b32", S_WMAX, ST[8], I32(15)),          # (KT - 1) * 32 KB
              isa.sop("s_lshl_b32", S_XLDS, S_WAVE, I32(13)), isa.sop("s_add_u32", S_WLDS, S_XLDS, I32(32768))]
        ql, g, t = T_[1], T_[2], T_
        if c.mi == 16:
            # 16 x 16 x 32 fragments: lane -> row l % 16, 16-byte chunk 4 s + (l / 16) of the 128-byte k-row (s = k32-step)
            o += [isa.vop("v_and_b32", ql, I32(15), LANE), isa.vop("v_lshrrev_b32", g, I32(4), LANE)]
        else:
            o += [isa.vop("v_and_b32", ql, I32(31), LANE), isa.vop("v_lshrrev_b32", g, I32(5), LANE)]
        # fragment read addresses: row r (128 B), chunk (2 ks + g) ^ ((r >> 1) & 7); x rows 128 wm + 32 mb + ql, W rows 128 wn + ...
        o += [isa.vop("v_lshrrev_b32", t[3], I32(1), ql), isa.vop("v_and_b32", t[3], I32(7), t[3]), isa.vop("v_lshlrev_b32", t[4], I32(7), ql),
              isa.vop("v_lshlrev_b32", t[5], I32(14), S_WM), isa.vop("v_add_u32", t[5], t[5], t[4]),                     # x: wm * 128 rows * 128 B
              isa.vop("v_lshlrev_b32", t[6], I32(14), S_WN), isa.vop("v_add_u32", t[6], t[6], t[4]),
              isa.vop("v_add_u32", t[6], I32(32768), t[6])]
        for ks in range(2 if c.mi == 16 else 4):
            o += [isa.vop("v_or_b32", t[7], I32((4 if c.mi == 16 else 2) * ks), g), isa.vop("v_xor_b32", t[7], t[7], t[3]),
                  isa.vop("v_lshl_add_u32", XADDR[0][ks], t[7], I32(4), t[5]), isa.vop("v_lshl_add_u32", WADDR[0][ks], t[7], I32(4), t[6]),
                  isa.vop("v_add_u32", XADDR[1][ks], I32(65536), XADDR[0][ks]), isa.vop("v_add_u32", WADDR[1][ks], I32(65536), WADDR[0][ks])]
        # source offsets: piece i of this wave = tile rows 64 w + 8 i + (lane >> 3), 16-byte chunk lane & 7 of the 128-byte k-row.
        # LDS image: row r, chunk c lives at chunk position c ^ ((r >> 1) & 7).  LDS-DMA writes lane-linearly, so there the SOURCE
        # chunk is permuted; register staging permutes the WRITE address instead (source stays coalesced).
        mlast = ST[9]
        o += [isa.sop("s_sub_u32", mlast, S_M, S_M0T), isa.sop("s_sub_u32", mlast, mlast, I32(1)),         # last valid x row of the tile
              isa.vop("v_lshrrev_b32", t[3], I32(3), LANE), isa.vop("v_and_b32", t[4], I32(7), LANE),
              isa.vop("v_lshlrev_b32", t[5], I32(6), S_WAVE)]
        dma = c.stage in ("dma", "dma2")
        for i in range(8):
            o += [isa.vop("v_add_u32", t[6], I32(8 * i), t[3]), isa.vop("v_add_u32", t[6], t[6], t[5])]           # row in tile
            if dma:
                o += [isa.vop("v_lshrrev_b32", t[7], I32(1), t[6]), isa.vop("v_and_b32", t[7], I32(7), t[7]), isa.vop("v_xor_b32", t[7], t[4], t[7])]
            else:
                o += [isa.vop("v_mov_b32", t[7], t[4])]
            o += [isa.vop("v_min_u32", t[8], t[6], mlast), isa.vop("v_mul_lo_u32", t[8], t[8], ldab),
                  isa.vop("v_lshl_add_u32", t[8], t[7], I32(4), t[8]), isa.vop("v_subrev_u32", XDMA[i], I32(1024 * (i & 3) if dma else 0), t[8]),
                  isa.vop("v_mul_lo_u32", t[9], t[6], kb2), isa.vop("v_lshl_add_u32", t[9], t[7], I32(4), t[9]),
                  isa.vop("v_subrev_u32", WDMA[i], I32(1024 * (i & 3) if dma else 0), t[9])]
            # TIMING ABLATIONS "wpack" / "xpack": every LDS-DMA piece reads 1 KB of CONTIGUOUS memory, as a tile-major pre-packed
            # operand would allow (wrong results: the addresses are not the operand's)
            if c.wpacked:
                # packed W: piece i of this wave = bytes [(8 wave + i) KB, +1 KB) of the tile's 32 KB block, lane-linear (the block is the
                # LDS image: the source swizzle was applied by scail_gemm_pack_w)
                o += [isa.vop("v_lshlrev_b32", t[8], I32(13), S_WAVE), isa.vop("v_lshl_add_u32", t[8], LANE, I32(4), t[8]),
                      isa.vop("v_add_u32", WDMA[i], I32(1024 * i - (1024 * (i & 3) if dma else 0)), t[8])]
            for tag, reg in (("wpack", WDMA[i]), ("xpack", XDMA[i])):
                if tag in c.abl.split(","):
                    o += [isa.vop("v_lshlrev_b32", t[8], I32(13), S_WAVE), isa.vop("v_lshl_add_u32", t[8], LANE, I32(4), t[8]),
                          isa.vop("v_add_u32", reg, I32(1024 * i - (1024 * (i & 3) if dma else 0)), t[8])]
        if not dma:
            # LDS write address of the lane inside a piece: (lane >> 3) * 128 + ((lane & 7) ^ (4 * parity + (lane >> 4))) * 16
            o += [isa.vop("v_lshrrev_b32", t[6], I32(4), LANE), isa.vop("v_lshlrev_b32", t[7], I32(7), t[3]),
                  isa.vop("v_lshlrev_b32", t[8], I32(13), S_WAVE), isa.vop("v_add_u32", t[7], t[7], t[8])]
            for par in range(2):
                o += [isa.vop("v_add_u32", t[8], I32(4 * par), t[6]), isa.vop("v_xor_b32", t[8], t[4], t[8]),
                      isa.vop("v_lshl_add_u32", WRADDR[0][par], t[8], I32(4), t[7]), isa.vop("v_add_u32", WRADDR[1][par], I32(65536), WRADDR[0][par])]
        zero_acc = [isa.vop("v_accvgpr_write_b32", A(i), I32(0)) for i in range(256)]
        # pipeline fill: tiles 0 and 1 in LDS (tile 2 in flight in staging set 0 for the register path), first fragments of tile 0
        if dma:
            # the 32 LDS-DMA pieces go out first and the 256 accumulator clears run under their latency
            o += self.dma_tile(0, 0, 0) + self.dma_tile(1, 0, 0) + zero_acc
            o += [isa.waitcnt(vmcnt=0), isa.barrier()]
        elif c.stage == "spread":
            o += zero_acc
            ld = lambda q, k: isa.buffer_load(4, STG(q, *self.PIECES[k]), (XDMA if self.PIECES[k][0] == 0 else WDMA)[self.PIECES[k][1]],
                                              S_XRSRC if self.PIECES[k][0] == 0 else S_WRSRC, S_KOFF, 0)
            wr = lambda q, slot, k: isa.ds_write(16, WRADDR[slot][self.PIECES[k][1] & 1], STG(q, *self.PIECES[k]),
                                                 self.PIECES[k][0] * 32768 + 1024 * self.PIECES[k][1])
            adv = lambda: [isa.sop("s_add_u32", S_KOFF, S_KOFF, I32(128)), isa.sop("s_min_u32", S_KOFF, S_KOFF, S_KMAX)]
            o += [ld(0, k) for k in range(16)] + adv() + [isa.waitcnt(vmcnt=0)] + [wr(0, 0, k) for k in range(16)]          # tile 0 -> slot 0
            o += [ld(1, k) for k in range(16)] + adv() + [isa.waitcnt(vmcnt=0)] + [wr(1, 1, k) for k in range(c.ne)]        # tile 1: early pieces
            o += [ld(0, k) for k in range(16)] + adv()                                                                        # tile 2 -> set 0
            o += [ld(1, k) for k in range(c.ne)]                                                                              # tile 3 early -> set 1
            o += [isa.waitcnt(lgkmcnt=0), isa.barrier()]
        else:
            o += zero_acc
            o += self.load_tile(0, 0, 0) + self.load_tile(1, 0, 0) + [isa.waitcnt(vmcnt=0)]
            o += self.write_tile(0, 0, 0, 0) + self.write_tile(1, 1, 0, 0) + self.load_tile(0, 0, 0)
            o += [isa.waitcnt(lgkmcnt=0), isa.barrier()]
        o = sched.pad_hazards(sched.insert_lgkm_waits(o))
        return o + self.first_reads()          # the 8 reads stay in flight into the first body (see loop())

    def first_reads(self) -> List[Instr]:
        """The first fragment reads of tile 0 (slot 0), issued by the prologue in the SAME order in which a loop body leaves the next
        tile's first reads in flight, so that the counted LDS waits at the top of a body are right on both ways into it."""
        if "lds" in self.cfg.abl.split(","):
            return []
        tail: List[Instr] = []
        sched.insert_lgkm_waits(self.body(1), carry_in=[], carry_out=tail)
        n = self.n_carry
        order = [tuple(i.writes()) for i in tail[-n:]]
        if self.cfg.mi == 16:
            first = self.frag_reads16(0, 0, 0, 0)
        else:
            first = self.frag_reads(0, 0, 0, 0) + (self.frag_reads(0, 1, 0, 0) if n == 16 else [])
        reads = {tuple(i.writes()): i for i in first}
        assert sorted(order) == sorted(reads), "the last LDS operations of a body must be the next tile's first fragment reads"
        return [reads[k] for k in order]

    @property
    def n_carry(self) -> int:
        """fragment reads a body leaves in flight for its successor"""
        return 16 if self.cfg.stage == "dma2" else 8

    def loop(self) -> List[Instr]:
        """The k-loop, two bodies (slot 0 / slot 1).  The first fragments of tile t+1 are read at the end of body t and consumed
        at the start of body t+1: the counted LDS waits of a body start from the 8 reads its predecessor left in flight."""
        sig = lambda q: [tuple(i.writes()) for i in q]
        first = self.first_reads()
        c0: List[Instr] = []
        c1: List[Instr] = []
        # (LDS operations retire in order: as long as the 8 reads are the youngest operations in flight when a body ends, the
        # counts that wait for them do not depend on what older operations -- the staging writes -- are still queued before them)
        b0 = sched.insert_lgkm_waits(self.body(0), carry_in=first, carry_out=c0)
        b1 = sched.insert_lgkm_waits(self.body(1), carry_in=first, carry_out=c1)
        if "nowait" in self.cfg.abl.split(","):           # timing ablation: fragment reads never waited for (stale operands)
            b0 = [i for i in b0 if not (i.op == "s_waitcnt" and getattr(i, "lgkmcnt", None) not in (None, 0))]
            b1 = [i for i in b1 if not (i.op == "s_waitcnt" and getattr(i, "lgkmcnt", None) not in (None, 0))]
        if "lds" not in self.cfg.abl.split(",") and "nost" not in self.cfg.abl.split(","):
            n = self.n_carry
            assert sig(c0[-n:]) == sig(first) and sig(c1[-n:]) == sig(first), "a body must end with the next tile's first fragment reads in flight"
        o: List[Instr] = []
        waves = range(4) if self.cfg.stagger else range(1)
        if self.cfg.stagger:
            for wv in (1, 2, 3):
                o += [isa.sop("s_cmp_eq_u32", None, S_WAVE, I32(wv)), isa.branch("s_cbranch_scc1", f"L_loop_w{wv}")]
        for wv in waves:
            if wv:
                b0 = sched.insert_lgkm_waits(self.body(0, wv), carry_in=first)
                b1 = sched.insert_lgkm_waits(self.body(1, wv), carry_in=first)
            o += [isa.label(f"L_loop_w{wv}")]
            o += b0
            o += [isa.sop("s_add_u32", S_T, S_T, I32(1)), isa.sop("s_cmp_lt_u32", None, S_T, S_KT), isa.branch("s_cbranch_scc0", "L_done")]
            o += b1
            o += [isa.sop("s_add_u32", S_T, S_T, I32(1)), isa.sop("s_cmp_lt_u32", None, S_T, S_KT), isa.branch("s_cbranch_scc1", f"L_loop_w{wv}")]
            if wv != waves[-1]:
                o += [isa.branch("s_branch", "L_done")]
        o += [isa.label("L_done"), isa.waitcnt(vmcnt=0), isa.waitcnt(lgkmcnt=0), isa.nop(15), isa.nop(15)]
        return o

    # ---------------------------------------------------------------------------------------------
    STG_ROW = 272          # bytes per staged row: 256 + 16 (conflict-free 8-byte writes of 16 rows, 16-byte aligned chunks)
    STG_WAVE = 16 * 272

    def epilogue_staged(self) -> List[Instr]:
        """mi = 16, whole-line epilogue.  Accumulator layout: lane (v = l % 16, g = l / 16) owns row 16 mb + v, columns 16 nb + 4 g .. + 3 of the
        wave's 128 x 128 block; memory layout of a row block: chunk j = 64 i + l (i = 0..3) = row j / 16, 16-byte chunk j % 16 of the row's 256 bytes.
        Per row block: [residual rows: 4 whole-line loads -> LDS -> 8 pairs in accumulator layout] -> bias / GELU / gate / residual in fp32 ->
        packed pairs -> LDS -> 4 quads in memory layout -> 4 whole-line stores.  Loads of block mb + 1 are requested before block mb is waited
        for (counted vmcnt: vector-memory operations retire in issue order)."""
        c = self.cfg
        assert c.mi == 16
        e: List[Instr] = []
        ql, g, t = T_[1], T_[2], T_
        n_mb = 8
        BQ = lambda nb: V(nb * 4, 4)
        LQ = lambda par, i: V(32 + 16 * par + 4 * i, 4)        # residual row block in memory layout, two in flight
        RB = lambda nb: V(64 + 2 * nb, 2)                       # ... back in accumulator layout
        OUTP = lambda nb: V(80 + 2 * nb, 2)                     # packed outputs, accumulator layout
        GQ = lambda par, nb: V(96 + 32 * par + 4 * nb, 4)       # gate quads, two row blocks in flight
        RQ = lambda i: V(202 + 4 * i, 4)                        # output row block in memory layout
        ROW = [V(164), V(197)]                                  # accumulator-layout row of the lane, by block parity (gate batch index, e3)
        GOFF = [V(165), V(198)]
        TB = V(166)
        E_W = V(236)
        E_R = [V(237 + i) for i in range(4)]
        E_Y = [V(241 + i) for i in range(4)]
        E_Z = [V(245 + i) for i in range(4)]
        E_M = [V(249 + i) for i in range(4)]                    # memory-layout row of the lane's chunk (row mask), advanced per block
        nw = ST[4]            # n0 + 128 wn (first column of the wave)
        e += [isa.sop("s_lshl_b32", nw, S_WN, I32(7)), isa.sop("s_add_u32", nw, nw, S_N0T)]
        e += self.addr64_madd(S_Y, nw, I32(1), 1)
        for i in range(32):
            e.append(isa.vop("v_mov_b32", V(i), I32(0)))
        e += [isa.sop("s_cmp_eq_u64", None, S_BIAS, I32(0)), isa.branch("s_cbranch_scc1", "L_nobias")]
        e += self.addr64_madd(S_BIAS, nw, I32(1), 2)
        e += [isa.vop("v_lshlrev_b32", t[3], I32(4), g)]                       # 4 g floats = 16 g bytes
        for nb in range(8):
            e.append(isa.global_load(4, BQ(nb), t[3], nb * 64, saddr=S_BIAS))
        e += [isa.label("L_nobias")]
        mw = ST[5]
        e += [isa.sop("s_lshl_b32", mw, S_WM, I32(7)), isa.sop("s_add_u32", mw, mw, S_M0T)]
        ldcb, ldrb = ST[6], ST[7]
        e += [isa.sop("s_lshl_b32", ldcb, S_LDC.sub(0), I32(1))]
        RCP, KC0, KC1 = V(161), V(162), V(163)
        if c.epi in (3, 4):
            e += self.addr64_madd(S_RES, nw, I32(1), 1)
            e += [isa.sop("s_lshl_b32", ldrb, S_LDR.sub(0), I32(1))]
        if c.epi == 3:
            e += self.addr64_madd(S_GATE, nw, I32(1), 2)
            e += [isa.vop("v_cvt_f32_u32", RCP, S_RPB), isa.vop("v_rcp_f32", RCP, RCP),
                  isa.sop("s_cmp_eq_u32", None, S_RPB, I32(0)), isa.sop("s_cselect_b32", ST[9], I32(0), I32(0xFFFFFFFF)),
                  isa.sop("s_lshl_b32", ST[12], S_GS.sub(0), I32(2))]
        if c.epi == 1:
            K0, K1 = 0.7978845608028654, 0.044715
            sc = 2.0 * 1.4426950408889634
            e += [isa.vop("v_mov_b32", KC0, F32(K0 * sc)), isa.vop("v_mov_b32", KC1, F32(K0 * K1 * sc))]
        # staging addresses and the lane's place in the memory layout
        sb = ST[8]
        e += [isa.sop("s_mul_i32", sb, S_WAVE, I32(self.STG_WAVE)), isa.sop("s_add_u32", sb, sb, I32(131072)),
              isa.vop("v_mul_u32_u24", t[4], I32(self.STG_ROW), ql), isa.vop("v_lshl_add_u32", t[4], g, I32(3), t[4]), isa.vop("v_add_u32", E_W, sb, t[4]),
              isa.vop("v_lshlrev_b32", t[5], I32(4), ql)]                                     # chunk l % 16 -> 16 (l % 16) bytes
        for i in range(4):
            e += [isa.vop("v_add_u32", t[4], I32(4 * i), g),                                  # row of the block: 4 i + l / 16
                  isa.vop("v_add_u32", E_M[i], mw, t[4]),
                  isa.vop("v_mul_u32_u24", t[6], I32(self.STG_ROW), t[4]), isa.vop("v_add_u32", t[6], t[6], t[5]), isa.vop("v_add_u32", E_R[i], sb, t[6]),
                  isa.vop("v_mul_lo_u32", t[6], E_M[i], ldcb), isa.vop("v_add_u32", E_Y[i], t[6], t[5])]
            if c.epi in (3, 4):
                e += [isa.vop("v_mul_lo_u32", t[6], E_M[i], ldrb), isa.vop("v_add_u32", E_Z[i], t[6], t[5])]
        step_y, step_z = ST[10], ST[11]
        e += [isa.sop("s_lshl_b32", step_y, ldcb, I32(4)), isa.sop("s_lshl_b32", step_z, ldrb, I32(4))]          # 16 rows on

        def masked(i, ins_list):
            return ([isa.v_cmp("v_cmp_lt_u32", E_M[i], S_M),
                     Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)] + ins_list +
                    [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)])

        def loads(mb):
            """residual rows (memory layout; E_M / E_Z point at block mb) and the gate quads of the lane's row (accumulator layout)."""
            o_ = []
            for i in range(4):
                o_ += masked(i, [isa.global_load(4, LQ(mb & 1, i), E_Z[i], 0, saddr=S_RES, extra_reads=[EXEC])])
            if c.epi == 3:
                row, goff = ROW[mb & 1], GOFF[mb & 1]
                o_ += [isa.vop("v_add_u32", row, mw, ql)] + ([isa.vop("v_add_u32", row, I32(16 * mb), row)] if mb else [])
                # batch index of the row: floor((m + 0.5) / rows_per_batch); rows_per_batch == 0 -> 0
                o_ += [isa.vop("v_cvt_f32_u32", TB, row), isa.vop("v_add_f32", TB, F32(0.5), TB), isa.vop("v_mul_f32", TB, TB, RCP),
                       isa.vop("v_cvt_u32_f32", TB, TB), isa.vop("v_and_b32", TB, ST[9], TB),
                       isa.vop("v_mul_lo_u32", goff, TB, ST[12]), isa.vop("v_lshl_add_u32", goff, g, I32(4), goff),
                       isa.v_cmp("v_cmp_lt_u32", row, S_M),
                       Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)]
                for nb in range(8):
                    o_.append(isa.global_load(4, GQ(mb & 1, nb), goff, nb * 64, saddr=S_GATE, extra_reads=[EXEC]))
                o_ += [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]
            return o_

        def advance_z():
            return [isa.vop("v_add_u32", E_Z[i], step_z, E_Z[i]) for i in range(4)]

        n_loads = {0: 0, 1: 0, 4: 4, 3: 12}[c.epi]
        # E_M / E_Y follow the block being STORED, E_Z the block being LOADED (one ahead); the load masks of block mb + 1 use E_M + 16
        if n_loads:
            e += loads(0) + advance_z()
        e.append(isa.waitcnt(vmcnt=n_loads))             # the bias quads (older than the first block's loads)
        for mb in range(n_mb):
            if n_loads:
                if mb + 1 < n_mb:
                    # block mb + 1's loads: rows 16 further (E_M is advanced after block mb's stores: mask with a temporary)
                    for i in range(4):
                        e.append(isa.vop("v_add_u32", E_M[i], I32(16), E_M[i]))
                    e += loads(mb + 1) + advance_z()
                    for i in range(4):
                        e.append(isa.vop("v_subrev_u32", E_M[i], I32(16), E_M[i]))
                # behind the loads of block mb: the stores of block mb - 1 (4) and the loads of block mb + 1
                n_after = (4 if mb >= 1 else 0) + (n_loads if mb + 1 < n_mb else 0)
                e.append(isa.waitcnt(vmcnt=n_after))
                e += [isa.ds_write(16, E_R[i], LQ(mb & 1, i)) for i in range(4)]
                e += [isa.ds_read(8, RB(nb), E_W, 32 * nb) for nb in range(8)]
            for nb in range(8):
                base = 170 + 16 * (nb % 2)         # two rotating register groups
                f = [V(base + i) for i in range(4)]
                r_, u2 = V(base + 8), [V(base + 9), V(base + 10)]
                acc = ACC16(nb, mb)
                for i in range(4):
                    e += [isa.vop("v_accvgpr_read_b32", f[i], acc.sub(i)), isa.vop("v_add_f32", f[i], f[i], BQ(nb).sub(i))]
                if c.epi == 1:
                    for i in range(4):           # gelu_tanh(v) = v - v / (exp2(2 log2e k0 (v + k1 v^3)) + 1)
                        u = u2[i & 1]
                        e += [isa.vop("v_mul_f32", u, f[i], f[i]), isa.vop("v_fma_f32", u, u, KC1, KC0),
                              isa.vop("v_mul_f32", u, u, f[i]), isa.vop("v_exp_f32", u, u), isa.vop("v_add_f32", u, F32(1.0), u),
                              isa.vop("v_rcp_f32", u, u), isa.vop("v_fma_f32", f[i], Neg(f[i]), u, f[i])]
                if c.epi in (3, 4):
                    if c.epi == 3:
                        for i in range(4):
                            e.append(isa.vop("v_mul_f32", f[i], f[i], GQ(mb & 1, nb).sub(i)))
                    for i in range(4):           # + residual (bf16 pairs: low half << 16, high half & 0xffff0000)
                        src = RB(nb).sub(i >> 1)
                        e += [isa.vop("v_lshlrev_b32", r_, I32(16), src) if (i & 1) == 0 else isa.vop("v_and_b32", r_, I32(0xFFFF0000), src),
                              isa.vop("v_add_f32", f[i], f[i], r_)]
                e += [isa.vop("v_cvt_pk_bf16_f32", OUTP(nb).sub(0), f[0], f[1]), isa.vop("v_cvt_pk_bf16_f32", OUTP(nb).sub(1), f[2], f[3])]
            e += [isa.ds_write(8, E_W, OUTP(nb), 32 * nb) for nb in range(8)]
            e += [isa.ds_read_b128(RQ(i), E_R[i]) for i in range(4)]
            for i in range(4):
                st = isa.global_store(4, E_Y[i], RQ(i), 0, saddr=S_Y, extra_reads=[EXEC])
                if c.nt_store:
                    st.text += " nt"
                e += masked(i, [st])
            if mb + 1 < n_mb:
                for i in range(4):
                    e += [isa.vop("v_add_u32", E_M[i], I32(16), E_M[i]), isa.vop("v_add_u32", E_Y[i], step_y, E_Y[i])]
        e += [isa.label("L_exit"), Instr("s_endpgm", cls=isa.BRANCH)]       # (stores may still be in flight: the hardware drains them)
        return sched.pad_hazards(sched.insert_lgkm_waits(e))

    def epilogue(self) -> List[Instr]:
        """mi = 32: y[m][n .. n+3] for the lane's rows m = m0 + 128 wm + 32 mb + (lane & 31), n = n0 + 128 wn + 32 nb + 8 rr + 4 g;
        mi = 16: rows m = m0 + 128 wm + 16 mb + (lane & 15), n = n0 + 128 wn + 16 nb + 4 (lane >> 4) -- in both layouts a lane owns
        quads of 4 consecutive n, `quads` below lists them as (n offset, accumulator quad of row block mb, bias quad)."""
        c = self.cfg
        if c.stg:
            return self.epilogue_staged()
        e: List[Instr] = []
        ql, g, t = T_[1], T_[2], T_
        if c.mi == 16:
            n_mb, rows_mb = 8, 16
            quads = [(16 * nb, (lambda mb, nb=nb: ACC16(nb, mb)), V(nb * 4, 4)) for nb in range(8)]
        else:
            n_mb, rows_mb = 4, 32
            quads = [(32 * nb + 8 * rr, (lambda mb, nb=nb, rr=rr: A((nb * 4 + mb) * 16 + 4 * rr, 4)), V(nb * 16 + rr * 4, 4))
                     for nb in range(4) for rr in range(4)]
        nw = ST[4]            # n0 + 128 wn (first column of the wave)
        e += [isa.sop("s_lshl_b32", nw, S_WN, I32(7)), isa.sop("s_add_u32", nw, nw, S_N0T)]
        e += self.addr64_madd(S_Y, nw, I32(1), 1)
        # bias quads (fragment registers are dead): zeros when bias == NULL
        for i in range(64):
            e.append(isa.vop("v_mov_b32", V(i), I32(0)))
        e += [isa.sop("s_cmp_eq_u64", None, S_BIAS, I32(0)), isa.branch("s_cbranch_scc1", "L_nobias")]
        e += self.addr64_madd(S_BIAS, nw, I32(1), 2)
        e += [isa.vop("v_lshlrev_b32", t[3], I32(4), g)]                       # 4 g floats = 16 g bytes
        for noff, _, bq_ in quads:
            e.append(isa.global_load(4, bq_, t[3], noff * 4, saddr=S_BIAS))
        e += [isa.waitcnt(vmcnt=0), isa.label("L_nobias")]
        # rows
        mw = ST[5]
        e += [isa.sop("s_lshl_b32", mw, S_WM, I32(7)), isa.sop("s_add_u32", mw, mw, S_M0T)]
        ldcb = ST[6]
        e += [isa.sop("s_lshl_b32", ldcb, S_LDC.sub(0), I32(1))]
        RCP, KC0, KC1 = V(161), V(162), V(163)
        if c.epi in (3, 4):
            e += self.addr64_madd(S_RES, nw, I32(1), 1)
            e += [isa.sop("s_lshl_b32", ST[7], S_LDR.sub(0), I32(1))]
        if c.epi == 3:
            e += self.addr64_madd(S_GATE, nw, I32(1), 2)
            e += [isa.vop("v_cvt_f32_u32", RCP, S_RPB), isa.vop("v_rcp_f32", RCP, RCP),
                  isa.sop("s_cmp_eq_u32", None, S_RPB, I32(0)), isa.sop("s_cselect_b32", ST[9], I32(0), I32(0xFFFFFFFF)),
                  isa.sop("s_lshl_b32", ST[12], S_GS.sub(0), I32(2))]
        if c.epi == 1:
            K0, K1 = 0.7978845608028654, 0.044715
            sc = 2.0 * 1.4426950408889634
            e += [isa.vop("v_mov_b32", KC0, F32(K0 * sc)), isa.vop("v_mov_b32", KC1, F32(K0 * K1 * sc))]
        # Row blocks as a two-stage software pipeline (epilogues 3 / 4 read a residual row block, 3 also the gate of the row's batch
        # element): the operands of block mb + 1 are requested BEFORE block mb is waited for, with a COUNTED wait -- loads return in
        # order, so "at most <what was issued after them> outstanding" means block mb's have landed; the stores of block mb - 1
        # are never waited for (a first version that did not count them as allowed gained only +1.8 %: every block waited for the
        # write acknowledgements of the previous one).  Round 2 waited vmcnt(0) after every quad's loads: 64 dependent L2 round trips per lane, ~9 % of a K = 5120 tile's time
        # (out-projection 1253 vs 1364 TFLOP/s for the K = 13 824 MLP-down, profiles/r03_gemm_table_modes.log).
        # Registers (the fragment / staging registers v0..v159 are dead here): bias quads v0.., residual pairs v64.. and gate quads
        # v96.. in two sets by block parity (mi = 32, measurement build: one set, uncounted wait).
        pipelined = c.mi == 16 and c.epi in (3, 4)
        nq = len(quads)
        RP = lambda par, k: V(64 + (16 * par if pipelined else 0) + 2 * k, 2)
        GQ = lambda par, k: V(96 + (32 * par if pipelined else 0) + 4 * k, 4)
        ADDR = lambda par: [V((164 if par == 0 or not pipelined else 204) + i) for i in range(5)]        # row, yoff, roff, goff, bq

        def addresses(mb):
            row, yoff, roff, goff, bq = ADDR(mb & 1)
            o_ = [isa.vop("v_add_u32", row, mw, ql)]
            if mb:
                o_ += [isa.vop("v_add_u32", row, I32(rows_mb * mb), row)]
            o_ += [isa.vop("v_mul_lo_u32", yoff, row, ldcb), isa.vop("v_lshl_add_u32", yoff, g, I32(3), yoff)]
            if c.epi in (3, 4):
                o_ += [isa.vop("v_mul_lo_u32", roff, row, ST[7]), isa.vop("v_lshl_add_u32", roff, g, I32(3), roff)]
            if c.epi == 3:
                # batch index of the row: floor((m + 0.5) / rows_per_batch); rows_per_batch == 0 -> 0
                o_ += [isa.vop("v_cvt_f32_u32", bq, row), isa.vop("v_add_f32", bq, F32(0.5), bq), isa.vop("v_mul_f32", bq, bq, RCP),
                       isa.vop("v_cvt_u32_f32", bq, bq), isa.vop("v_and_b32", bq, ST[9], bq),
                       isa.vop("v_mul_lo_u32", goff, bq, ST[12]), isa.vop("v_lshl_add_u32", goff, g, I32(4), goff)]
            return o_

        def mask(mb):       # EXEC = rows of this block that exist (the full mask stays in S_SAVE)
            return [isa.v_cmp("v_cmp_lt_u32", ADDR(mb & 1)[0], S_M),
                    Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)]

        unmask = lambda: [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]

        def loads(mb):
            _, _, roff, goff, _ = ADDR(mb & 1)
            o_ = []
            for k, (noff, accq, bq_) in enumerate(quads):
                o_.append(isa.global_load(2, RP(mb & 1, k), roff, noff * 2, saddr=S_RES, extra_reads=[EXEC]))
                if c.epi == 3:
                    o_.append(isa.global_load(4, GQ(mb & 1, k), goff, noff * 4, saddr=S_GATE, extra_reads=[EXEC]))
            return o_

        n_loads = nq * (2 if c.epi == 3 else 1)
        if pipelined:
            e += addresses(0) + mask(0) + loads(0) + unmask()
        for mb in range(n_mb):
            if pipelined:
                if mb + 1 < n_mb:
                    e += addresses(mb + 1) + mask(mb + 1) + loads(mb + 1) + unmask()
                # vector-memory operations retire in issue order (loads and stores share the counter on gfx9 / CDNA): behind the
                # loads of block mb there are the stores of block mb - 1 and the loads of block mb + 1 -- neither has to finish
                n_after = (nq if mb >= 1 else 0) + (n_loads if mb + 1 < n_mb else 0)
                e += mask(mb) + [isa.waitcnt(vmcnt=n_after)]
            else:
                e += addresses(mb) + mask(mb)
                if c.epi in (3, 4):
                    e += loads(mb) + [isa.waitcnt(vmcnt=0)]
            yoff = ADDR(mb & 1)[1]
            for k, (noff, accq, bq_) in enumerate(quads):
                base = 170 + 16 * (k % 2)         # two rotating register groups
                f = [V(base + i) for i in range(4)]
                w, r_, u2 = V(base + 4, 2), V(base + 8), [V(base + 9), V(base + 10)]
                rp, gq = RP(mb & 1, k), GQ(mb & 1, k)
                acc = accq(mb)
                for i in range(4):
                    e += [isa.vop("v_accvgpr_read_b32", f[i], acc.sub(i)),
                          isa.vop("v_add_f32", f[i], f[i], bq_.sub(i))]
                if c.epi == 1:
                    for i in range(4):           # gelu_tanh(v) = v - v / (exp2(2 log2e k0 (v + k1 v^3)) + 1)
                        u = u2[i & 1]
                        e += [isa.vop("v_mul_f32", u, f[i], f[i]), isa.vop("v_fma_f32", u, u, KC1, KC0),
                              isa.vop("v_mul_f32", u, u, f[i]), isa.vop("v_exp_f32", u, u), isa.vop("v_add_f32", u, F32(1.0), u),
                              isa.vop("v_rcp_f32", u, u), isa.vop("v_fma_f32", f[i], Neg(f[i]), u, f[i])]
                if c.epi in (3, 4):
                    if c.epi == 3:
                        for i in range(4):
                            e.append(isa.vop("v_mul_f32", f[i], f[i], gq.sub(i)))
                    for i in range(4):           # + residual (bf16 pairs: low half << 16, high half & 0xffff0000)
                        src = rp.sub(i >> 1)
                        e += [isa.vop("v_lshlrev_b32", r_, I32(16), src) if (i & 1) == 0 else isa.vop("v_and_b32", r_, I32(0xFFFF0000), src),
                              isa.vop("v_add_f32", f[i], f[i], r_)]
                st = isa.global_store(2, yoff, w, noff * 2, saddr=S_Y, extra_reads=[EXEC])
                if c.nt_store:
                    st.text += " nt"
                e += [isa.vop("v_cvt_pk_bf16_f32", w.sub(0), f[0], f[1]), isa.vop("v_cvt_pk_bf16_f32", w.sub(1), f[2], f[3]), st]
            e += [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]
        e += [isa.label("L_exit"), isa.waitcnt(vmcnt=0), Instr("s_endpgm", cls=isa.BRANCH)]
        return self._pad_between_labels(e)

    @staticmethod
    def _pad_between_labels(seq: List[Instr]) -> List[Instr]:
        return sched.pad_hazards(seq)

    # =============================================================================================
    # PERSISTENT form (cfg.persist; dma2 + 16x16x32 only).  A 512-register workgroup has no co-resident partner, so everything
    # around the k-loop of a tile is exposed: workgroup launch, the descriptor / address set-up, the latency of the first two k-tiles
    # and the epilogue -- 9 % of a K = 5120 tile (profiles/r03_gemm_epilogue_probe.log: t(K) of the bias-only kernel).  Here a
    # workgroup walks table entries b, b + grid, b + 2 grid, ... (grid is a multiple of 8, so it stays on its XCD's sequence, and the
    # 32 workgroups of an XCD still work on 32 consecutive entries of it), and at the end of a tile it
    #   1. requests what the epilogue needs of the CURRENT tile (bias quads, residual / gate operands of the first two row blocks),
    #   2. looks up its NEXT tile, re-points the descriptors and issues that tile's first two k-tiles (32 LDS-DMA pieces per wave)
    #      -- always, also when there is no next tile (the pieces then re-read the last tile; the operation counts behind the
    #      counted waits stay static),
    #   3. runs the epilogue of the current tile under the latency of those pieces (it touches neither LDS nor v64..v95),
    #   4. clears the accumulators and enters the loop again.
    # Register plan of the epilogue: bias quads v0..v31, residual pairs v96..v127 and gate quads v128..v191 in two sets by row-block
    # parity, constants / addresses / working groups v192..v226; v64..v95 (fragment and DMA source addresses) stay live.
    # =============================================================================================
    def tile_setup(self, tag: str) -> List[Instr]:
        """S_E -> S_M0N / S_N0N, descriptors and x source offsets of that tile (S_NVALID = 1), or S_NVALID = 0 and nothing touched."""
        o: List[Instr] = []
        ent, t = ST[13], T_        # scalar temporaries s69..s71, s64, s65: the epilogue keeps its own in ST[4..6], ST[10..12]
        o += [isa.sop("s_mov_b32", S_NVALID, I32(0)), isa.sop("s_cmp_lt_u32", None, S_E, S_ENT), isa.branch("s_cbranch_scc0", f"L_none{tag}"),
              isa.sop("s_lshl_b32", ST[14], S_E, I32(2)), isa.sop("s_add_u32", S_TABC.sub(0), S_TAB.sub(0), ST[14]),
              isa.sop("s_addc_u32", S_TABC.sub(1), S_TAB.sub(1), I32(0)), isa.s_load(1, ent, S_TABC, 0), isa.waitcnt(lgkmcnt=0),
              isa.sop("s_cmp_eq_u32", None, ent, I32(0xFFFFFFFF)), isa.branch("s_cbranch_scc1", f"L_none{tag}"),
              isa.sop("s_mov_b32", S_NVALID, I32(1)),
              isa.sop("s_and_b32", ST[14], ent, I32(0xFFFF)), isa.sop("s_lshr_b32", ST[15], ent, I32(16)),
              isa.sop("s_lshl_b32", S_M0N, ST[14], I32(8)), isa.sop("s_lshl_b32", S_N0N, ST[15], I32(8)),
              Instr("s_mov_b64", [S_XC], [S_X], cls=isa.SALU), Instr("s_mov_b64", [S_WC], [S_W], cls=isa.SALU)]
        o += self.addr64_madd(S_XC, S_M0N, S_LDA.sub(0), 1) + self.addr64_madd(S_WC, S_N0N, S_K, 1)
        for rs, ptr in ((S_XRSRC, S_XC), (S_WRSRC, S_WC)):
            o += [isa.sop("s_mov_b32", rs.sub(0), ptr.sub(0)), isa.sop("s_and_b32", rs.sub(1), ptr.sub(1), I32(0xFFFF))]
        # x source offsets: piece i of this wave = tile rows 64 w + 8 i + (lane >> 3) clamped to the tile's last valid row, chunk
        # (lane & 7) ^ ((row >> 1) & 7) of the 128-byte k-row (as in prologue(); W's do not depend on the tile)
        mlast, ldab = ST[8], ST[9]
        o += [isa.sop("s_sub_u32", mlast, S_M, S_M0N), isa.sop("s_sub_u32", mlast, mlast, I32(1)),
              isa.sop("s_lshl_b32", ldab, S_LDA.sub(0), I32(1)),
              isa.vop("v_lshrrev_b32", t[3], I32(3), LANE), isa.vop("v_and_b32", t[4], I32(7), LANE), isa.vop("v_lshlrev_b32", t[5], I32(6), S_WAVE)]
        for i in range(8):
            o += [isa.vop("v_add_u32", t[6], I32(8 * i), t[3]), isa.vop("v_add_u32", t[6], t[6], t[5]),
                  isa.vop("v_lshrrev_b32", t[7], I32(1), t[6]), isa.vop("v_and_b32", t[7], I32(7), t[7]), isa.vop("v_xor_b32", t[7], t[4], t[7]),
                  isa.vop("v_min_u32", t[8], t[6], mlast), isa.vop("v_mul_lo_u32", t[8], t[8], ldab),
                  isa.vop("v_lshl_add_u32", t[8], t[7], I32(4), t[8]), isa.vop("v_subrev_u32", XDMA[i], I32(1024 * (i & 3)), t[8])]
        o += [isa.label(f"L_none{tag}"), isa.nop(7),
              isa.sop("s_mov_b32", S_KOFF, I32(0))]
        return o

    def program_persistent(self) -> List[Instr]:
        c = self.cfg
        assert c.stage == "dma2" and c.mi == 16 and not c.wpacked and not c.stagger
        # ---- once per workgroup: everything of prologue() that does not depend on the tile --------------------------------------
        o: List[Instr] = [isa.label(c.name)]
        o += [isa.s_load(8, S(8, 8), S_KARG, 0), isa.s_load(4, S(16, 4), S_KARG, 32), isa.s_load(2, S_TAB, S_KARG, 48),
              isa.s_load(8, S(24, 8), S_KARG, 56), isa.s_load(4, S(32, 4), S_KARG, 88), isa.s_load(2, S(S_G.idx, 2), S_KARG, 104),
              isa.vop("v_and_b32", LANE, I32(63), V(0)), isa.vop("v_lshrrev_b32", T_[0], I32(6), V(0)),
              isa.waitcnt(lgkmcnt=0), isa.vop("v_readfirstlane_b32", S_WAVE, T_[0]),
              isa.sop("s_mov_b32", S_E, S_WG),
              isa.sop("s_lshr_b32", S_WM, S_WAVE, I32(1)), isa.sop("s_and_b32", S_WN, S_WAVE, I32(1))]
        for rs in (S_XRSRC, S_WRSRC):
            o += [isa.sop("s_mov_b32", rs.sub(2), I32(0xFFFFFFFF)), isa.sop("s_mov_b32", rs.sub(3), I32(0x00020000))]
        kb2 = ST[7]
        o += [isa.sop("s_lshl_b32", kb2, S_K, I32(1)),
              isa.sop("s_lshr_b32", S_KT, S_K, I32(6)), isa.sop("s_sub_u32", ST[8], S_KT, I32(1)), isa.sop("s_lshl_b32", S_KMAX, ST[8], I32(7)),
              isa.sop("s_lshl_b32", S_XLDS, S_WAVE, I32(13)), isa.sop("s_add_u32", S_WLDS, S_XLDS, I32(32768))]
        ql, g, t = T_[1], T_[2], T_
        o += [isa.vop("v_and_b32", ql, I32(15), LANE), isa.vop("v_lshrrev_b32", g, I32(4), LANE)]
        o += [isa.vop("v_lshrrev_b32", t[3], I32(1), ql), isa.vop("v_and_b32", t[3], I32(7), t[3]), isa.vop("v_lshlrev_b32", t[4], I32(7), ql),
              isa.vop("v_lshlrev_b32", t[5], I32(14), S_WM), isa.vop("v_add_u32", t[5], t[5], t[4]),
              isa.vop("v_lshlrev_b32", t[6], I32(14), S_WN), isa.vop("v_add_u32", t[6], t[6], t[4]),
              isa.vop("v_add_u32", t[6], I32(32768), t[6])]
        for ks in range(2):
            o += [isa.vop("v_or_b32", t[7], I32(4 * ks), g), isa.vop("v_xor_b32", t[7], t[7], t[3]),
                  isa.vop("v_lshl_add_u32", XADDR[0][ks], t[7], I32(4), t[5]), isa.vop("v_lshl_add_u32", WADDR[0][ks], t[7], I32(4), t[6]),
                  isa.vop("v_add_u32", XADDR[1][ks], I32(65536), XADDR[0][ks]), isa.vop("v_add_u32", WADDR[1][ks], I32(65536), WADDR[0][ks])]
        # W source offsets (tile independent: the descriptor base carries n0)
        o += [isa.vop("v_lshrrev_b32", t[3], I32(3), LANE), isa.vop("v_and_b32", t[4], I32(7), LANE), isa.vop("v_lshlrev_b32", t[5], I32(6), S_WAVE)]
        for i in range(8):
            o += [isa.vop("v_add_u32", t[6], I32(8 * i), t[3]), isa.vop("v_add_u32", t[6], t[6], t[5]),
                  isa.vop("v_lshrrev_b32", t[7], I32(1), t[6]), isa.vop("v_and_b32", t[7], I32(7), t[7]), isa.vop("v_xor_b32", t[7], t[4], t[7]),
                  isa.vop("v_mul_lo_u32", t[9], t[6], kb2), isa.vop("v_lshl_add_u32", t[9], t[7], I32(4), t[9]),
                  isa.vop("v_subrev_u32", WDMA[i], I32(1024 * (i & 3)), t[9])]
        # ---- the first tile: nothing to overlap with ------------------------------------------------------------------------------
        o += self.tile_setup("_first")
        o += [isa.sop("s_cmp_eq_u32", None, S_NVALID, I32(0)), isa.branch("s_cbranch_scc1", "L_exit"),
              isa.sop("s_mov_b32", S_M0T, S_M0N), isa.sop("s_mov_b32", S_N0T, S_N0N)]
        zero_acc = [isa.vop("v_accvgpr_write_b32", A(i), I32(0)) for i in range(256)]
        o += self.dma_tile(0, 0, 0) + self.dma_tile(1, 0, 0) + zero_acc
        o += [isa.waitcnt(vmcnt=0), isa.barrier(), isa.label("L_enter"), isa.nop(7), isa.sop("s_mov_b32", S_T, I32(0))]
        o = sched.pad_hazards(sched.insert_lgkm_waits(o))
        o += self.first_reads()
        # ---- the k-loop (unchanged) -------------------------------------------------------------------------------------------------
        o += self.loop()                                  # ends: L_done, vmcnt(0), lgkmcnt(0)
        # ---- tile end ---------------------------------------------------------------------------------------------------------------
        e: List[Instr] = [isa.barrier()]                  # every wave has read its last fragments: both LDS slots are free
        n_mb, rows_mb = 8, 16
        quads = [(16 * nb, (lambda mb, nb=nb: ACC16(nb, mb)), V(nb * 4, 4)) for nb in range(8)]
        nq = len(quads)
        nw, mw, ldcb = ST[4], ST[5], ST[6]
        e += [isa.sop("s_lshl_b32", nw, S_WN, I32(7)), isa.sop("s_add_u32", nw, nw, S_N0T)]
        e += [Instr("s_mov_b64", [S_YC], [S_Y], cls=isa.SALU)] + self.addr64_madd(S_YC, nw, I32(1), 1)
        for i in range(32):
            e.append(isa.vop("v_mov_b32", V(i), I32(0)))
        # 1. what the epilogue needs of this tile: bias quads ...
        e += [isa.sop("s_cmp_eq_u64", None, S_BIAS, I32(0)), isa.branch("s_cbranch_scc1", "L_nobias"),
              Instr("s_mov_b64", [S_BC], [S_BIAS], cls=isa.SALU)]
        e += self.addr64_madd(S_BC, nw, I32(1), 2)
        e += [isa.vop("v_lshlrev_b32", t[3], I32(4), g)]
        for noff, _, bq_ in quads:
            e.append(isa.global_load(4, bq_, t[3], noff * 4, saddr=S_BC))
        e += [isa.label("L_nobias"), isa.nop(7)]
        e += [isa.sop("s_lshl_b32", mw, S_WM, I32(7)), isa.sop("s_add_u32", mw, mw, S_M0T)]
        RCP, KC0, KC1 = V(192), V(193), V(194)
        ldrb = ST[10]
        if c.epi in (3, 4):
            e += [Instr("s_mov_b64", [S_RC], [S_RES], cls=isa.SALU)] + self.addr64_madd(S_RC, nw, I32(1), 1)
            e += [isa.sop("s_lshl_b32", ldrb, S_LDR.sub(0), I32(1))]
        if c.epi == 3:
            e += [Instr("s_mov_b64", [S_GC], [S_GATE], cls=isa.SALU)] + self.addr64_madd(S_GC, nw, I32(1), 2)
            e += [isa.vop("v_cvt_f32_u32", RCP, S_RPB), isa.vop("v_rcp_f32", RCP, RCP),
                  isa.sop("s_cmp_eq_u32", None, S_RPB, I32(0)), isa.sop("s_cselect_b32", ST[11], I32(0), I32(0xFFFFFFFF)),
                  isa.sop("s_lshl_b32", ST[12], S_GS.sub(0), I32(2))]
        if c.epi == 1:
            K0, K1 = 0.7978845608028654, 0.044715
            sc = 2.0 * 1.4426950408889634
            e += [isa.vop("v_mov_b32", KC0, F32(K0 * sc)), isa.vop("v_mov_b32", KC1, F32(K0 * K1 * sc))]
        e += [isa.sop("s_lshl_b32", ldcb, S_LDC.sub(0), I32(1))]
        RP = lambda par, k: V(96 + 16 * par + 2 * k, 2)
        GQ = lambda par, k: V(128 + 32 * par + 4 * k, 4)
        ADDR = lambda par: [V(195 + 5 * par + i) for i in range(5)]        # row, yoff, roff, goff, bq

        def addresses(mb):
            row, yoff, roff, goff, bq = ADDR(mb & 1)
            o_ = [isa.vop("v_add_u32", row, mw, ql)]
            if mb:
                o_ += [isa.vop("v_add_u32", row, I32(rows_mb * mb), row)]
            o_ += [isa.vop("v_mul_lo_u32", yoff, row, ldcb), isa.vop("v_lshl_add_u32", yoff, g, I32(3), yoff)]
            if c.epi in (3, 4):
                o_ += [isa.vop("v_mul_lo_u32", roff, row, ldrb), isa.vop("v_lshl_add_u32", roff, g, I32(3), roff)]
            if c.epi == 3:
                o_ += [isa.vop("v_cvt_f32_u32", bq, row), isa.vop("v_add_f32", bq, F32(0.5), bq), isa.vop("v_mul_f32", bq, bq, RCP),
                       isa.vop("v_cvt_u32_f32", bq, bq), isa.vop("v_and_b32", bq, ST[11], bq),
                       isa.vop("v_mul_lo_u32", goff, bq, ST[12]), isa.vop("v_lshl_add_u32", goff, g, I32(4), goff)]
            return o_

        def mask(mb):
            return [isa.v_cmp("v_cmp_lt_u32", ADDR(mb & 1)[0], S_M),
                    Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)]

        unmask = lambda: [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]

        def loads(mb):
            _, _, roff, goff, _ = ADDR(mb & 1)
            o_ = []
            if c.epi in (3, 4):
                for k, (noff, accq, bq_) in enumerate(quads):
                    o_.append(isa.global_load(2, RP(mb & 1, k), roff, noff * 2, saddr=S_RC, extra_reads=[EXEC]))
                    if c.epi == 3:
                        o_.append(isa.global_load(4, GQ(mb & 1, k), goff, noff * 4, saddr=S_GC, extra_reads=[EXEC]))
            return o_

        n_loads = nq * (2 if c.epi == 3 else 1 if c.epi == 4 else 0)
        # ... and the residual / gate operands of the first two row blocks (both register sets are free here)
        for mb in (0, 1):
            e += addresses(mb) + mask(mb) + loads(mb) + unmask()
        # 2. the next tile: descriptors, source offsets, its first two k-tiles (issued unconditionally, see above)
        e += [isa.sop("s_add_u32", S_E, S_E, S_G)]
        e += self.tile_setup("_next")
        n_dma = 32
        e += self.dma_tile(0, 0, 0) + self.dma_tile(1, 0, 0)
        # 3. the epilogue of the current tile.  Vector-memory operations retire in issue order: a counted wait names how many
        #    operations issued AFTER the wanted ones may still be in flight.
        for mb in range(n_mb):
            if mb >= 1 and mb + 1 < n_mb:
                e += addresses(mb + 1) + mask(mb + 1) + loads(mb + 1) + unmask()
            if mb == 0:
                after = n_loads + n_dma                       # block 1's operands, the DMA pieces
            elif mb == 1:
                after = n_dma + nq + n_loads                  # the DMA pieces, block 0's stores, block 2's operands
            else:
                after = nq + (n_loads if mb + 1 < n_mb else 0)
            e += mask(mb) + [isa.waitcnt(vmcnt=min(after, 63))]
            yoff = ADDR(mb & 1)[1]
            for k, (noff, accq, bq_) in enumerate(quads):
                base = 205 + 11 * (k % 2)
                f = [V(base + i) for i in range(4)]
                w, r_, u2 = V(base + 4 + ((base + 4) & 1), 2), V(base + 7), [V(base + 8), V(base + 9)]
                rp, gq = RP(mb & 1, k), GQ(mb & 1, k)
                acc = accq(mb)
                for i in range(4):
                    e += [isa.vop("v_accvgpr_read_b32", f[i], acc.sub(i)), isa.vop("v_add_f32", f[i], f[i], bq_.sub(i))]
                if c.epi == 1:
                    for i in range(4):
                        u = u2[i & 1]
                        e += [isa.vop("v_mul_f32", u, f[i], f[i]), isa.vop("v_fma_f32", u, u, KC1, KC0),
                              isa.vop("v_mul_f32", u, u, f[i]), isa.vop("v_exp_f32", u, u), isa.vop("v_add_f32", u, F32(1.0), u),
                              isa.vop("v_rcp_f32", u, u), isa.vop("v_fma_f32", f[i], Neg(f[i]), u, f[i])]
                if c.epi in (3, 4):
                    if c.epi == 3:
                        for i in range(4):
                            e.append(isa.vop("v_mul_f32", f[i], f[i], gq.sub(i)))
                    for i in range(4):
                        src = rp.sub(i >> 1)
                        e += [isa.vop("v_lshlrev_b32", r_, I32(16), src) if (i & 1) == 0 else isa.vop("v_and_b32", r_, I32(0xFFFF0000), src),
                              isa.vop("v_add_f32", f[i], f[i], r_)]
                st = isa.global_store(2, yoff, w, noff * 2, saddr=S_YC, extra_reads=[EXEC])
                if c.nt_store:
                    st.text += " nt"
                e += [isa.vop("v_cvt_pk_bf16_f32", w.sub(0), f[0], f[1]), isa.vop("v_cvt_pk_bf16_f32", w.sub(1), f[2], f[3]), st]
            e += unmask()
        # 4. next tile (its DMA pieces are older than everything the epilogue issued: with at most 63 operations left in flight
        #    they have landed) or the end
        e += [isa.sop("s_cmp_eq_u32", None, S_NVALID, I32(0)), isa.branch("s_cbranch_scc1", "L_exit"),
              isa.sop("s_mov_b32", S_M0T, S_M0N), isa.sop("s_mov_b32", S_N0T, S_N0N)]
        e += [isa.vop("v_accvgpr_write_b32", A(i), I32(0)) for i in range(256)]
        e += [isa.waitcnt(vmcnt=0 if (n_loads * 6 + nq * 8) < 63 else 63), isa.barrier(), isa.branch("s_branch", "L_enter")]
        e += [isa.label("L_exit"), isa.waitcnt(vmcnt=0), Instr("s_endpgm", cls=isa.BRANCH)]
        prog = o + sched.pad_hazards(sched.insert_lgkm_waits(e))
        return prog

    def program(self) -> List[Instr]:
        if self.cfg.persist:
            prog = self.program_persistent()
            pre = f"L_{self.cfg.name}"
            for i in prog:
                if i.label and i.label.startswith("L_"):
                    new = pre + i.label[1:]
                    if getattr(i, "text", None):
                        i.text = i.text.replace(i.label, new)
                    i.label = new
            return prog
        prog = self.prologue() + self.loop() + self.epilogue()
        pre = f"L_{self.cfg.name}"
        for i in prog:
            if i.label and i.label.startswith("L_"):
                new = pre + i.label[1:]
                if getattr(i, "text", None):
                    i.text = i.text.replace(i.label, new)
                i.label = new
        return prog


HEAD = """// GENERATED by scail_amd/asmgen/gemm4.py -- do not edit; regenerate with `python -m scail_amd.asmgen.gemm4`.
// Hand-scheduled 4-wave bf16 GEMM for gfx950 (256 x 256 x 64 tile, accumulators in a[0:255]); see the generator's docstring.
\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
"""


def kernel_text(c: Cfg) -> str:
    body = isa.render(Gen(c).program())
    return f"""// ---- kernel {c.name}: epilogue {c.epi}, <= {c.cap} fillers per MFMA gap ----
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
\t\t.amdhsa_group_segment_fixed_size {131072 + 4 * Gen.STG_WAVE if c.stg else 131072}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {KERNARG_SIZE}
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
        .size:           {KERNARG_SIZE}
        .value_kind:     by_value
    .group_segment_fixed_size: {131072 + 4 * Gen.STG_WAVE if c.stg else 131072}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {KERNARG_SIZE}
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


# the shipped kernels: LDS-DMA two tiles deep, v_mfma_f32_16x16x32_bf16, <= 1 filler per 16-cycle MFMA gap, slot released at gap 11,
# one DMA piece per 7 gaps (measured best of the sweep in profiles/r02_gemm4_mi16.log: 1467 TFLOP/s where q8 does 1309, the vendor 1495)
# Round 3: the epilogue goes through LDS so that every global store / residual load moves whole cache lines (stg): +3.1 % qkv, +6.6 % out-projection,
# +3.5 / +5.2 % cross q / out, +0.3 % MLP up (GELU: VALU-bound), +2.8 % MLP down, bit-identical (profiles/r03_gemm_staged_epilogue_ab.log)
SHIPPED = dict(stage="dma2", mi=16, cap=1, rd2_step=0.25, b1_at=5.5, dma_step=1.75, stg=True)
DEFAULTS = [Cfg(epi=e, name=f"scail_gemm4_e{e}", **SHIPPED) for e in (0, 1, 3, 4)]


def pack_w(w_bits, N: int, K: int):
    """the tile-major weight layout of the ``wpacked`` experiment (test infrastructure for the emulator): (N, K) uint16 -> packed.
    Block (n tile, k tile) = 32 pieces x 64 lanes x 8 elements; lane l of piece p holds row 8 p + (l >> 3) of the tile, source
    chunk (l & 7) ^ ((row >> 1) & 7) -- the LDS image the kernels' fragment reads expect."""
    import numpy as np
    tn, tk = N // 256, K // 64
    w4 = w_bits.reshape(tn, 256, tk, 64)                                  # [n tile][row][k tile][k]
    out = np.zeros((tn, tk, 32, 64, 8), dtype=w_bits.dtype)
    lane = np.arange(64)
    for p in range(32):
        row = 8 * p + (lane >> 3)
        ch = (lane & 7) ^ ((row >> 1) & 7)
        cols = ch[:, None] * 8 + np.arange(8)[None, :]
        blk = w4[:, row[:, None], :, cols]                                # advanced indices first: (64, 8, tn, tk)
        out[:, :, p] = np.transpose(blk, (2, 3, 0, 1))
    return out.reshape(-1)


def variant_cfgs():
    out = [Cfg(epi=e, name=f"scail_gemm4_e{e}_reg") for e in (0, 1, 3, 4)]       # round-2 first version: register staging, 32x32x16
    PART = {**SHIPPED, "stg": False}
    out += [Cfg(epi=e, name=f"scail_gemm4_e{e}_pst", persist=True, **PART) for e in (0, 1, 3, 4)]         # round 3: persistent workgroups (partial-line epilogue)
    out += [Cfg(epi=e, name=f"scail_gemm4_e{e}_part", **PART) for e in (0, 1, 3, 4)]                      # the epilogue before the LDS staging: 64 stores of 16 rows x 32 bytes per lane
    out += [Cfg(epi=e, name=f"scail_gemm4_e{e}_stgnt", nt_store=True, **SHIPPED) for e in (0, 1, 3, 4)]   # staged + non-temporal stores (measured: no gain)
    for cap in (2, 4):
        out.append(Cfg(epi=0, cap=cap, name=f"scail_gemm4_e0_c{cap}"))
    out.append(Cfg(epi=0, stage="dma", name="scail_gemm4_e0_lds_dma"))
    out.append(Cfg(epi=0, stage="dma2", name="scail_gemm4_e0_dma2"))
    out.append(Cfg(epi=0, stage="dma2", cap=2, name="scail_gemm4_e0_dma2_c2"))
    out.append(Cfg(epi=0, stage="dma2", rd2_step=1.0, b1_at=17.5, name="scail_gemm4_e0_dma2_rd1"))
    out.append(Cfg(epi=0, stage="dma2", dma_step=0.5, name="scail_gemm4_e0_dma2_d05"))
    out.append(Cfg(epi=0, stage="dma2", mi=16, cap=2, name="scail_gemm4_e0_mi16"))
    out.append(Cfg(epi=0, stage="dma2", mi=16, cap=1, name="scail_gemm4_e0_mi16_c1"))
    out.append(Cfg(epi=0, stage="dma2", mi=16, cap=3, name="scail_gemm4_e0_mi16_c3"))
    M16 = dict(epi=0, stage="dma2", mi=16, cap=1)
    out.append(Cfg(**M16, rd2_step=1.0, b1_at=17.5, name="scail_gemm4_e0_mi16_rd1"))
    out.append(Cfg(**M16, rd2_step=0.25, b1_at=5.5, name="scail_gemm4_e0_mi16_rd025"))
    out.append(Cfg(**M16, dma_step=0.75, name="scail_gemm4_e0_mi16_d075"))
    out.append(Cfg(**M16, dma_step=1.25, name="scail_gemm4_e0_mi16_d125"))
    out.append(Cfg(**M16, rd2_step=0.25, b1_at=5.5, dma_step=1.4, name="scail_gemm4_e0_mi16_early"))
    out.append(Cfg(**M16, dma_step=1.5, name="scail_gemm4_e0_mi16_d15"))
    out.append(Cfg(epi=0, name="scail_gemm4p_e0", wpacked=True, **SHIPPED))
    out.append(Cfg(epi=3, name="scail_gemm4p_e3", wpacked=True, **SHIPPED))
    for abl in ("wpack", "xpack", "wpack,xpack"):
        out.append(Cfg(epi=0, abl=abl, name="scail_gemm4_e0_abl_" + abl.replace(",", "_"), **SHIPPED))
    # round 6: timing ablations of the SHIPPED schedule (wrong results on purpose): what each instruction class of the k-loop costs beside the MFMAs
    for abl in ("dma", "lds", "bar", "dma,lds", "dma,lds,bar"):
        out.append(Cfg(epi=0, abl=abl, name="scail_gemm4_e0_s_abl_" + abl.replace(",", "_"), **SHIPPED))
    out.append(Cfg(**M16, rd2_step=0.25, b1_at=5.5, dma_step=1.75, name="scail_gemm4_e0_mi16_early175"))
    out.append(Cfg(**M16, rd2_step=0.25, b1_at=5.5, dma_step=1.9, name="scail_gemm4_e0_mi16_early19"))
    out.append(Cfg(**M16, rd2_step=0.25, b1_at=4.5, dma_step=1.85, name="scail_gemm4_e0_mi16_early185"))
    out.append(Cfg(epi=0, stage="dma2", mi=16, cap=2, rd2_step=0.25, b1_at=5.5, dma_step=1.75, name="scail_gemm4_e0_mi16_early175c2"))
    out.append(Cfg(**M16, dma_step=1.25, nt_store=True, name="scail_gemm4_e0_mi16_d125nt"))
    out.append(Cfg(**M16, dma_step=1.25, lookahead=2.0, name="scail_gemm4_e0_mi16_d125la2"))
    out.append(Cfg(**M16, b2_at=27.5, name="scail_gemm4_e0_mi16_b2e"))
    out.append(Cfg(**M16, b2_at=39.5, name="scail_gemm4_e0_mi16_b2l"))
    out.append(Cfg(epi=0, stage="dma2", mi=16, cap=2, abl="dma", name="scail_gemm4_e0_mi16_abl_dma"))
    out.append(Cfg(epi=0, stage="dma2", mi=16, cap=2, abl="lds", name="scail_gemm4_e0_mi16_abl_lds"))
    out.append(Cfg(epi=0, stage="dma2", mi=16, cap=2, abl="dma,lds", name="scail_gemm4_e0_mi16_abl_dma_lds"))
    out.append(Cfg(epi=0, stage="dma2", abl="dma", name="scail_gemm4_e0_dma2_abl_dma"))
    out.append(Cfg(epi=0, stage="dma2", abl="lds", name="scail_gemm4_e0_dma2_abl_lds"))
    out.append(Cfg(epi=0, stage="dma2", abl="dma,lds", name="scail_gemm4_e0_dma2_abl_dma_lds"))
    out.append(Cfg(epi=0, ld_from=1.0, ld_step=1.0, name="scail_gemm4_e0_ld1"))
    out.append(Cfg(epi=0, ld_from=16.0, ld_step=1.5, name="scail_gemm4_e0_ldmid"))
    out.append(Cfg(epi=0, dma_step=0.5, name="scail_gemm4_e0_st05"))
    for abl in ("dma", "lds", "bar", "dma,lds", "nowait", "nold", "nost", "dma,nowait"):
        out.append(Cfg(epi=0, abl=abl, name="scail_gemm4_e0_abl_" + abl.replace(",", "_")))
    out.append(Cfg(epi=0, stage="spread", name="scail_gemm4_e0_spread"))
    out.append(Cfg(epi=0, stage="spread", stagger=0.9, name="scail_gemm4_e0_spreadstag"))
    out.append(Cfg(epi=0, stage="spread", stagger=0.9, ne=2, late_step=3.2, name="scail_gemm4_e0_spread2"))
    out.append(Cfg(epi=0, stagger=16.0, ld_step=1.0, name="scail_gemm4_e0_stag16"))
    out.append(Cfg(epi=0, stagger=12.0, ld_step=0.75, name="scail_gemm4_e0_stag12"))
    out.append(Cfg(epi=0, stagger=0.5, ld_step=2.0, name="scail_gemm4_e0_stag05"))
    out.append(Cfg(epi=0, rd_step=0.0, cap=9, name="scail_gemm4_e0_rburst"))
    out.append(Cfg(epi=0, rd_step=0.75, name="scail_gemm4_e0_rd075"))
    return out


def main():
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "csrc", "gemm4.s")
    text = assembly(DEFAULTS)
    if "--check" in sys.argv:
        sys.exit(0 if open(out).read() == text else 1)
    if not os.path.exists(out) or open(out).read() != text:
        open(out, "w").write(text)
    print(out, len(text.splitlines()), "lines")


if __name__ == "__main__":
    main()
