"""conv4 -- hand-scheduled 3x3x3 causal convolution for the Wan2.1 VAE on gfx950 (generator of csrc/conv4.s).

Reference: CausalConv3d (sgm/models/wan_vae.py:17-36: F.pad(2 frames in front, 1 voxel around) + nn.Conv3d(k = 3, stride 1)) as
used by every ResidualBlock (wan_vae.py:162-205) -- 26 of the 33 decoder convolutions, > 90 % of the VAE's FLOPs.  Same C entry
point as the kernels of csrc/conv.hip (scail_conv3d_cl), which selects this kernel for kt = kh = kw = 3, stride 1, Cin % 32 == 0,
N % 96 == 0.

Shape (the gemm4 / attn4 structure: 4 waves, one per SIMD, accumulators in the AGPR file, LDS-DMA staging, MFMA-spine scheduling):
  * channels-last activations x (T, H, W, Cin), y (T, H, W, N) bf16; weights (N, Kpad) bf16 with k = ((dt 3 + dh) 3 + dw) Cin + c
    (scail_amd.ops.prep_conv_weight); output frame t sees input frames t - 2 .. t (causal), rows / columns -1 .. +1.
  * workgroup = 2 output frames x 16 x 16 voxels x 96 output channels.  Wave w: frame w >> 1, rows 8 (w & 1) .. + 7 -> 8 voxel
    blocks of 16 (one patch row each) x 6 channel blocks of 16 = 48 v_mfma_f32_16x16x32_bf16 per (tap, 32-channel slice);
    192 accumulators per lane in a[0:191]; x fragments (MFMA B operand) in a[192:255] (two sets), W fragments (A) in v[0:47].
  * the input PATCH of a 32-channel slice -- 4 frames x 18 x 18 voxels -- lives in LDS in a CHUNK-PLANAR layout: plane c (0..3) holds
    the 16-byte chunk c (8 channels) of every patch voxel, voxel (p, r, col) at  c * 24576 + p * 6144 + (r * 18 + col) * 16.
    Written by LDS-DMA (64 consecutive voxels of a plane per instruction, wave w loads patch frame w; out-of-range voxels and the
    causal frames t < 0 read zeros through the buffer descriptor's range check), read as B fragments: lane (voxel l % 16, chunk
    l / 16) -> 16 consecutive voxels of 4 planes = all 64 banks exactly once, and a tap is a plain immediate offset
    ((dt * 384 + dh * 18 + dw) * 16): no swizzle, no padding, no per-tap address arithmetic.
  * the W tile of one tap (96 rows x 64 B) arrives by LDS-DMA three taps ahead into three 8 KB buffers (rows unpadded, 16-byte chunk
    XOR-ed with (row >> 1) & 3 on the source address: conflict-free A-fragment reads); one s_barrier per tap.
  * per tap: 48 MFMAs || the 14 fragment reads of the next tap || 2 W-tile DMA pieces; 27 taps unrolled, a loop over the slices.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import List

from . import isa, sched
from .isa import A, S, V, I32, F32, VCC, EXEC, M0, Instr

KERNARG_SIZE = 128
# x w bias y resid | Ti To H W | Cin N Kpad pt | tiles_h tiles_w tiles_n magic_n | magic_w magic_h n_slices ot_mul | ot_off pad | ldc ldr
KERNARG_FMT = "<5Q4i4i3iI2I2ii4x2q"

TH, TW, NF = 16, 16, 2
PR, PC = TH + 2, TW + 2                 # patch rows / columns
FVOX = 384                              # voxels reserved per patch frame (324 used; 6 DMA pieces of 64)
PLANE = 4 * FVOX * 16                   # bytes per chunk plane: 24576
PATCH = 4 * PLANE                       # 98304
WBUF = 8192                             # bytes per W tap buffer (96 rows x 64 B used; 8 DMA pieces of 1 KB)
NWB = 3
LDS_BYTES = PATCH + NWB * WBUF          # 122880
OOB = 0x7FFFFF00                        # lane offset that fails every descriptor's range check (-> zeros)


def magic31(d: int) -> int:
    return -(-(1 << 31) // d)


def pack_args(x, w, bias, y, resid, Ti, To, H, W, Cin, N, Kpad, pt=2, ot_mul=1, ot_off=0, ldc=0, ldr=0) -> bytes:
    """ldc / ldr: output / residual row strides in elements (0 = N)."""
    th, tw, tn = (H + TH - 1) // TH, (W + TW - 1) // TW, (N + 95) // 96
    b = struct.pack(KERNARG_FMT, x, w, bias, y, resid, Ti, To, H, W, Cin, N, Kpad, pt, th, tw, tn, magic31(tn), magic31(tw), magic31(th),
                    Cin // 32, ot_mul, ot_off, ldc or N, ldr or N)
    assert len(b) == KERNARG_SIZE, len(b)
    return b


def grid_blocks(T, H, W, N) -> int:
    return ((T + NF - 1) // NF) * ((H + TH - 1) // TH) * ((W + TW - 1) // TW) * ((N + 95) // 96)


@dataclass
class Cfg:
    epi: int = 0            # 0: y = conv + bias;  3: y = resid + conv + bias
    cap: int = 1
    lookahead: float = 2.0
    name: str = "scail_conv4_e0"
    rd_at: float = 1.0      # first gap of the next tap's 14 fragment reads, rd_step apart
    rd_step: float = 3.0
    dma_at: float = 4.0     # gap of the first W-tile DMA piece of tap + 3, the second dma_step later
    dma_step: float = 20.0
    abl: str = ""


# ---- registers ------------------------------------------------------------------------------------------------------------------
def ACC(nb, mb): return A((nb * 8 + mb) * 4, 4)
def XF(s, mb): return A(192 + s * 32 + mb * 4, 4)          # x fragments (B operand), set s
def WF(s, nb): return V(s * 24 + nb * 4, 4)                 # W fragments (A operand), set s


PB = V(48)                                                  # per-lane patch fragment base
WB = V(49)                                                  # per-lane W fragment base (inside a buffer)
PDMA = [V(50 + k) for k in range(6)]                        # per-lane source offsets of the 6 voxel groups of a patch frame
WDMA = [V(56 + i) for i in range(2)]                        # per-lane source offsets of this wave's 2 W-tile pieces
LANE = V(58)
T_ = [V(60 + i) for i in range(60)]                         # v60..v119 temporaries (prologue / epilogue)

S_KARG = S(0, 2)
S_WG = S(2)
S_X, S_Wp, S_BIAS, S_Y, S_RES = S(8, 2), S(10, 2), S(12, 2), S(14, 2), S(16, 2)
S_TI, S_T, S_H, S_Wd = S(20), S(21), S(22), S(23)            # input frames, output frames, rows, columns
S_CIN, S_N, S_KPAD, S_PT = S(24), S(25), S(26), S(27)
S_TLH, S_TLW, S_TLN, S_MGN = S(28), S(29), S(30), S(31)
S_MGW, S_MGH, S_NSL, S_OTM = S(32), S(33), S(34), S(35)
S_OTO = S(36)
S_LDC, S_LDR = S(40, 2), S(42, 2)
S_XRW = S(44, 4)                                            # buffer descriptor of THIS wave's patch frame (frame = wave)
S_WR = S(48, 4)                                             # W descriptor of this workgroup's 96 rows
S_WAVE, S_F, S_RH = S(52), S(53), S(54)
S_T0, S_H0, S_W0, S_N0 = S(55), S(56), S(57), S(58)
S_SL, S_XOFF, S_WNEXT, S_CIN2 = S(59), S(60), S(61), S(62)   # slice counter, channel byte offset of the slice, W source offset of the next DMA tap, 2 Cin
S_PLDS, S_WLDS = S(63), S(64)                               # LDS byte offset of this wave's patch frame / W piece
S_FB = S(66, 2)                                             # input frame bytes (64 bit)
S_YF, S_RF = S(68, 2), S(70, 2)                             # output / residual frame base of this wave
ST = [S(72 + i) for i in range(16)]                         # s72..s87 temporaries
S_SAVE = S(88, 2)


class Gen:
    def __init__(self, cfg: Cfg):
        self.cfg = cfg

    # ---- building blocks -------------------------------------------------------------------------------------------------------
    @staticmethod
    def tap_off(tap: int, mb: int) -> int:
        dt, dh, dw = tap // 9, (tap // 3) % 3, tap % 3
        return (dt * FVOX + (mb + dh) * PC + dw) * 16

    def mfmas(self, s: int) -> List[Instr]:
        # n-block major: the A fragment (W) stays for 8 consecutive MFMAs
        return [isa.mfma16(ACC(nb, mb), WF(s, nb), XF(s, mb), ACC(nb, mb), tag="mm") for nb in range(6) for mb in range(8)]

    def frag_reads(self, s: int, tap: int, t0: float, step: float) -> List[Instr]:
        """fragments of ``tap`` into set s: 6 W quads from W buffer tap % 3, 8 x quads from the patch, in the order the MFMAs want them."""
        out = []
        wbuf = tap % NWB
        order = [("w", 0)] + [("x", mb) for mb in range(8)] + [("w", nb) for nb in range(1, 6)]
        for k, (kind, i) in enumerate(order):
            if kind == "w":
                out.append(isa.ds_read_b128(WF(s, i), WB, wbuf * WBUF + i * 1024, target_gap=t0 + step * k))
            else:
                out.append(isa.ds_read_b128(XF(s, i), PB, self.tap_off(tap, i), target_gap=t0 + step * k))
        return out

    def w_dma(self, wbuf: int, t0: float, step: float) -> List[Instr]:
        """this wave's 2 pieces (of 8: pieces w and w + 4; rows >= 96 fail the range check = zeros) of the W tile at S_WNEXT."""
        out = []
        for i in range(2):
            out.append(isa.sop("s_add_u32", M0, S_WLDS, I32(wbuf * WBUF + 4096 * i), target_gap=t0 + step * i - 0.5))
            out.append(isa.buffer_load_lds(WDMA[i], S_WR, S_WNEXT, 0, target_gap=t0 + step * i, tag="dma"))
        return out

    def patch_dma(self) -> List[Instr]:
        """this wave's patch frame (frame = wave) of the current slice: 4 chunk planes x 6 voxel groups = 24 pieces."""
        out = []
        for c in range(4):
            for k in range(6):
                out.append(isa.sop("s_add_u32", M0, S_PLDS, I32(c * PLANE + k * 1024 - 16 * c)))     # the instruction offset moves the LDS side too
                out.append(isa.buffer_load_lds(PDMA[k], S_XRW, S_XOFF, 16 * c, tag="pdma"))
        return out

    def tap_block(self, tap: int) -> List[Instr]:
        """One tap of the unrolled slice body, scheduled: top (fragments of this tap in registers, W tile of the next tap landed,
        barrier) + 48 MFMAs with the next tap's fragment reads and the W DMA of tap + 3 in their gaps."""
        c = self.cfg
        abl = c.abl.split(",")
        s = tap & 1
        top = [isa.waitcnt(lgkmcnt=0), isa.waitcnt(vmcnt=2)] + ([] if "bar" in abl else [isa.barrier()])
        blk: List[Instr] = []
        if tap + 1 < 27 and "lds" not in abl:
            blk += self.frag_reads(s ^ 1, tap + 1, c.rd_at, c.rd_step)
        if "dma" not in abl:
            blk += self.w_dma(tap % NWB, c.dma_at, c.dma_step)
            # advance the W stream: the tap just requested was tap + 3 of this slice (or (tap + 3) - 27 of the next one)
            nxt = tap + 3
            if nxt % 27 == 26:      # its successor is tap 0 of the following slice: back 26 taps, forward one slice (64 bytes)
                blk += [isa.sop("s_mul_i32", ST[0], S_CIN2, I32(26), target_gap=c.dma_at + c.dma_step + 1.0),
                        isa.sop("s_sub_u32", S_WNEXT, S_WNEXT, ST[0], target_gap=c.dma_at + c.dma_step + 1.2),
                        isa.sop("s_add_u32", S_WNEXT, S_WNEXT, I32(64), target_gap=c.dma_at + c.dma_step + 1.4)]
            else:
                blk += [isa.sop("s_add_u32", S_WNEXT, S_WNEXT, S_CIN2, target_gap=c.dma_at + c.dma_step + 1.0)]
        blk += self.mfmas(s)
        seq = sched.schedule(blk, cap=c.cap, lookahead=c.lookahead)
        return top + seq

    # ---- prologue ----------------------------------------------------------------------------------------------------------------
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
        o += [isa.s_load(8, S(8, 8), S_KARG, 0), isa.s_load(2, S_RES, S_KARG, 32), isa.s_load(4, S(20, 4), S_KARG, 40),
              isa.s_load(4, S(24, 4), S_KARG, 56), isa.s_load(4, S(28, 4), S_KARG, 72), isa.s_load(4, S(32, 4), S_KARG, 88),
              isa.s_load(1, S_OTO, S_KARG, 104), isa.s_load(4, S(40, 4), S_KARG, 112),
              isa.vop("v_and_b32", LANE, I32(63), V(0)), isa.vop("v_lshrrev_b32", T_[0], I32(6), V(0)),
              isa.waitcnt(lgkmcnt=0), isa.vop("v_readfirstlane_b32", S_WAVE, T_[0]),
              isa.sop("s_lshr_b32", S_F, S_WAVE, I32(1)), isa.sop("s_and_b32", S_RH, S_WAVE, I32(1))]
        # ---- workgroup id -> (frame pair, tile row, tile column, n tile); n fastest ----
        tt = ST[4]
        q1, q2, q3 = ST[5], ST[6], ST[7]
        o += [isa.sop("s_lshl_b32", tt, S_WG, I32(1)), isa.sop("s_mul_hi_u32", q1, tt, S_MGN),                 # q1 = wid / tiles_n
              isa.sop("s_mul_i32", tt, q1, S_TLN), isa.sop("s_sub_u32", tt, S_WG, tt), isa.sop("s_mul_i32", S_N0, tt, I32(96)),
              isa.sop("s_lshl_b32", tt, q1, I32(1)), isa.sop("s_mul_hi_u32", q2, tt, S_MGW),                   # q2 = q1 / tiles_w
              isa.sop("s_mul_i32", tt, q2, S_TLW), isa.sop("s_sub_u32", tt, q1, tt), isa.sop("s_lshl_b32", S_W0, tt, I32(4)),
              isa.sop("s_lshl_b32", tt, q2, I32(1)), isa.sop("s_mul_hi_u32", q3, tt, S_MGH),                   # q3 = q2 / tiles_h
              isa.sop("s_mul_i32", tt, q3, S_TLH), isa.sop("s_sub_u32", tt, q2, tt), isa.sop("s_lshl_b32", S_H0, tt, I32(4)),
              isa.sop("s_lshl_b32", S_T0, q3, I32(1))]
        # ---- input frame bytes (64 bit) = H * W * Cin * 2 ----
        o += [isa.sop("s_mul_i32", ST[8], S_H, S_Wd), isa.sop("s_lshl_b32", S_CIN2, S_CIN, I32(1)),
              isa.sop("s_mul_i32", S_FB.sub(0), ST[8], S_CIN2), isa.sop("s_mul_hi_u32", S_FB.sub(1), ST[8], S_CIN2)]
        # ---- this wave's patch frame: input frame t = t0 - pt + wave; descriptor base = x + t * FB, num_records = FB (0 when t is outside [0, Ti)) ----
        tfr = ST[9]
        o += [isa.sop("s_add_u32", tfr, S_T0, S_WAVE), isa.sop("s_sub_u32", tfr, tfr, S_PT),                  # may wrap below 0 -> huge unsigned
              isa.sop("s_cmp_lt_u32", None, tfr, S_TI), isa.sop("s_cselect_b32", ST[10], S_FB.sub(0), I32(0)),   # num_records
              isa.sop("s_cselect_b32", tfr, tfr, I32(0))]
        o += [isa.sop("s_mul_i32", ST[0], S_FB.sub(0), tfr), isa.sop("s_mul_hi_u32", ST[1], S_FB.sub(0), tfr),
              isa.sop("s_mul_i32", ST[2], S_FB.sub(1), tfr), isa.sop("s_add_u32", ST[1], ST[1], ST[2]),
              isa.sop("s_add_u32", S_XRW.sub(0), S_X.sub(0), ST[0]), isa.sop("s_addc_u32", ST[1], S_X.sub(1), ST[1]),
              isa.sop("s_and_b32", S_XRW.sub(1), ST[1], I32(0xFFFF)), isa.sop("s_mov_b32", S_XRW.sub(2), ST[10]),
              isa.sop("s_mov_b32", S_XRW.sub(3), I32(0x00020000))]
        # ---- W descriptor: base = w + n0 * Kpad * 2, num_records = min(96, N - n0) * Kpad * 2 ----
        kp2 = ST[11]
        o += [isa.sop("s_lshl_b32", kp2, S_KPAD, I32(1)),
              isa.sop("s_mul_i32", ST[0], S_N0, kp2), isa.sop("s_mul_hi_u32", ST[1], S_N0, kp2),
              isa.sop("s_add_u32", S_WR.sub(0), S_Wp.sub(0), ST[0]), isa.sop("s_addc_u32", ST[1], S_Wp.sub(1), ST[1]),
              isa.sop("s_and_b32", S_WR.sub(1), ST[1], I32(0xFFFF)),
              isa.sop("s_sub_u32", ST[2], S_N, S_N0), isa.sop("s_min_u32", ST[2], ST[2], I32(96)), isa.sop("s_mul_i32", S_WR.sub(2), ST[2], kp2),
              isa.sop("s_mov_b32", S_WR.sub(3), I32(0x00020000))]
        # ---- LDS offsets of this wave's DMA regions ----
        o += [isa.sop("s_mul_i32", S_PLDS, S_WAVE, I32(FVOX * 16)),
              isa.sop("s_lshl_b32", S_WLDS, S_WAVE, I32(10)), isa.sop("s_add_u32", S_WLDS, S_WLDS, I32(PATCH))]
        t = T_
        # ---- patch voxel groups: lane l of group k = patch voxel pv = 64 k + l = (r, col); source offset inside the frame, or OOB ----
        hm1, wm1 = ST[12], ST[13]
        o += [isa.sop("s_sub_u32", hm1, S_H0, I32(1)), isa.sop("s_sub_u32", wm1, S_W0, I32(1))]
        for k in range(6):
            pv, r, col, hi, wi, ok = t[1], t[2], t[3], t[4], t[5], t[6]
            o += [isa.vop("v_add_u32", pv, I32(64 * k), LANE),
                  isa.vop("v_mul_u32_u24", r, I32(3641), pv), isa.vop("v_lshrrev_b32", r, I32(16), r),             # r = pv / 18 (pv < 384)
                  isa.vop("v_mul_u32_u24", col, I32(18), r), isa.vop("v_sub_u32", col, pv, col),
                  isa.vop("v_add_u32", hi, hm1, r), isa.vop("v_add_u32", wi, wm1, col),                             # wraps below 0 -> huge unsigned
                  isa.v_cmp("v_cmp_gt_u32", S_H, hi), isa.vop("v_mul_lo_u32", t[7], hi, S_Wd),
                  isa.v_cndmask(ok, I32(0), I32(1)),
                  isa.v_cmp("v_cmp_gt_u32", S_Wd, wi), isa.v_cndmask(t[8], I32(0), I32(1)), isa.vop("v_and_b32", ok, ok, t[8]),
                  isa.v_cmp("v_cmp_gt_u32", I32(PR * PC), pv), isa.v_cndmask(t[8], I32(0), I32(1)), isa.vop("v_and_b32", ok, ok, t[8]),
                  isa.vop("v_add_u32", t[7], t[7], wi), isa.vop("v_mul_lo_u32", t[7], t[7], S_CIN2),
                  isa.v_cmp("v_cmp_ne_u32", I32(0), ok), isa.vop("v_mov_b32", t[9], I32(OOB)),
                  isa.v_cndmask(PDMA[k], t[9], t[7])]
        # ---- W pieces of this wave: piece w + 4 i = rows 16 (w + 4 i) + l / 4, LDS position q = l % 4 holds source chunk q ^ ((row >> 1) & 3) ----
        o += [isa.vop("v_lshrrev_b32", t[1], I32(2), LANE), isa.vop("v_and_b32", t[2], I32(3), LANE)]
        for i in range(2):
            o += [isa.sop("s_add_u32", ST[0], S_WAVE, I32(4 * i)), isa.sop("s_lshl_b32", ST[0], ST[0], I32(4)),
                  isa.vop("v_add_u32", t[3], ST[0], t[1]),                                                        # row
                  isa.vop("v_lshrrev_b32", t[4], I32(1), t[3]), isa.vop("v_and_b32", t[4], I32(3), t[4]), isa.vop("v_xor_b32", t[4], t[2], t[4]),
                  isa.vop("v_mul_lo_u32", t[5], t[3], kp2), isa.vop("v_lshl_add_u32", WDMA[i], t[4], I32(4), t[5])]
        # ---- fragment bases ----
        ql, g = t[10], t[11]
        o += [isa.vop("v_and_b32", ql, I32(15), LANE), isa.vop("v_lshrrev_b32", g, I32(4), LANE)]
        # patch: chunk plane g, voxel (frame f, row 8 rh, column ql)
        o += [isa.sop("s_mul_i32", ST[0], S_F, I32(FVOX)), isa.sop("s_mul_i32", ST[1], S_RH, I32(8 * PC)), isa.sop("s_add_u32", ST[0], ST[0], ST[1]),
              isa.vop("v_add_u32", t[1], ST[0], ql), isa.vop("v_lshlrev_b32", t[1], I32(4), t[1]),
              isa.vop("v_mul_u32_u24", t[2], I32(PLANE), g), isa.vop("v_add_u32", PB, t[1], t[2])]
        # W: row ql (64 B), chunk g ^ ((ql >> 1) & 3)
        o += [isa.vop("v_lshrrev_b32", t[1], I32(1), ql), isa.vop("v_and_b32", t[1], I32(3), t[1]), isa.vop("v_xor_b32", t[1], g, t[1]),
              isa.vop("v_lshlrev_b32", t[2], I32(6), ql), isa.vop("v_lshl_add_u32", WB, t[1], I32(4), t[2]),
              isa.vop("v_add_u32", WB, I32(PATCH), WB)]
        for i in range(192):
            o.append(isa.vop("v_accvgpr_write_b32", A(i), I32(0)))
        # ---- streams: W taps 0, 1, 2 of slice 0 ----
        o += [isa.sop("s_mov_b32", S_SL, I32(0)), isa.sop("s_mov_b32", S_XOFF, I32(0)), isa.sop("s_mov_b32", S_WNEXT, I32(0))]
        for b in range(3):
            o += self.w_dma(b, 0, 0) + [isa.sop("s_add_u32", S_WNEXT, S_WNEXT, S_CIN2)]
        return sched.pad_hazards(sched.insert_lgkm_waits(o))

    def slice_body(self) -> List[Instr]:
        """One 32-channel slice: patch in, 27 taps."""
        o: List[Instr] = [isa.label("L_slice"), isa.nop(7)]
        pre: List[Instr] = [isa.barrier()]                         # every wave is done with the previous slice's patch
        pre += self.patch_dma()
        pre += [isa.waitcnt(vmcnt=0), isa.barrier()]
        pre += self.frag_reads(0, 0, 0, 0)
        o += sched.pad_hazards(pre)
        for tap in range(27):
            o += self.tap_block(tap)
        o += [isa.waitcnt(lgkmcnt=0),
              isa.sop("s_add_u32", S_SL, S_SL, I32(1)), isa.sop("s_add_u32", S_XOFF, S_XOFF, I32(64)),
              isa.sop("s_cmp_lt_u32", None, S_SL, S_NSL), isa.branch("s_cbranch_scc1", "L_slice")]
        return o

    # ---- epilogue ----------------------------------------------------------------------------------------------------------------
    def epilogue(self) -> List[Instr]:
        """lane: voxel (frame t0 + f, row h0 + 8 rh + mb, column w0 + l % 16), channels n0 + 16 nb + 4 (l / 16) + e;
        y / resid rows: voxel index ((frame * ot_mul + ot_off) * H + row) * W + column, strides ldc / ldr elements."""
        c = self.cfg
        t = T_
        e: List[Instr] = [isa.waitcnt(vmcnt=0), isa.nop(15), isa.nop(15)]
        ql, g = t[10], t[11]
        tf, tfo, hw = ST[4], ST[5], ST[6]
        ldc2, ldr2 = ST[13], ST[14]
        e += [isa.sop("s_add_u32", tf, S_T0, S_F),                                                               # output frame
              isa.sop("s_mul_i32", tfo, tf, S_OTM), isa.sop("s_add_u32", tfo, tfo, S_OTO),                       # its frame slot in y / resid
              isa.sop("s_mul_i32", hw, S_H, S_Wd),
              isa.sop("s_lshl_b32", ldc2, S_LDC.sub(0), I32(1)), isa.sop("s_lshl_b32", ldr2, S_LDR.sub(0), I32(1))]
        # frame bases (64 bit): ptr + slot * H * W * ld * 2 + n0 * 2
        for base, src, ld2 in ((S_YF, S_Y, ldc2), (S_RF, S_RES, ldr2)):
            e += [isa.sop("s_mul_i32", ST[0], hw, ld2), isa.sop("s_mul_hi_u32", ST[1], hw, ld2),                 # frame bytes (64 bit)
                  isa.sop("s_mul_i32", ST[2], ST[0], tfo), isa.sop("s_mul_hi_u32", ST[3], ST[0], tfo), isa.sop("s_mul_i32", ST[7], ST[1], tfo),
                  isa.sop("s_add_u32", ST[3], ST[3], ST[7]),
                  isa.sop("s_add_u32", base.sub(0), src.sub(0), ST[2]), isa.sop("s_addc_u32", base.sub(1), src.sub(1), ST[3]),
                  isa.sop("s_lshl_b32", ST[7], S_N0, I32(1)),
                  isa.sop("s_add_u32", base.sub(0), base.sub(0), ST[7]), isa.sop("s_addc_u32", base.sub(1), base.sub(1), I32(0))]
        # bias quads (zeros when bias == NULL)
        BQ = [V(120 + 4 * nb, 4) for nb in range(6)]
        for i in range(24):
            e.append(isa.vop("v_mov_b32", V(120 + i), I32(0)))
        e += [isa.sop("s_cmp_eq_u64", None, S_BIAS, I32(0)), isa.branch("s_cbranch_scc1", "L_nobias"),
              isa.sop("s_lshl_b32", ST[7], S_N0, I32(2)), isa.sop("s_add_u32", S_BIAS.sub(0), S_BIAS.sub(0), ST[7]),
              isa.sop("s_addc_u32", S_BIAS.sub(1), S_BIAS.sub(1), I32(0)), isa.vop("v_lshlrev_b32", t[1], I32(4), g)]
        for nb in range(6):
            e.append(isa.global_load(4, BQ[nb], t[1], 64 * nb, saddr=S_BIAS))
        e += [isa.waitcnt(vmcnt=0), isa.label("L_nobias"), isa.nop(7)]
        # column and frame validity are the same for all 8 row blocks
        wcol = t[2]
        e += [isa.vop("v_add_u32", wcol, S_W0, ql),
              isa.sop("s_cmp_lt_u32", None, tf, S_T), isa.sop("s_cselect_b32", ST[8], S_Wd, I32(0))]      # frame outside [0, To): no column is valid
        row0 = ST[9]
        e += [isa.sop("s_lshl_b32", row0, S_RH, I32(3)), isa.sop("s_add_u32", row0, row0, S_H0)]
        RP = [V(144 + 2 * nb, 2) for nb in range(6)]
        for mb in range(8):
            yoff, roff, vox, hrow = t[4], t[5], t[6], ST[10]
            e += [isa.sop("s_add_u32", hrow, row0, I32(mb)),
                  isa.sop("s_cmp_lt_u32", None, hrow, S_H), isa.sop("s_cselect_b32", ST[11], ST[8], I32(0)),      # columns allowed in this row
                  isa.sop("s_mul_i32", ST[12], hrow, S_Wd),
                  isa.vop("v_add_u32", vox, ST[12], wcol),
                  isa.vop("v_mul_lo_u32", yoff, vox, ldc2), isa.vop("v_lshl_add_u32", yoff, g, I32(3), yoff),
                  isa.v_cmp("v_cmp_gt_u32", ST[11], wcol),
                  Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)]
            if c.epi == 3:
                e += [isa.vop("v_mul_lo_u32", roff, vox, ldr2), isa.vop("v_lshl_add_u32", roff, g, I32(3), roff)]
                for nb in range(6):
                    e.append(isa.global_load(2, RP[nb], roff, 32 * nb, saddr=S_RF, extra_reads=[EXEC]))
                e.append(isa.waitcnt(vmcnt=0))
            for nb in range(6):
                base = 160 + 8 * (nb % 2)
                f = [V(base + i) for i in range(4)]
                w, r_ = V(base + 4, 2), V(base + 6)
                acc = ACC(nb, mb)
                for i in range(4):
                    e += [isa.vop("v_accvgpr_read_b32", f[i], acc.sub(i)), isa.vop("v_add_f32", f[i], f[i], BQ[nb].sub(i))]
                if c.epi == 3:
                    for i in range(4):
                        src = RP[nb].sub(i >> 1)
                        e += [isa.vop("v_lshlrev_b32", r_, I32(16), src) if (i & 1) == 0 else isa.vop("v_and_b32", r_, I32(0xFFFF0000), src),
                              isa.vop("v_add_f32", f[i], f[i], r_)]
                e += [isa.vop("v_cvt_pk_bf16_f32", w.sub(0), f[0], f[1]), isa.vop("v_cvt_pk_bf16_f32", w.sub(1), f[2], f[3]),
                      isa.global_store(2, yoff, w, 32 * nb, saddr=S_YF, extra_reads=[EXEC])]
            e += [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]
        e += [isa.waitcnt(vmcnt=0), Instr("s_endpgm", cls=isa.BRANCH)]
        return sched.pad_hazards(e)

    def program(self) -> List[Instr]:
        prog = self.prologue() + self.slice_body() + self.epilogue()
        pre = f"L_{self.cfg.name}"
        for i in prog:
            if i.label and i.label.startswith("L_"):
                new = pre + i.label[1:]
                if getattr(i, "text", None):
                    i.text = i.text.replace(i.label, new)
                i.label = new
        return prog


HEAD = """// GENERATED by scail_amd/asmgen/conv4.py -- do not edit; regenerate with `python -m scail_amd.asmgen.conv4`.
// Hand-scheduled 3x3x3 causal convolution for gfx950 (2 frames x 16 x 16 voxels x 96 channels per workgroup); see the generator.
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
\t\t.amdhsa_group_segment_fixed_size {LDS_BYTES}
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
    .group_segment_fixed_size: {LDS_BYTES}
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


DEFAULTS = [Cfg(epi=0, name="scail_conv4_e0"), Cfg(epi=3, name="scail_conv4_e3")]


def variant_cfgs():
    out = []
    for abl in ("dma", "lds", "bar", "dma,lds"):
        out.append(Cfg(epi=0, abl=abl, name="scail_conv4_e0_abl_" + abl.replace(",", "_")))
    out.append(Cfg(epi=0, cap=2, name="scail_conv4_e0_c2"))
    out.append(Cfg(epi=0, rd_step=2.0, name="scail_conv4_e0_rd2"))
    out.append(Cfg(epi=0, rd_at=6.0, rd_step=2.5, name="scail_conv4_e0_rd6"))
    return out


def main():
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "csrc", "conv4.s")
    text = assembly(DEFAULTS)
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(out) and open(out).read() == text else 1)
    if not os.path.exists(out) or open(out).read() != text:
        open(out, "w").write(text)
    print(out, len(text.splitlines()), "lines")


if __name__ == "__main__":
    main()
