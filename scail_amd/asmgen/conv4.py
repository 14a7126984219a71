"""conv4 -- hand-scheduled 3x3x3 causal convolution for the Wan2.1 VAE on gfx950 (generator of csrc/conv4.s).

Reference: CausalConv3d (sgm/models/wan_vae.py:17-36: F.pad(2 frames in front, 1 voxel around) + nn.Conv3d(k = 3, stride 1)) as used by
every ResidualBlock (wan_vae.py:162-205) -- > 90 % of the VAE's FLOPs.  Same C entry point as the kernels of csrc/conv.hip
(scail_conv3d_cl), which runs these kernels where scail_conv3d_kernel_for says 4: kt = kh = kw = 3, stride 1, 'same' spatial extent, 0..2
padding frames in front, Cin % 32 == 0, N % 96 == 0, at least two output frames.  Epilogues: e0  y = conv + bias;  e3  y = resid + conv + bias.

Shape (the gemm4 / attn4 structure: 4 waves, one per SIMD, accumulators in the AGPR file, LDS-DMA staging, MFMA-spine scheduling):
  * channels-last activations x (Ti, H, W, Cin), y (frames, H, W, ldc) bf16; weights (N, Kpad) bf16, k = ((dt 3 + dh) 3 + dw) Cin + c
    (scail_amd.ops.prep_conv_weight); output frame t sees input frames t - pt .. t - pt + 2, rows / columns -1 .. +1.
  * TILE = 2 output frames x 16 x 16 voxels x 96 output channels.  Wave w: frame w >> 1, rows 8 (w & 1) .. + 7 -> 8 voxel blocks of 16 (one
    patch row each) x 6 channel blocks of 16 = 48 v_mfma_f32_16x16x32_bf16 per (tap, 32-channel slice); 192 accumulators per lane in a[0:191].
  * PERSISTENT workgroups, one per compute unit: workgroup b runs on XCD b % 8 and walks a stride of that XCD's contiguous tile range (n tiles
    of a patch and neighbouring patches share an L2: 1.4-1.8 x the algorithmic traffic where the one-tile-per-workgroup hipcc kernel moves
    2.6-5 x).  The next tile's setup, first DMAs and bias loads are issued BEFORE the finished tile's epilogue, which hides their latency.
  * the input PATCH of a 32-channel slice -- 4 frames x 18 x 18 voxels -- lives in LDS voxel-major: a voxel's slice is 64 contiguous bytes,
    rows of 20 voxels, the 16-byte chunk q of voxel (r, col) holds source chunk q ^ ((col >> 1) & 3): fragment reads (lane = voxel l % 16 + dw,
    chunk l / 16) are conflict-free, a tap row is an immediate offset, and an LDS-DMA instruction moves 16 voxels x 64 contiguous bytes (16
    cache lines; a chunk-planar layout with 64 lines per instruction cost 6.7-9 k cycles of VMEM issue for a tile's first loads against 4-4.7 k).
    Spatial / causal padding = the buffer descriptors' range check (out-of-range lanes and whole padding frames read zeros).
  * FRAME-SLOT RING: 5 slots for the 4 frames of a slice.  Taps run dt-major, so frame 0 of a slice is released after a third of the taps,
    frame 1 after two thirds: the next slice's frames are requested into the slots as they free up, 9-18 taps before their first use -- the
    patch load of a slice never stalls the MFMAs (a load-then-compute version: 1023 / 1170 / 1214 TF/s; the ring: 1277 / 1432 / 1514).
  * the W tile of one tap (96 rows x 64 B) arrives by LDS-DMA four taps ahead into a ring of four 8 KB buffers (rows unpadded, 16-byte chunk
    XOR-ed with (row >> 1) & 3 on the source address: conflict-free A-fragment reads); one s_barrier per tap.
  * TAP ORDER (dt, dw, dh) with dh innermost: the 10 patch rows of a group (dt, dw) are read once into registers and serve 3 taps (9.3 instead
    of 14 ds_read_b128 per tap); fragment sets rotate modulo 3 (27 taps and 9 groups per slice: the sets line up across slices).
  * per tap: 48 MFMAs || ~10 fragment reads of the next tap / group || 2 W-tile DMA pieces || 1 patch piece; 27 taps unrolled, a loop over
    the slices; counted s_waitcnt vmcnt(n) at the top of every tap, computed from the DMA issue order and each DMA's deadline.
  * EPILOGUE through LDS: a row block (16 voxels x 96 channels) is written in accumulator layout and read back in memory layout, so every global
    store (and residual load) instruction moves 64 x 16 contiguous bytes instead of 64 x 8 scattered ones (48 partial-line stores cost 7-10 k
    cycles per tile, 24 whole-line ones 2.6 k); the stores are non-temporal and drain behind the next tile's taps.  The bias is the
    accumulators' initial value.
Measured (profiles/r03_conv4_*.log, 21 x 512 x 896 x 96 / 256 x 448 x 192 / 128 x 224 x 384): 1350 / 1500 / 1560 TF/s against 960 / 1070 /
1100 for the hipcc halo kernel; s_memtime phase timers: 800 cycles per tap of the 768 the MFMAs need, ~12 k cycles between tiles, shader clock
1.4-1.7 GHz under this load (the "_prof" variant of the measurement build).
Negative results kept as notes: splitting the DMA work by kind (waves 0, 1 the W tiles, waves 2, 3 the patch) -1..2 %; staggered workgroup
starts neutral (the per-tile cost is per-CU VMEM issue, not an HBM burst).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import List

from . import isa, sched
from .isa import A, S, V, I32, F32, VCC, EXEC, M0, Instr

KERNARG_SIZE = 152
# x w bias y resid | Ti To H W | Cin N Kpad pt | tiles_t tiles_w tiles_n magic_n | magic_w magic_t n_slices ot_mul | ot_off tiles_per_wg | ldc ldr | wgs_per_xcd tiles
# | gamma (epi 5 / 6: fp32 [96]) | y2 - y in bytes (epi 5: where the normalised copy goes, same row stride and frame mapping as y)
KERNARG_FMT = "<5Q4i4i3iI2I2iii2q2iQq"
KARG_GAMMA, KARG_Y2D = 136, 144

TH, TW, NF = 16, 16, 2
PR, PC = TH + 2, TW + 2                 # patch rows / columns
FVOX = 384                              # voxels reserved per patch frame slot (18 rows x 20 = 360 used; 24 DMA pieces of 16)
PCL = 20                                # voxels per patch row in LDS (18 used: a multiple of 4 keeps the chunk swizzle a function of the column)
ROWB = PCL * 64                         # bytes per patch row in LDS: a voxel's 32-channel slice is 64 contiguous bytes (4 chunks of 16)
FSLOT = FVOX * 64                       # one patch frame of a 32-channel slice: 24576 (18 x 20 voxels = 23040 used; 24 DMA pieces of 16 voxels)
NSLOT = 5                               # ring of frame slots: the 4 frames of a slice + 1 (frames of the next slice arrive as the taps release the old ones)
WBUF = 8192                             # bytes per W tap buffer (96 rows x 64 B used; 8 DMA pieces of 1 KB)
NWB = 4                                 # W tiles in flight: tap g + 4 is requested during tap g
WREG = NWB * WBUF                       # the W ring sits first (32768 = a power of two: the ring index wraps with one AND)
PBASE0 = WREG
LDS_BYTES = WREG + NSLOT * FSLOT        # 155648
OOB = 0x7FFFFF00                        # lane offset that fails every descriptor's range check (-> zeros)


def magic31(d: int) -> int:
    return -(-(1 << 31) // d)


def pack_args(x, w, bias, y, resid, Ti, To, H, W, Cin, N, Kpad, pt=2, ot_mul=1, ot_off=0, ldc=0, ldr=0, cus=256, gamma=0, y2=0) -> bytes:
    """ldc / ldr: output / residual row strides in elements (0 = N); cus: compute units (one persistent workgroup each); gamma / y2: the norm
    weights and the second output of the residual + norm epilogues (epi 5 / 6)."""
    tp, tw, tn = (To + NF - 1) // NF, (W + TW - 1) // TW, (N + 95) // 96
    tiles, grid = grid_tiles(To, H, W, N), grid_blocks(To, H, W, N, cus)
    b = struct.pack(KERNARG_FMT, x, w, bias, y, resid, Ti, To, H, W, Cin, N, Kpad, pt, tp, tw, tn, magic31(tn), magic31(tw), magic31(tp),
                    Cin // 32, ot_mul, ot_off, tiles // grid if tn == 1 else 0, ldc or N, ldr or N, grid // 8, tiles, gamma, (y2 - y) if y2 else 0)
    assert len(b) == KERNARG_SIZE, len(b)
    return b


def grid_tiles(T, H, W, N) -> int:
    return ((T + NF - 1) // NF) * ((H + TH - 1) // TH) * ((W + TW - 1) // TW) * ((N + 95) // 96)


def grid_blocks(T, H, W, N, cus=256) -> int:
    """workgroups launched: persistent, at most one per compute unit, a multiple of 8.  Tiles are numbered with the N TILE fastest, then the FRAME
    PAIR (then the tile column, the tile row); workgroup b (XCD b % 8) is number w = (b % 8) * (grid / 8) + b / 8 of the grid.
    One n tile (N = 96): w takes the next tiles // grid (+ 1 for the first tiles % grid workgroups) tiles of that order -- it walks the frame
    pairs of a spatial tile (the two input frames consecutive pairs share come back from L2 / the Infinity Cache, the per-lane patch offsets
    stay), the workgroups of an XCD work on neighbouring spatial tiles.  Several n tiles: w takes tiles w, w + grid, ...: an XCD works on
    grid / 8 consecutive tiles at a time = the n tiles of a few neighbouring frame pairs of one spatial tile, which share their patch through
    the XCD's L2 (a run per workgroup would fetch the patch once per n tile: 5.7 / 11 x the algorithmic traffic at 192 / 384 channels)."""
    per = (grid_tiles(T, H, W, N) + 7) // 8
    return 8 * min(per, max(cus // 8, 1))


@dataclass
class Cfg:
    epi: int = 0            # 0: y = conv + bias;  3: y = resid + conv + bias;  4: y = SiLU(RMS_norm(bf16(conv + bias)) * gamma) -- the first
                            # convolution of a ResidualBlock with its RMS_norm and SiLU (wan_vae.py:190-196); one n tile (N = 96), gamma (fp32 [96])
                            # arrives in the `resid` argument
                            # 6: y = SiLU(RMS_norm(bf16(resid + conv + bias)) * gamma) -- the LAST convolution of a ResidualBlock with its shortcut add and
                            # the norm + SiLU of whatever reads the block's output next, when nothing else reads it (the decoder head, wan_vae.py:417);
                            # 5: y = bf16(resid + conv + bias) AND y2 = SiLU(RMS_norm(y) * gamma) -- the same when the raw sum is still needed (the next
                            # ResidualBlock's shortcut): the block-input rms_silu pass disappears (round 6).  One n tile; gamma: kernel argument 136,
                            # y2 - y: argument 144
                            # 7: y = conv + bias AND y2 = SiLU(RMS_norm(bf16(y)) * gamma), no residual -- epi 5 for a producer that is not a
                            # ResidualBlock: Resample's 1x3x3 convolution (kt = 1) in front of the full-resolution blocks of the decoder
    kt: int = 3             # temporal taps: 3 = CausalConv3d 3x3x3;  1 = the 1x3x3 convolution of Resample (behind the nearest 2x upsample when the
                            # kernel argument `pt` -- no padding frames exist for kt = 1 -- is 1: patch voxel (h, w) reads input (h >> 1, w >> 1))
    cont: bool = False      # TILE CONTINUATION (one n tile, kt = 3): when the next tile of a workgroup's run is the next frame pair of the same spatial tile,
                            # the last slice's "next slice" prefetch -- otherwise a wasted re-read of slice 0 -- fetches the NEXT TILE's frames 0, 1, 2 of
                            # slice 0 (= this tile's frames 2, 3 and one new frame) and its W taps 0..3; the rings just continue, the next tile issues no
                            # first loads.  Costs a dynamic staging slot for the epilogue (the ring phase differs from tile to tile)
    nb: int = 6             # 16-channel output blocks of the tile: 6 = 96 channels;  1 = a NARROW output (N <= 16: the decoder's RGB head, 96 -> 3):
                            # 8 MFMAs per (tap, slice) instead of 48, the W rows past N read zeros, 8-byte stores straight from the accumulator layout
    cap: int = 1
    lookahead: float = 2.0
    name: str = "scail_conv4_e0"
    rd_at: float = 1.0      # first gap of the next tap's 14 fragment reads, rd_step apart
    rd_step: float = 3.0
    dma_at: float = 4.0     # gap of the first W-tile DMA piece of tap + 3, the second dma_step later
    dma_step: float = 20.0
    p_at: float = 12.0      # gap of the tap's patch DMA piece
    stagger: int = 0        # (A/B) workgroup i of an XCD starts i * 64 * stagger cycles late; measured neutral to -1 %: the per-tile cost is VMEM issue, not an HBM burst
    nt: bool = True         # non-temporal output stores: the output does not displace the patch lines whose second 64-byte half the next slice wants (-1..3 %)
    prof: bool = False      # measurement variant: s_memtime stamps around the phases of a tile; wave 0 of every workgroup writes the sums to `resid`
    abl: str = ""


# ---- registers ------------------------------------------------------------------------------------------------------------------
def ACC(nb, mb): return A((nb * 8 + mb) * 4, 4)
def WF(s, nb): return V(s * 24 + nb * 4, 4)                 # W fragments (A operand), set s = tap % 3 (27 taps: the sets line up across slices)


def XF(s, r):
    """x fragments (B operand): patch row r (0..9) of tap group (dt, dw), set s = group % 3: the 10 rows serve the 3 taps dh = 0, 1, 2."""
    if s == 0:
        return A(192 + 4 * r, 4)
    if s == 1:
        return A(232 + 4 * r, 4) if r < 6 else V(72 + 4 * (r - 6), 4)
    return V(88 + 4 * r, 4)


PBASE = [V(128 + dt) for dt in range(3)]                    # per-lane fragment base inside the slot of patch frame f + dt of the current slice
WB = V(131)                                                 # per-lane W fragment base inside the W ring (advances one buffer per tap)
PDMA = [V(132 + k) for k in range(6)]                       # per-lane source offsets of this wave's 6 pieces (16 voxels x 4 chunks each) of a patch frame
WDMA = [V(138 + i) for i in range(2)]                       # per-lane source offsets of this wave's 2 W-tile pieces
LANE = V(141)
T_ = [V(144 + i) for i in range(40)]                        # v144..v183 temporaries
EPI_BQ, EPI_RP, EPI_F = 184, 208, 220                       # epilogue: bias quads v184..207, residual pairs v208..219, staging v220..235
EPI_GQ = 228                                                # (epi 4 / 5 / 6) gamma quads v228..251 of this lane's channels 16 nb + 4 (l / 16) + e, loaded once at entry
EPI_Y2D = 225                                               # (epi 5) v225, v226: y2 - y in bytes (64 bit), kept in a VGPR pair: the SGPR file is full and the kernarg pointer is gone after entry

S_KARG = S(0, 2)
S_WG = S(2)
S_TILE, S_G, S_TEND = S(3), S(4), S(5)                      # this workgroup's current tile, workgroups per XCD (entry only), end of its tile range
S_STEP = S(4)                                               # ... then the tile step (1: a run of consecutive tiles; the grid: strided)
S_ET0, S_EH0, S_EW0, S_EN0 = S(6), S(7), S(18), S(19)       # coordinates of the tile whose accumulators wait for the epilogue
S_X, S_Wp, S_BIAS, S_Y, S_RES = S(8, 2), S(10, 2), S(12, 2), S(14, 2), S(16, 2)
S_TI, S_T, S_H, S_Wd = S(20), S(21), S(22), S(23)            # input frames, output frames, rows, columns
S_CIN, S_N, S_KPAD, S_PT = S(24), S(25), S(26), S(27)
S_TLT, S_TLW, S_TLN, S_MGN = S(28), S(29), S(30), S(31)     # frame pairs, tile columns, n tiles; magic numbers of the divisions
S_MGW, S_MGT, S_NSL, S_OTM = S(32), S(33), S(34), S(35)
S_OTO, S_PER = S(36), S(37)
S_KP2 = S(37)                                               # 2 Kpad (takes the place of tiles_per_wg once the tile range is known)
S_SAVE = S(38, 2)                                           # epilogue: saved exec
S_LDC, S_LDR = S(40, 2), S(42, 2)
S_XR = [S(44 + 4 * j, 4) for j in range(4)]                 # buffer descriptors of the 4 input frames of the patch (num_records = 0: a padding frame)
S_WR = S(60, 4)                                             # W descriptor of this workgroup's 96 rows
S_WAVE, S_F, S_RH = S(64), S(65), S(66)
S_T0, S_H0, S_W0, S_N0 = S(67), S(68), S(69), S(70)
S_SL, S_XOFF, S_XOFFN = S(71), S(72), S(73)                 # slice counter, channel byte offset of this / the next slice
S_WNEXT, S_CIN2, S_C26 = S(74), S(75), S(76)                # W source offset of the next DMA tap, 2 Cin, (taps per slice - 1) * 2 Cin - 64
S_WI = S(81)                                                # (kt = 1) columns of an input frame: W >> ups (S_SLOT[3] is unused there)
S_WM0 = S(77)                                               # LDS offset of this wave's first piece in the W buffer of the current tap
S_SLOT = [S(78 + j) for j in range(4)]                      # LDS offset (plane of this wave) of the slot the next load of patch frame j goes to
ST = [S(82 + i) for i in range(16)]                         # s82..s97 temporaries
S_C3, S_C5 = S(98), S(99)                                   # 3 * 2 Cin, 5 * 2 Cin
S_FB = S(100, 2)                                            # bytes of an input frame (64 bit)
S_YF, S_RF = S(0, 2), S(96, 2)                              # epilogue: output / residual frame base (the kernarg pointer and two temporaries are free then)
S_LDR2 = S(2)                                               # epilogue: residual row stride in bytes (the workgroup id is consumed at entry)
N_SGPR = 102
S_PROFWG = S(43)                                            # (prof variant; the unused high word of ldr) workgroup id
S_CONT = S(43)                                              # (cont variants; same register) bit 0: the tile after this one continues the rings; bit 1: this tile's first loads
                                                            # were prefetched; bit 2: the running slice is the last one of a tile whose successor continues the rings
PBL, WBL = V(142), V(143)                                   # tile-independent parts of the fragment bases
LP = [V(221 + dw) for dw in range(3)]                       # the lane's place in a patch row shifted by dw columns: (u 64 + ((chunk ^ ((u >> 1) & 3)) 16), u = l % 16 + dw
XB = V(224)                                                 # fragment base of the tap group being read
# epilogue staging (a row block of 16 voxels x 96 channels goes through LDS so that global stores / residual loads are whole contiguous lines):
STG_VOX = 208                                               # bytes per staged voxel (192 + 16: conflict-free 8-byte writes, 16-byte aligned chunks)
STG_WAVE = 16 * STG_VOX                                     # 3328 bytes per wave, in frame slot 3 (idle between a tile's last tap and the next tile's first)
E_W = V(208)                                                # LDS address of this lane's accumulator-layout piece: voxel l % 16, channels 4 (l / 16) + 16 nb
E_R = [V(209 + i) for i in range(3)]                        # LDS address of this lane's memory-layout chunk j = 64 i + l: voxel j / 12, 16-byte chunk j % 12
E_V = [V(212 + i) for i in range(3)]                        # its voxel j / 12
E_Y = [V(215 + i) for i in range(3)]                        # its byte offset in an output row: (j / 12) * ldc * 2 + (j % 12) * 16
E_Z = [V(218 + i) for i in range(3)]                        # ... in a residual row


class Gen:
    def __init__(self, cfg: Cfg):
        self.cfg = cfg
        assert cfg.kt in (1, 3)
        self.kt1 = cfg.kt == 1
        self.TAPS = 9 * cfg.kt                  # taps of a 32-channel slice
        self.NGRP = self.TAPS // 3              # tap groups (dt, dw) of a slice: the 10 patch rows of a group serve its 3 taps dh
        self.NFR = NF + cfg.kt - 1              # patch frames of a slice
        assert cfg.nb in (1, 6) and not (cfg.nb == 1 and (cfg.epi != 0 or cfg.kt != 3)) and cfg.epi in (0, 3, 4, 5, 6, 7)
        assert not (cfg.epi in (5, 6) and (cfg.kt != 3 or cfg.prof)) and not (cfg.epi == 7 and cfg.prof)
        self.NB = cfg.nb
        self.gs = cfg.nb / 6.0                  # the fillers' target gaps scale with the MFMAs of a tap (48 -> 8)
        assert not (cfg.cont and (cfg.kt != 3 or cfg.prof))
        # kt = 3: which patch piece a position issues: (frame, voxel group, next slice?, needed at the top of relative position).  Frame 2 is first
        # read (wave frame 1, dt = 1: group 3) during positions 6..8, frame 3 (group 6) during 15..17, frames 0 / 1 of the next slice during
        # 24..26; their slots were released by frames 1 (after the top of 15), 2 (24), 3 (24) of the slice before and 0 (6) of this one.
        # kt = 1: two frames per slice in two alternating slot pairs.  Group 0 of slice s + 1 is read during positions 6..8 of slice s, so both
        # frames of the next slice are requested during positions 0..2 (four pieces each) into the pair slice s - 1 released at the top of its
        # position 6, and are due at the top of position 6.
        if self.kt1:
            self.PIECES = {t: [(j, 2 * t + h, True, 6) for j in range(2) for h in range(2)] for t in range(3)}
        else:
            self.PIECES = {**{i: [(3, i, False, 15)] for i in range(6)}, **{6 + i: [(0, i, True, 24)] for i in range(6)},
                           **{12 + i: [(1, i, True, 24)] for i in range(6)}, **{18 + i: [(2, i, True, 27 + 6)] for i in range(6)}}

    # ---- building blocks -------------------------------------------------------------------------------------------------------
    N_PHASE = 8
    PROF_PREV, PROF_T = V(236), V(237)

    def stamp(self, phase: int) -> List[Instr]:
        """(prof variant) cycles since the previous stamp are added to phase ``phase`` (v240 + phase); phase < 0: just restart the clock."""
        if not self.cfg.prof:
            return []
        tm = S(ST[0].idx, 2)
        i = Instr("s_memtime", [tm], [], cls=isa.SALU)
        i.text = f"s_memtime {tm}"
        o = [i, isa.waitcnt(lgkmcnt=0)]
        if phase >= 0:
            o += [isa.vop("v_sub_u32", self.PROF_T, tm.sub(0), self.PROF_PREV), isa.vop("v_add_u32", V(240 + phase), V(240 + phase), self.PROF_T)]
        o += [isa.vop("v_mov_b32", self.PROF_PREV, tm.sub(0))]
        return o

    def prof_init(self) -> List[Instr]:
        return [isa.vop("v_mov_b32", V(240 + i), I32(0)) for i in range(self.N_PHASE)] if self.cfg.prof else []

    def prof_dump(self) -> List[Instr]:
        """wave 0, lane 0: phase sums -> resid[workgroup][8] (uint32)."""
        if not self.cfg.prof:
            return []
        o = [isa.sop("s_cmp_eq_u32", None, S_WAVE, I32(0)), isa.branch("s_cbranch_scc0", "L_exit"),
             isa.sop("s_lshl_b32", ST[0], S_PROFWG, I32(5)), isa.vop("v_lshlrev_b32", T_[1], I32(2), LANE), isa.vop("v_add_u32", T_[1], ST[0], T_[1]),
             Instr("s_mov_b64", [EXEC], [I32(1)], cls=isa.SALU)]
        for i in range(self.N_PHASE):
            o.append(isa.global_store(1, T_[1], V(240 + i), 4 * i, saddr=S_RES, extra_reads=[EXEC]))
        return o

    # Tap order: position i = (dt 3 + dw) 3 + dh -- dh innermost: the 10 patch rows of a group (dt, dw) are read once and serve 3 taps.
    @staticmethod
    def tap_id(i: int) -> int:
        """weight tap (dt 3 + dh) 3 + dw of position i (kt = 1: dt = 0, positions 0..8)."""
        dt, dw, dh = i // 9, (i // 3) % 3, i % 3
        return (dt * 3 + dh) * 3 + dw

    def mfmas(self, i: int) -> List[Instr]:
        # n-block major: the A fragment (W) stays for 8 consecutive MFMAs
        ws, xs, dh = i % 3, (i // 3) % 3, i % 3
        return [isa.mfma16(ACC(nb, mb), WF(ws, nb), XF(xs, mb + dh), ACC(nb, mb), tag="mm") for nb in range(self.NB) for mb in range(8)]

    def w_reads(self, i: int, t0: float, step: float) -> List[Instr]:
        """the 6 W quads of position i from the W buffer WB points at."""
        return [isa.ds_read_b128(WF(i % 3, nb), WB, nb * 1024, target_gap=t0 + step * nb) for nb in range(self.NB)]

    def x_reads(self, grp: int, rows, t0: float, step: float) -> List[Instr]:
        """patch rows ``rows`` of tap group grp = dt 3 + dw (of the slice the PBASE registers point into) -> set grp % 3."""
        dt, dw = grp // 3, grp % 3
        return [isa.ds_read_b128(XF(grp % 3, r), XB, r * ROWB, target_gap=t0 + step * k) for k, r in enumerate(rows)]

    def xb_set(self, grp: int, tg: float) -> List[Instr]:
        """fragment base of tap group grp = dt 3 + dw: slot of frame f + dt (+ this wave's first row) + the lane's place in a row shifted by dw."""
        return [isa.vop("v_add_u32", XB, PBASE[grp // 3], LP[grp % 3], target_gap=tg)]      # (kt = 1: groups 0..2 = dw, all in frame f)

    def w_dma(self, t0: float, step: float, need: int) -> List[Instr]:
        """this wave's 2 pieces (of 8: pieces w and w + 4; rows >= 96 fail the range check = zeros) of the W tile at S_WNEXT -> the buffer
        S_WM0 points at; then the ring moves on.  (Splitting the DMA work by kind instead -- waves 0, 1 the W tiles, waves 2, 3 the patch frames,
        so that a W tile never queues behind a patch piece from HBM in the in-order return queue -- measured 1-2 % SLOWER.)"""
        out = []
        for i in range(2):
            out.append(isa.sop("s_add_u32", M0, S_WM0, I32(4096 * i), target_gap=t0 + step * i - 0.5))
            d = isa.buffer_load_lds(WDMA[i], S_WR, S_WNEXT, 0, target_gap=t0 + step * i, tag="dma")
            d.need = need
            out.append(d)
        out += [isa.sop("s_add_u32", S_WM0, S_WM0, I32(WBUF), target_gap=t0 + step + 1.0),
                isa.sop("s_and_b32", S_WM0, S_WM0, I32(WREG - 1), target_gap=t0 + step + 1.2)]
        return out

    def w_next(self, pos: int, tg: float) -> List[Instr]:
        """advance S_WNEXT from the weight tap of position ``pos`` (0..26) to that of the next position."""
        if pos == self.TAPS - 1:  # position 0 of the following slice: back TAPS - 1 taps, forward one slice (64 bytes)
            return [isa.sop("s_sub_u32", S_WNEXT, S_WNEXT, S_C26, target_gap=tg)]
        d = self.tap_id(pos + 1) - self.tap_id(pos)
        return [isa.sop("s_add_u32", S_WNEXT, S_WNEXT, {3: S_C3, 1: S_CIN2}[d], target_gap=tg) if d > 0 else
                isa.sop("s_sub_u32", S_WNEXT, S_WNEXT, {-5: S_C5}[d], target_gap=tg)]

    def patch_piece(self, j: int, k: int, soff, tg: float, need: int) -> List[Instr]:
        """piece wave + 4 k (16 voxels x 64 bytes) of patch frame j -> the slot S_SLOT[j] points at."""
        d = isa.buffer_load_lds(PDMA[k], S_XR[j], soff, 0, target_gap=tg, tag="pdma")
        d.need = need
        return [isa.sop("s_add_u32", M0, S_SLOT[j], I32(k * 4096), target_gap=tg - 0.5), d]

    def slot_next(self, j: int, tg: float) -> List[Instr]:
        """the next load of frame j goes one slot down the ring (4 frames per slice, 5 slots: -1 mod 5); kt = 1: to the other slot pair."""
        t = ST[0]
        if self.kt1:
            return [isa.sop("s_add_u32", t, S_SLOT[j], I32(2 * FSLOT), target_gap=tg), isa.sop("s_cmp_ge_u32", None, t, I32(PBASE0 + 4 * FSLOT), target_gap=tg + 0.1),
                    isa.sop("s_cselect_b32", ST[1], I32(4 * FSLOT), I32(0), target_gap=tg + 0.2), isa.sop("s_sub_u32", S_SLOT[j], t, ST[1], target_gap=tg + 0.3)]
        return [isa.sop("s_sub_u32", t, S_SLOT[j], I32(FSLOT), target_gap=tg), isa.sop("s_cmp_lt_u32", None, t, I32(PBASE0), target_gap=tg + 0.1),
                isa.sop("s_cselect_b32", ST[1], I32(NSLOT * FSLOT), I32(0), target_gap=tg + 0.2), isa.sop("s_add_u32", S_SLOT[j], t, ST[1], target_gap=tg + 0.3)]

    def pbase_next(self, dt: int, tg: float) -> List[Instr]:
        t0, t1 = T_[0], T_[1]
        if self.kt1:
            return [isa.vop("v_add_u32", t0, I32(2 * FSLOT), PBASE[dt], target_gap=tg), isa.v_cmp("v_cmp_le_u32", I32(PBASE0 + 4 * FSLOT), t0, target_gap=tg + 0.1),
                    isa.vop("v_subrev_u32", t1, I32(4 * FSLOT), t0, target_gap=tg + 0.2), isa.v_cndmask(PBASE[dt], t0, t1, target_gap=tg + 0.3)]
        return [isa.vop("v_subrev_u32", t0, I32(FSLOT), PBASE[dt], target_gap=tg), isa.v_cmp("v_cmp_gt_u32", I32(PBASE0), t0, target_gap=tg + 0.1),
                isa.vop("v_add_u32", t1, I32(NSLOT * FSLOT), t0, target_gap=tg + 0.2), isa.v_cndmask(PBASE[dt], t0, t1, target_gap=tg + 0.3)]

    XSPLIT = ((0, 1, 2, 3), (4, 5, 6), (7, 8, 9))          # rows of the next group read during dh = 0, 1, 2 of this one

    def tap_fillers(self, i: int) -> List[Instr]:
        c = self.cfg
        g = self.gs
        abl = c.abl.split(",")
        blk: List[Instr] = []
        if "lds" not in abl:
            blk += [isa.vop("v_add_u32", WB, I32(WBUF), WB, target_gap=0.0), isa.vop("v_and_b32", WB, I32(WREG - 1), WB, target_gap=0.1)]
            rows = self.XSPLIT[i % 3]
            if i % 3 == 0:
                blk += self.xb_set((i // 3 + 1) % self.NGRP, 0.2)
            blk += self.w_reads((i + 1) % self.TAPS, c.rd_at * g, c.rd_step * g)
            blk += self.x_reads((i // 3 + 1) % self.NGRP, rows, (c.rd_at + 6 * c.rd_step) * g, c.rd_step * g)
        if "dma" not in abl:
            blk += self.w_dma(c.dma_at * g, c.dma_step * g, need=i + NWB - 1)
            blk += self.w_next((i + NWB) % self.TAPS, (c.dma_at + c.dma_step) * g + 1.5)
            if "patch" not in abl:
                for n, (j, k, nxt, need) in enumerate(self.PIECES.get(i, ())):
                    blk += self.patch_piece(j, k, S_XOFFN if nxt else S_XOFF, (c.p_at + 8.0 * n) * g, need)
                    if k == 5:
                        blk += self.slot_next(j, (c.p_at + 8.0 * n) * g + 1.0)
        # the fragment base of frame f + dt moves to the next slice's slot once the last group that reads through it has been requested
        # (kt = 3: groups 3 dt .. 3 dt + 2 are read during positions 9 dt - 3 .. 9 dt + 5; kt = 1: group 2 during positions 3..5)
        nxt_base = {5: 0} if self.kt1 else {10: 0, 19: 1, 26: 2}
        if i in nxt_base:
            blk += self.pbase_next(nxt_base[i], 40.0 * g)
        return blk

    def tap_block(self, tap: int) -> List[Instr]:
        """One tap of the unrolled slice body, scheduled: 48 MFMAs with the next tap's fragment reads, the W DMA of tap + 4 and a patch
        piece of the frames that are due in their gaps.  (The top -- waits + barrier -- is added by slice_body, which knows the DMA order.)"""
        c = self.cfg
        # (narrow tile: ~40 fillers for 8 MFMAs -- the tap is bound by instruction issue and the fragment reads, several fillers per gap)
        return sched.schedule(self.tap_fillers(tap) + self.mfmas(tap), cap=c.cap if self.NB == 6 else 6, lookahead=c.lookahead)

    def slice_body(self) -> List[Instr]:
        """One 32-channel slice = 27 taps (kt = 1: 9).  At the top of tap i: this tap's fragments are in registers (lgkmcnt 0), every DMA whose data
        the reads issued during tap i need has landed (counted vmcnt: DMAs retire in order) -- in every wave (barrier)."""
        c = self.cfg
        abl = c.abl.split(",")
        TAPS = self.TAPS
        blocks = [self.tap_block(t) for t in range(TAPS)]
        dmas = [[i for i in blk if i.cls == isa.LDS_DMA] for blk in blocks]
        o: List[Instr] = [isa.label("L_slice"), isa.nop(7)]
        for tap in range(TAPS):
            # DMAs in issue order up to here: the previous slice's (their deadlines are TAPS positions earlier) + this slice's before tap
            hist = [d.need - TAPS for blk in dmas for d in blk] + [d.need for blk in dmas[:tap] for d in blk]
            due = [k for k, need in enumerate(hist) if need <= tap]
            top = [isa.waitcnt(lgkmcnt=0)]
            if due:
                top.append(isa.waitcnt(vmcnt=len(hist) - 1 - due[-1]))
            if "bar" not in abl:
                top.append(isa.barrier())
            if c.cont and tap == 6:
                # the pieces of the "next slice" (frames 0 / 1 / 2 from positions 6 / 12 / 18 on) belong to the NEXT TILE during the last slice:
                # its frames 0, 1 are this tile's frames 2, 3; its frame 2 is input frame t0 + 4 - pt (slice 0: S_XOFFN is 0 there anyway)
                tfr = ST[9]
                top += [isa.sop("s_add_u32", tfr, S_T0, I32(4)), isa.sop("s_sub_u32", tfr, tfr, S_PT),
                        isa.sop("s_cmp_lt_u32", None, tfr, S_TI), isa.sop("s_cselect_b32", ST[10], S_FB.sub(0), I32(0)), isa.sop("s_cselect_b32", tfr, tfr, I32(0)),
                        isa.sop("s_mul_i32", ST[0], S_FB.sub(0), tfr), isa.sop("s_mul_hi_u32", ST[1], S_FB.sub(0), tfr),
                        isa.sop("s_mul_i32", ST[2], S_FB.sub(1), tfr), isa.sop("s_add_u32", ST[1], ST[1], ST[2]),
                        isa.sop("s_add_u32", ST[0], S_X.sub(0), ST[0]), isa.sop("s_addc_u32", ST[1], S_X.sub(1), ST[1]), isa.sop("s_and_b32", ST[1], ST[1], I32(0xFFFF)),
                        isa.sop("s_bitcmp1_b32", None, S_CONT, I32(2))]
                for k in range(4):
                    top += [isa.sop("s_cselect_b32", S_XR[0].sub(k), S_XR[2].sub(k), S_XR[0].sub(k)), isa.sop("s_cselect_b32", S_XR[1].sub(k), S_XR[3].sub(k), S_XR[1].sub(k))]
                top += [isa.sop("s_cselect_b32", S_XR[2].sub(0), ST[0], S_XR[2].sub(0)), isa.sop("s_cselect_b32", S_XR[2].sub(1), ST[1], S_XR[2].sub(1)),
                        isa.sop("s_cselect_b32", S_XR[2].sub(2), ST[10], S_XR[2].sub(2))]
            if c.cont and tap == 23:
                # ... and the W stream wraps to tap 0 of slice 0 (position 26 -> 0 was taken as "the next slice" during tap 22)
                top += [isa.sop("s_bitcmp1_b32", None, S_CONT, I32(2)), isa.sop("s_cselect_b32", S_WNEXT, I32(0), S_WNEXT)]
            o += top + blocks[tap]
        o += [isa.sop("s_add_u32", S_SL, S_SL, I32(1)), isa.sop("s_mov_b32", S_XOFF, S_XOFFN),
              isa.sop("s_add_u32", ST[0], S_SL, I32(1)), isa.sop("s_add_u32", ST[1], S_XOFF, I32(64)),
              isa.sop("s_cmp_lt_u32", None, ST[0], S_NSL), isa.sop("s_cselect_b32", S_XOFFN, ST[1], I32(0))]      # no slice after the last: re-read slice 0 (unused; cont: the next tile's)
        if c.cont:      # the slice that starts now is the last one of a tile whose successor continues the rings?
            o += [isa.sop("s_and_b32", ST[2], S_CONT, I32(1)), isa.sop("s_cmp_eq_u32", None, ST[0], S_NSL), isa.sop("s_cselect_b32", ST[2], ST[2], I32(0)),
                  isa.sop("s_lshl_b32", ST[2], ST[2], I32(2)), isa.sop("s_and_b32", S_CONT, S_CONT, I32(3)), isa.sop("s_or_b32", S_CONT, S_CONT, ST[2])]
        o += [isa.sop("s_cmp_lt_u32", None, S_SL, S_NSL), isa.branch("s_cbranch_scc1", "L_slice")]
        return o

    # ---- entry, per-tile setup, epilogue --------------------------------------------------------------------------------------------
    # The workgroups are PERSISTENT (one per compute unit): the setup and the first DMAs of the next tile are issued BEFORE the epilogue of
    # the tile just finished, whose stores then hide the DMA latency; a one-tile-per-workgroup launch measured 9.5 us of fixed cost per tile
    # (launch, kernel arguments, exposed first loads, store drain) against 38 us of taps on the 96-channel shape.
    def entry(self) -> List[Instr]:
        c = self.cfg
        t = T_
        o: List[Instr] = [isa.label(c.name)]
        o += [isa.s_load(8, S(8, 8), S_KARG, 0), isa.s_load(2, S_RES, S_KARG, 32), isa.s_load(4, S(20, 4), S_KARG, 40),
              isa.s_load(4, S(24, 4), S_KARG, 56), isa.s_load(4, S(28, 4), S_KARG, 72), isa.s_load(4, S(32, 4), S_KARG, 88),
              isa.s_load(2, S(36, 2), S_KARG, 104), isa.s_load(4, S(40, 4), S_KARG, 112), isa.s_load(2, S(4, 2), S_KARG, 128),
              isa.vop("v_and_b32", LANE, I32(63), V(0)), isa.vop("v_lshrrev_b32", t[0], I32(6), V(0)),
              isa.waitcnt(lgkmcnt=0), isa.vop("v_readfirstlane_b32", S_WAVE, t[0]),
              isa.sop("s_lshr_b32", S_F, S_WAVE, I32(1)), isa.sop("s_and_b32", S_RH, S_WAVE, I32(1))]
        # ---- this workgroup's tiles.  Workgroup number w = (b % 8) * G + b / 8 of the 8 G workgroups (the G workgroups of an XCD are neighbours):
        #   tiles_per_wg > 0 (one n tile): it takes base (+ 1 if w < rem) CONSECUTIVE tiles -- the frame pairs of a spatial tile;
        #   tiles_per_wg = 0 (several n tiles): it takes tiles w, w + 8 G, ...: at any time an XCD works on G consecutive tiles = the n tiles of
        #   a few neighbouring frame pairs of one spatial tile, which share their patch through the XCD's L2 ----
        wn, rem = ST[0], ST[1]
        o += [isa.sop("s_and_b32", wn, S_WG, I32(7)), isa.sop("s_mul_i32", wn, wn, S_G), isa.sop("s_lshr_b32", ST[2], S_WG, I32(3)),
              isa.sop("s_add_u32", wn, wn, ST[2]), isa.sop("s_lshl_b32", ST[3], S_G, I32(3)),                                          # ST[3] = grid
              isa.sop("s_cmp_eq_u32", None, S_PER, I32(0)), isa.branch("s_cbranch_scc1", "L_strided"),
              isa.sop("s_mul_i32", ST[2], ST[3], S_PER), isa.sop("s_sub_u32", rem, S_TEND, ST[2]),                                     # tiles - base * grid
              isa.sop("s_mul_i32", S_TILE, wn, S_PER), isa.sop("s_min_u32", ST[2], wn, rem), isa.sop("s_add_u32", S_TILE, S_TILE, ST[2]),
              isa.sop("s_cmp_lt_u32", None, wn, rem), isa.sop("s_cselect_b32", ST[2], I32(1), I32(0)), isa.sop("s_add_u32", ST[2], ST[2], S_PER),
              isa.sop("s_add_u32", S_TEND, S_TILE, ST[2]), isa.sop("s_mov_b32", S_STEP, I32(1)), isa.branch("s_branch", "L_ranged"),
              isa.label("L_strided"), isa.sop("s_mov_b32", S_TILE, wn), isa.sop("s_mov_b32", S_STEP, ST[3]),
              isa.label("L_ranged"),
              isa.sop("s_cmp_ge_u32", None, S_TILE, S_TEND), isa.branch("s_cbranch_scc1", "L_exit"),
              isa.sop("s_mov_b32", S_H0, I32(-1)), isa.sop("s_mov_b32", S_N0, I32(-1))]                          # no spatial / n tile yet
        # ---- constants ----
        o += [isa.sop("s_lshl_b32", S_KP2, S_KPAD, I32(1))]
        if self.kt1:       # input frame = (H >> ups) x (W >> ups) voxels (ups = the `pt` argument: no padding frames exist for kt = 1)
            o += [isa.sop("s_lshr_b32", ST[8], S_H, S_PT), isa.sop("s_lshr_b32", S_WI, S_Wd, S_PT), isa.sop("s_mul_i32", ST[8], ST[8], S_WI)]
        else:
            o += [isa.sop("s_mul_i32", ST[8], S_H, S_Wd)]
        o += [isa.sop("s_lshl_b32", S_CIN2, S_CIN, I32(1)),
              isa.sop("s_mul_i32", S_FB.sub(0), ST[8], S_CIN2), isa.sop("s_mul_hi_u32", S_FB.sub(1), ST[8], S_CIN2),
              isa.sop("s_mul_i32", S_C26, S_CIN2, I32(self.TAPS - 1)), isa.sop("s_sub_u32", S_C26, S_C26, I32(64)),
              isa.sop("s_mul_i32", S_C3, S_CIN2, I32(3)), isa.sop("s_mul_i32", S_C5, S_CIN2, I32(5))]
        # ---- W pieces of this wave: piece w + 4 i = rows 16 (w + 4 i) + l / 4, LDS position q = l % 4 holds source chunk q ^ ((row >> 1) & 3) ----
        o += [isa.vop("v_lshrrev_b32", t[1], I32(2), LANE), isa.vop("v_and_b32", t[2], I32(3), LANE)]
        for i in range(2):
            o += [isa.sop("s_add_u32", ST[0], S_WAVE, I32(4 * i)), isa.sop("s_lshl_b32", ST[0], ST[0], I32(4)),
                  isa.vop("v_add_u32", t[3], ST[0], t[1]),                                                        # row
                  isa.vop("v_lshrrev_b32", t[4], I32(1), t[3]), isa.vop("v_and_b32", t[4], I32(3), t[4]), isa.vop("v_xor_b32", t[4], t[2], t[4]),
                  isa.vop("v_mul_lo_u32", t[5], t[3], S_KP2), isa.vop("v_lshl_add_u32", WDMA[i], t[4], I32(4), t[5])]
        # ---- per-lane parts of the fragment bases ----
        ql, g = t[10], t[11]
        o += [isa.vop("v_and_b32", ql, I32(15), LANE), isa.vop("v_lshrrev_b32", g, I32(4), LANE)]
        if c.epi == 4:      # gamma quads (one n tile: the same 96 channels for every tile of the launch); landed long before the first epilogue
            assert not c.prof
            o += [isa.vop("v_lshlrev_b32", t[1], I32(4), g)] + [isa.global_load(4, V(EPI_GQ + 4 * nb, 4), t[1], 64 * nb, saddr=S_RES) for nb in range(6)]
        if c.epi in (5, 6, 7):  # the same through the kernel argument of their own (the residual pointer is in use); (5) the distance of the second output
            gp, yd = S(ST[6].idx, 2), S(ST[4].idx, 2)
            o += [isa.s_load(2, gp, S_KARG, KARG_GAMMA), isa.s_load(2, yd, S_KARG, KARG_Y2D), isa.waitcnt(lgkmcnt=0),
                  isa.vop("v_mov_b32", V(EPI_Y2D), yd.sub(0)), isa.vop("v_mov_b32", V(EPI_Y2D + 1), yd.sub(1)),
                  isa.vop("v_lshlrev_b32", t[1], I32(4), g)] + [isa.global_load(4, V(EPI_GQ + 4 * nb, 4), t[1], 64 * nb, saddr=gp) for nb in range(6)]
        # patch: row 8 rh of the slot of frame f (+ dt slots, + the ring position: tile_setup); the lane's place in a row per column shift
        o += [isa.sop("s_mul_i32", ST[0], S_RH, I32(8 * ROWB)), isa.sop("s_mul_i32", ST[1], S_F, I32(FSLOT)), isa.sop("s_add_u32", ST[1], ST[1], ST[0]),
              isa.sop("s_add_u32", ST[1], ST[1], I32(PBASE0)), isa.vop("v_mov_b32", PBL, ST[1])]
        for dw in range(3):
            u, fz = t[1], t[2]
            o += [isa.vop("v_add_u32", u, I32(dw), ql), isa.vop("v_lshrrev_b32", fz, I32(1), u), isa.vop("v_and_b32", fz, I32(3), fz),
                  isa.vop("v_xor_b32", fz, g, fz), isa.vop("v_lshlrev_b32", fz, I32(4), fz), isa.vop("v_lshl_add_u32", LP[dw], u, I32(6), fz)]
        # W: row ql (64 B), chunk g ^ ((ql >> 1) & 3)
        o += [isa.vop("v_lshrrev_b32", t[1], I32(1), ql), isa.vop("v_and_b32", t[1], I32(3), t[1]), isa.vop("v_xor_b32", t[1], g, t[1]),
              isa.vop("v_lshlrev_b32", t[2], I32(6), ql), isa.vop("v_lshl_add_u32", WBL, t[1], I32(4), t[2])]
        # ---- epilogue staging addresses ----
        sb = ST[2]
        # (kt = 1: slot 4 -- its four patch frames of two consecutive slices live in slots 0..3, and the next tile's first frames arrive in 0, 1)
        o += [isa.sop("s_mul_i32", sb, S_WAVE, I32(STG_WAVE)), isa.sop("s_add_u32", sb, sb, I32(PBASE0 + (4 if self.kt1 else 3) * FSLOT)),
              isa.sop("s_lshl_b32", ST[3], S_LDC.sub(0), I32(1)), isa.sop("s_lshl_b32", ST[4], S_LDR.sub(0), I32(1)),
              isa.vop("v_mul_u32_u24", t[1], I32(STG_VOX), ql), isa.vop("v_lshl_add_u32", t[1], g, I32(3), t[1]), isa.vop("v_add_u32", E_W, sb, t[1])]
        for i in range(3):
            j, vx, c12 = t[1], t[2], t[3]
            o += [isa.vop("v_add_u32", j, I32(64 * i), LANE), isa.vop("v_mul_u32_u24", vx, I32(171), j), isa.vop("v_lshrrev_b32", vx, I32(11), vx),   # j / 12 (j < 192)
                  isa.vop("v_mul_u32_u24", c12, I32(12), vx), isa.vop("v_sub_u32", c12, j, c12), isa.vop("v_lshlrev_b32", c12, I32(4), c12),
                  isa.vop("v_mov_b32", E_V[i], vx),
                  isa.vop("v_mul_u32_u24", t[4], I32(STG_VOX), vx), isa.vop("v_add_u32", t[4], t[4], c12), isa.vop("v_add_u32", E_R[i], sb, t[4]),
                  isa.vop("v_mul_lo_u32", t[4], vx, ST[3]), isa.vop("v_add_u32", E_Y[i], t[4], c12),
                  isa.vop("v_mul_lo_u32", t[4], vx, ST[4]), isa.vop("v_add_u32", E_Z[i], t[4], c12)]
        o += [isa.sop("s_mov_b32", self.HAVE_PREV, I32(0))]       # no tile waits for its epilogue yet
        if c.cont:
            o += [isa.sop("s_mov_b32", S_CONT, I32(0))]
        if c.stagger:
            # All workgroups start together and a tile takes every one of them the same time: unstaggered, the first loads of 256 tiles
            # (27 MB) and the stores of 256 tiles (25 MB) hit HBM at once and each costs ~7-10 k cycles of blocked VMEM issue per tile
            # (profiles/r03_conv4_phases.log) while the memory idles during the taps.
            sl = Instr("s_sleep", cls=isa.SALU)
            sl.text = f"s_sleep {c.stagger}"
            o += [isa.sop("s_lshr_b32", ST[0], S_WG, I32(3)), isa.label("L_stagger"), isa.sop("s_cmp_eq_u32", None, ST[0], I32(0)),
                  isa.branch("s_cbranch_scc1", "L_staggered"), sl, isa.sop("s_sub_u32", ST[0], ST[0], I32(1)), isa.branch("s_branch", "L_stagger"),
                  isa.label("L_staggered")]
        if c.prof:
            o += [isa.sop("s_mov_b32", S_PROFWG, S_WG)] + self.prof_init() + self.stamp(-1)
        return sched.pad_hazards(sched.insert_lgkm_waits(o))

    HAVE_PREV = S(41)                                           # (the unused high word of ldc)  0: first tile; 1: a finished tile's accumulators wait; 3: ... and no tile follows

    def tile_setup(self) -> List[Instr]:
        """S_TILE -> coordinates, descriptors, per-lane patch offsets, ring positions; the first DMAs (frames 0, 1, 2 of slice 0, W taps 0..3),
        the bias quads."""
        t = T_
        o: List[Instr] = [isa.label("L_tile"), isa.nop(7)]
        o += [isa.sop("s_bitcmp1_b32", None, self.HAVE_PREV, I32(1)), isa.branch("s_cbranch_scc1", "L_epilogue")]     # nothing follows: only the epilogue
        tt = ST[4]
        q1, q2, q3 = ST[5], ST[6], ST[7]
        nh0, nw0, nn0, same, same_n = ST[8], ST[11], ST[12], ST[14], ST[15]
        o += [isa.sop("s_lshl_b32", tt, S_TILE, I32(1)), isa.sop("s_mul_hi_u32", q1, tt, S_MGN),                # q1 = tile / tiles_n
              isa.sop("s_mul_i32", tt, q1, S_TLN), isa.sop("s_sub_u32", tt, S_TILE, tt), isa.sop("s_mul_i32", nn0, tt, I32(96)),
              isa.sop("s_lshl_b32", tt, q1, I32(1)), isa.sop("s_mul_hi_u32", q2, tt, S_MGT),                   # q2 = q1 / frame pairs
              isa.sop("s_mul_i32", tt, q2, S_TLT), isa.sop("s_sub_u32", tt, q1, tt), isa.sop("s_lshl_b32", S_T0, tt, I32(1)),
              isa.sop("s_lshl_b32", tt, q2, I32(1)), isa.sop("s_mul_hi_u32", q3, tt, S_MGW),                   # q3 = q2 / tiles_w = tile row
              isa.sop("s_mul_i32", tt, q3, S_TLW), isa.sop("s_sub_u32", tt, q2, tt), isa.sop("s_lshl_b32", nw0, tt, I32(4)),
              isa.sop("s_lshl_b32", nh0, q3, I32(4)),
              # the same spatial tile as before (another n tile / the next frame pair): the per-lane patch offsets stay; the same n tile: the
              # W descriptor and the bias quads stay
              isa.sop("s_xor_b32", ST[0], nh0, S_H0), isa.sop("s_xor_b32", ST[1], nw0, S_W0), isa.sop("s_or_b32", ST[0], ST[0], ST[1]),
              isa.sop("s_cmp_eq_u32", None, ST[0], I32(0)), isa.sop("s_cselect_b32", same, I32(1), I32(0)),
              isa.sop("s_cmp_eq_u32", None, nn0, S_N0), isa.sop("s_cselect_b32", same_n, I32(1), I32(0)),
              isa.sop("s_mov_b32", S_H0, nh0), isa.sop("s_mov_b32", S_W0, nw0), isa.sop("s_mov_b32", S_N0, nn0)]
        if self.cfg.cont:
            # bit 0: the workgroup's NEXT tile is the next frame pair of this spatial tile (a run of consecutive tiles: step 1, one n tile)
            o += [isa.sop("s_and_b32", S_CONT, S_CONT, I32(2)),
                  isa.sop("s_add_u32", ST[0], S_TILE, S_STEP), isa.sop("s_cmp_lt_u32", None, ST[0], S_TEND), isa.sop("s_cselect_b32", ST[0], I32(1), I32(0)),
                  isa.sop("s_cmp_eq_u32", None, S_STEP, I32(1)), isa.sop("s_cselect_b32", ST[0], ST[0], I32(0)),
                  isa.sop("s_lshr_b32", ST[1], S_T0, I32(1)), isa.sop("s_add_u32", ST[1], ST[1], I32(1)), isa.sop("s_cmp_lt_u32", None, ST[1], S_TLT),
                  isa.sop("s_cselect_b32", ST[0], ST[0], I32(0)), isa.sop("s_or_b32", S_CONT, S_CONT, ST[0])]
        # ---- patch frame j: input frame t = t0 - pt + j; descriptor base = x + t * FB, num_records = FB (0 when t is outside [0, Ti)) ----
        tfr = ST[9]
        for j in range(self.NFR):
            o += [isa.sop("s_add_u32", tfr, S_T0, I32(j))] + ([] if self.kt1 else [isa.sop("s_sub_u32", tfr, tfr, S_PT)]) + [   # may wrap below 0 -> huge unsigned
                  isa.sop("s_cmp_lt_u32", None, tfr, S_TI), isa.sop("s_cselect_b32", ST[10], S_FB.sub(0), I32(0)),   # num_records
                  isa.sop("s_cselect_b32", tfr, tfr, I32(0)),
                  isa.sop("s_mul_i32", ST[0], S_FB.sub(0), tfr), isa.sop("s_mul_hi_u32", ST[1], S_FB.sub(0), tfr),
                  isa.sop("s_mul_i32", ST[2], S_FB.sub(1), tfr), isa.sop("s_add_u32", ST[1], ST[1], ST[2]),
                  isa.sop("s_add_u32", S_XR[j].sub(0), S_X.sub(0), ST[0]), isa.sop("s_addc_u32", ST[1], S_X.sub(1), ST[1]),
                  isa.sop("s_and_b32", S_XR[j].sub(1), ST[1], I32(0xFFFF)), isa.sop("s_mov_b32", S_XR[j].sub(2), ST[10]),
                  isa.sop("s_mov_b32", S_XR[j].sub(3), I32(0x00020000))]
        # ---- ring positions: slice 0's frame j -> slot j; W buffer 0 ----  (cont: a prefetched tile keeps the rings where its predecessor left them)
        if self.cfg.cont:
            o += [isa.sop("s_bitcmp1_b32", None, S_CONT, I32(1)), isa.branch("s_cbranch_scc1", "L_keep_rings")]
        o += [isa.sop("s_lshl_b32", S_WM0, S_WAVE, I32(10)), isa.sop("s_add_u32", ST[0], S_WM0, I32(PBASE0))]
        for j in range(self.NFR):
            o.append(isa.sop("s_add_u32", S_SLOT[j], ST[0], I32(j * FSLOT)))
        o += [isa.vop("v_mov_b32", PBASE[0], PBL), isa.vop("v_add_u32", PBASE[1], I32(FSLOT), PBL), isa.vop("v_add_u32", PBASE[2], I32(2 * FSLOT), PBL),
              isa.vop("v_mov_b32", WB, WBL)]
        if self.cfg.cont:
            o += [isa.label("L_keep_rings"), isa.nop(7)]
        o += [isa.sop("s_cmp_eq_u32", None, same_n, I32(1)), isa.branch("s_cbranch_scc1", "L_same_n")]
        # ---- W descriptor: base = w + n0 * Kpad * 2, num_records = min(96, N - n0) * Kpad * 2 ----
        o += [isa.sop("s_mul_i32", ST[0], S_N0, S_KP2), isa.sop("s_mul_hi_u32", ST[1], S_N0, S_KP2),
              isa.sop("s_add_u32", S_WR.sub(0), S_Wp.sub(0), ST[0]), isa.sop("s_addc_u32", ST[1], S_Wp.sub(1), ST[1]),
              isa.sop("s_and_b32", S_WR.sub(1), ST[1], I32(0xFFFF)),
              isa.sop("s_sub_u32", ST[2], S_N, S_N0), isa.sop("s_min_u32", ST[2], ST[2], I32(96)), isa.sop("s_mul_i32", S_WR.sub(2), ST[2], S_KP2),
              isa.sop("s_mov_b32", S_WR.sub(3), I32(0x00020000))]
        # ---- bias quads of this lane's channels n0 + 16 nb + 4 (l / 16) + e (zeros when bias == NULL): the accumulators start from them ----
        g = t[11]
        for i in range(4 * self.NB):
            o.append(isa.vop("v_mov_b32", V(EPI_BQ + i), I32(0)))
        o += [isa.sop("s_cmp_eq_u64", None, S_BIAS, I32(0)), isa.branch("s_cbranch_scc1", "L_nobias"),
              isa.sop("s_lshl_b32", ST[7], S_N0, I32(2)), isa.sop("s_add_u32", ST[2], S_BIAS.sub(0), ST[7]),
              isa.sop("s_addc_u32", ST[3], S_BIAS.sub(1), I32(0)), isa.vop("v_lshlrev_b32", t[1], I32(4), g)]
        if self.NB == 6:
            for nb in range(6):
                o.append(isa.global_load(4, V(EPI_BQ + 4 * nb, 4), t[1], 64 * nb, saddr=S(ST[2].idx, 2)))
        else:       # N may be 8: a range-checked load (the quads of lanes 32..63 lie past the bias array and read zeros)
            bd = S(ST[6].idx, 4)       # s88..s91: a 4-aligned quad of temporaries that are free here
            o += [isa.sop("s_mov_b32", bd.sub(0), ST[2]), isa.sop("s_and_b32", bd.sub(1), ST[3], I32(0xFFFF)), isa.sop("s_lshl_b32", bd.sub(2), S_N, I32(2)),
                  isa.sop("s_mov_b32", bd.sub(3), I32(0x00020000)), isa.buffer_load(4, V(EPI_BQ, 4), t[1], bd, I32(0))]
        o += [isa.label("L_nobias"), isa.nop(7), isa.label("L_same_n"), isa.nop(7),
              isa.sop("s_cmp_eq_u32", None, same, I32(1)), isa.branch("s_cbranch_scc1", "L_same")]
        # ---- patch pieces of this wave: piece wave + 4 i = patch voxels p = 16 (wave + 4 i) + l / 4 = (r, col) of the 18 x 20 LDS grid; LDS
        # position q = l % 4 of a voxel holds source chunk q ^ ((col >> 1) & 3) (conflict-free fragment reads, like the W tile); source offset
        # inside the frame, or OOB (padding columns 18, 19, voxels past 360, rows / columns outside the tensor) ----
        hm1, wm1 = ST[12], ST[13]
        o += [isa.sop("s_sub_u32", hm1, S_H0, I32(1)), isa.sop("s_sub_u32", wm1, S_W0, I32(1)), isa.sop("s_lshl_b32", ST[1], S_WAVE, I32(4)),
              isa.vop("v_lshrrev_b32", t[12], I32(2), LANE), isa.vop("v_and_b32", t[13], I32(3), LANE)]
        for k in range(6):
            pv, r, col, hi, wi, ok = t[1], t[2], t[3], t[4], t[5], t[6]
            o += [isa.sop("s_add_u32", ST[2], ST[1], I32(64 * k)), isa.vop("v_add_u32", pv, ST[2], t[12]),
                  isa.vop("v_mul_u32_u24", r, I32(3277), pv), isa.vop("v_lshrrev_b32", r, I32(16), r),             # r = pv / 20 (pv < 384)
                  isa.vop("v_mul_u32_u24", col, I32(PCL), r), isa.vop("v_sub_u32", col, pv, col),
                  isa.vop("v_add_u32", hi, hm1, r), isa.vop("v_add_u32", wi, wm1, col),                             # wraps below 0 -> huge unsigned
                  isa.v_cmp("v_cmp_gt_u32", S_H, hi)] + ([isa.vop("v_lshrrev_b32", t[14], S_PT, hi), isa.vop("v_mul_lo_u32", t[7], t[14], S_WI)] if self.kt1 else
                                                           [isa.vop("v_mul_lo_u32", t[7], hi, S_Wd)]) + [
                  isa.v_cndmask(ok, I32(0), I32(1)),
                  isa.v_cmp("v_cmp_gt_u32", S_Wd, wi), isa.v_cndmask(t[8], I32(0), I32(1)), isa.vop("v_and_b32", ok, ok, t[8]),
                  isa.v_cmp("v_cmp_gt_u32", I32(PC), col), isa.v_cndmask(t[8], I32(0), I32(1)), isa.vop("v_and_b32", ok, ok, t[8]),
                  isa.v_cmp("v_cmp_gt_u32", I32(PR * PCL), pv), isa.v_cndmask(t[8], I32(0), I32(1)), isa.vop("v_and_b32", ok, ok, t[8]),
                  ] + ([isa.vop("v_lshrrev_b32", t[14], S_PT, wi), isa.vop("v_add_u32", t[7], t[7], t[14])] if self.kt1 else [isa.vop("v_add_u32", t[7], t[7], wi)]) + [
                  isa.vop("v_mul_lo_u32", t[7], t[7], S_CIN2),
                  isa.vop("v_lshrrev_b32", t[8], I32(1), col), isa.vop("v_and_b32", t[8], I32(3), t[8]), isa.vop("v_xor_b32", t[8], t[13], t[8]),
                  isa.vop("v_lshl_add_u32", t[7], t[8], I32(4), t[7]),
                  isa.v_cmp("v_cmp_ne_u32", I32(0), ok), isa.vop("v_mov_b32", t[9], I32(OOB)),
                  isa.v_cndmask(PDMA[k], t[9], t[7])]
        o += [isa.label("L_same"), isa.nop(7)]
        o += self.stamp(0)                                        # phase 0: tile decode, descriptors, lane offsets
        # ---- streams: patch frames 0, 1, 2 of slice 0, W taps 0 .. 3 ----
        o += [isa.sop("s_mov_b32", S_SL, I32(0)), isa.sop("s_mov_b32", S_XOFF, I32(0))] + ([] if self.cfg.cont else [isa.sop("s_mov_b32", S_WNEXT, I32(0))]) + [
              isa.sop("s_cmp_lt_u32", None, I32(1), S_NSL), isa.sop("s_cselect_b32", S_XOFFN, I32(64), I32(0))]
        if self.cfg.cont:
            # the last slice of this tile prefetches for the next one?  (one slice: this is the last)
            o += [isa.sop("s_and_b32", ST[0], S_CONT, I32(1)), isa.sop("s_cmp_eq_u32", None, S_NSL, I32(1)), isa.sop("s_cselect_b32", ST[0], ST[0], I32(0)),
                  isa.sop("s_lshl_b32", ST[0], ST[0], I32(2)), isa.sop("s_and_b32", S_CONT, S_CONT, I32(3)), isa.sop("s_or_b32", S_CONT, S_CONT, ST[0]),
                  isa.sop("s_bitcmp1_b32", None, S_CONT, I32(1)), isa.branch("s_cbranch_scc1", "L_prefetched"),
                  isa.sop("s_mov_b32", S_WNEXT, I32(0))]
        first = 2 if self.kt1 else 3
        for j in range(first):
            for k in range(6):
                o += self.patch_piece(j, k, S_XOFF, 0, 0)
        # where the body's loads go: frame 3 of slice 0 -> slot 3; frames 0, 1, 2 of slice 1 -> slots 4, 0, 1  (kt = 1: slice 1 -> slots 2, 3)
        for j in range(first):
            o += self.slot_next(j, 0)
        for b in range(NWB):
            o += self.w_dma(0, 0, 0) + self.w_next(b, 0)
        if self.cfg.cont:
            o += [isa.label("L_prefetched"), isa.nop(7)]
        o += self.stamp(1) + [                                             # phase 1: DMA + bias issue
              isa.sop("s_cmp_eq_u32", None, self.HAVE_PREV, I32(0)), isa.branch("s_cbranch_scc1", "L_first")]
        return sched.pad_hazards(o)

    def tile_start(self) -> List[Instr]:
        """accumulators = bias, first DMAs landed (every wave), first fragments."""
        o: List[Instr] = [isa.label("L_first"), isa.waitcnt(vmcnt=0), isa.label("L_start"), isa.nop(7)]
        for nb in range(self.NB):
            for mb in range(8):
                for i in range(4):
                    o.append(isa.vop("v_accvgpr_write_b32", ACC(nb, mb).sub(i), V(EPI_BQ + 4 * nb + i)))
        o += [isa.barrier()]
        o += self.xb_set(0, 0) + self.w_reads(0, 0, 0) + self.x_reads(0, range(10), 0, 0)
        o = sched.pad_hazards(o)
        return o + self.stamp(5)                                  # phase 5: accumulators = bias, barrier (first tile: + the wait for the first loads)

    def tile_end(self) -> List[Instr]:
        """after the last slice: every wave is done with the LDS contents; remember the tile for the epilogue; next tile (or none)."""
        o = self.stamp(6) + [isa.waitcnt(lgkmcnt=0), isa.barrier()] + self.stamp(7) + [       # phase 6: the slices; 7: the closing barrier
             isa.sop("s_mov_b32", S_ET0, S_T0), isa.sop("s_mov_b32", S_EH0, S_H0), isa.sop("s_mov_b32", S_EW0, S_W0), isa.sop("s_mov_b32", S_EN0, S_N0),
             isa.sop("s_add_u32", S_TILE, S_TILE, S_STEP)] + ([isa.sop("s_and_b32", S_CONT, S_CONT, I32(1)), isa.sop("s_lshl_b32", S_CONT, S_CONT, I32(1))] if self.cfg.cont else []) + [
             isa.sop("s_cmp_lt_u32", None, S_TILE, S_TEND),
             isa.sop("s_cselect_b32", self.HAVE_PREV, I32(1), I32(3)), isa.branch("s_branch", "L_tile")]
        return sched.pad_hazards(o)

    def epilogue_narrow(self) -> List[Instr]:
        """Cfg.nb = 1 (N <= 16, one n tile, no residual): lane = voxel (frame t0 + f, row h0 + 8 rh + mb, column w0 + l % 16), channels
        4 (l / 16) + e.  A voxel's <= 16 channels are <= 32 contiguous bytes and the 16 voxels of a row block are neighbours in memory, so the
        8-byte stores of a wave instruction already fill whole runs (16 x ldc x 2 bytes): no staging through LDS.  Lanes whose channels lie
        past N do not store (N = 8: lanes 32..63)."""
        t = T_
        e: List[Instr] = [isa.label("L_epilogue"), isa.nop(15), isa.nop(15)]
        tf, tfo, hw = ST[4], ST[5], ST[6]
        ldc2 = ST[13]
        e += [isa.sop("s_add_u32", tf, S_ET0, S_F), isa.sop("s_mul_i32", tfo, tf, S_OTM), isa.sop("s_add_u32", tfo, tfo, S_OTO),
              isa.sop("s_mul_i32", hw, S_H, S_Wd), isa.sop("s_lshl_b32", ldc2, S_LDC.sub(0), I32(1)),
              isa.sop("s_mul_i32", ST[0], hw, ldc2), isa.sop("s_mul_hi_u32", ST[1], hw, ldc2),                 # frame bytes (64 bit)
              isa.sop("s_mul_i32", ST[2], ST[0], tfo), isa.sop("s_mul_hi_u32", ST[3], ST[0], tfo), isa.sop("s_mul_i32", ST[7], ST[1], tfo),
              isa.sop("s_add_u32", ST[3], ST[3], ST[7]),
              isa.sop("s_add_u32", S_YF.sub(0), S_Y.sub(0), ST[2]), isa.sop("s_addc_u32", S_YF.sub(1), S_Y.sub(1), ST[3]),
              isa.sop("s_cmp_lt_u32", None, tf, S_T), isa.sop("s_cselect_b32", ST[8], S_Wd, I32(0))]           # frame outside [0, To): no column is valid
        row0 = ST[9]
        e += [isa.sop("s_lshl_b32", row0, S_RH, I32(3)), isa.sop("s_add_u32", row0, row0, S_EH0)]
        ql, g, vcol, loff, voff = t[10], t[11], t[12], t[13], t[4]
        e += [isa.vop("v_and_b32", ql, I32(15), LANE), isa.vop("v_lshrrev_b32", g, I32(4), LANE),
              isa.vop("v_add_u32", vcol, S_EW0, ql), isa.vop("v_lshlrev_b32", t[14], I32(2), g),
              isa.v_cmp("v_cmp_gt_u32", S_N, t[14]), isa.vop("v_mov_b32", t[15], I32(0x7FFFFFFF)), isa.v_cndmask(vcol, t[15], vcol),   # channels past N: never valid
              isa.vop("v_mul_lo_u32", loff, ql, ldc2), isa.vop("v_lshl_add_u32", loff, g, I32(3), loff)]
        OUT = lambda mb: V(2 * mb, 2)
        f = [V(120 + i) for i in range(4)]
        for mb in range(8):
            for i in range(4):
                e.append(isa.vop("v_accvgpr_read_b32", f[i], ACC(0, mb).sub(i)))
            e += [isa.vop("v_cvt_pk_bf16_f32", OUT(mb).sub(0), f[0], f[1]), isa.vop("v_cvt_pk_bf16_f32", OUT(mb).sub(1), f[2], f[3])]
        e += self.stamp(2)
        e.append(isa.waitcnt(vmcnt=0))        # the next tile's first loads have landed (see epilogue)
        e += self.stamp(3)
        for mb in range(8):
            hrow = ST[10]
            e += [isa.sop("s_add_u32", hrow, row0, I32(mb)),
                  isa.sop("s_cmp_lt_u32", None, hrow, S_H), isa.sop("s_cselect_b32", ST[11], ST[8], I32(0)),
                  isa.sop("s_mul_i32", ST[12], hrow, S_Wd), isa.sop("s_add_u32", ST[12], ST[12], S_EW0), isa.sop("s_mul_i32", ST[7], ST[12], ldc2)]
            st = isa.global_store(2, voff, OUT(mb), 0, saddr=S_YF, extra_reads=[EXEC])
            if self.cfg.nt:
                st.text += " nt"
            e += [isa.v_cmp("v_cmp_gt_u32", ST[11], vcol),
                  Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU),
                  isa.vop("v_add_u32", voff, ST[7], loff), st, Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)]
        e += self.stamp(4)
        e += [isa.sop("s_bitcmp1_b32", None, self.HAVE_PREV, I32(1)), isa.branch("s_cbranch_scc1", "L_done" if self.cfg.prof else "L_exit"), isa.branch("s_branch", "L_start")]
        return sched.pad_hazards(sched.insert_lgkm_waits(e))

    def epilogue(self) -> List[Instr]:
        """lane: voxel (frame t0 + f, row h0 + 8 rh + mb, column w0 + l % 16), channels n0 + 16 nb + 4 (l / 16) + e of tile (S_ET0, ...);
        y / resid rows: voxel index ((frame * ot_mul + ot_off) * H + row) * W + column, strides ldc / ldr elements.  The bias is in the
        accumulators already."""
        if self.NB == 1:
            return self.epilogue_narrow()
        c = self.cfg
        t = T_
        e: List[Instr] = [isa.label("L_epilogue"), isa.nop(15), isa.nop(15)]
        ql, g = t[10], t[11]
        tf, tfo, hw = ST[4], ST[5], ST[6]
        ldc2, ldr2 = ST[13], S_LDR2
        e += [isa.sop("s_add_u32", tf, S_ET0, S_F),                                                              # output frame
              isa.sop("s_mul_i32", tfo, tf, S_OTM), isa.sop("s_add_u32", tfo, tfo, S_OTO),                       # its frame slot in y / resid
              isa.sop("s_mul_i32", hw, S_H, S_Wd),
              isa.sop("s_lshl_b32", ldc2, S_LDC.sub(0), I32(1)), isa.sop("s_lshl_b32", ldr2, S_LDR.sub(0), I32(1))]
        # frame bases (64 bit): ptr + slot * H * W * ld * 2 + n0 * 2
        for base, src, ld2 in ((S_YF, S_Y, ldc2), (S_RF, S_RES, ldr2)):
            e += [isa.sop("s_mul_i32", ST[0], hw, ld2), isa.sop("s_mul_hi_u32", ST[1], hw, ld2),                 # frame bytes (64 bit)
                  isa.sop("s_mul_i32", ST[2], ST[0], tfo), isa.sop("s_mul_hi_u32", ST[3], ST[0], tfo), isa.sop("s_mul_i32", ST[7], ST[1], tfo),
                  isa.sop("s_add_u32", ST[3], ST[3], ST[7]),
                  isa.sop("s_add_u32", base.sub(0), src.sub(0), ST[2]), isa.sop("s_addc_u32", base.sub(1), src.sub(1), ST[3]),
                  isa.sop("s_lshl_b32", ST[7], S_EN0, I32(1)),
                  isa.sop("s_add_u32", base.sub(0), base.sub(0), ST[7]), isa.sop("s_addc_u32", base.sub(1), base.sub(1), I32(0))]
        # frame validity is the same for all 8 row blocks; lane validity = its voxel's column (memory layout: voxel E_V[i] of the 16)
        VCOL = [t[12 + i] for i in range(3)]
        e += [isa.vop("v_add_u32", VCOL[i], S_EW0, E_V[i]) for i in range(3)]
        e += [isa.sop("s_cmp_lt_u32", None, tf, S_T), isa.sop("s_cselect_b32", ST[8], S_Wd, I32(0))]      # frame outside [0, To): no column is valid
        row0 = ST[9]
        e += [isa.sop("s_lshl_b32", row0, S_RH, I32(3)), isa.sop("s_add_u32", row0, row0, S_EH0)]
        voff = t[4]
        EW, ER = E_W, E_R
        if c.cont:
            # The staging strip normally sits in frame slot 3, which is idle between a tile's last tap and the next tile's first loads (these go to
            # slots 0, 1, 2).  When the next tile was PREFETCHED the ring did not restart and any slot may hold one of its frames: the strip moves to
            # the slot of this tile's frame 3 of its last slice (dead since the last tap) = one slot above where the next frame 3 will be loaded.
            EW, ER = t[20], [t[21], t[22], t[23]]
            e += [isa.sop("s_lshl_b32", ST[0], S_WAVE, I32(10)), isa.sop("s_sub_u32", ST[0], S_SLOT[3], ST[0]), isa.sop("s_add_u32", ST[0], ST[0], I32(FSLOT)),
                  isa.sop("s_cmp_ge_u32", None, ST[0], I32(PBASE0 + NSLOT * FSLOT)), isa.sop("s_cselect_b32", ST[1], I32(NSLOT * FSLOT), I32(0)),
                  isa.sop("s_sub_u32", ST[0], ST[0], ST[1]), isa.sop("s_sub_u32", ST[0], ST[0], I32(PBASE0 + 3 * FSLOT)),
                  isa.sop("s_bitcmp1_b32", None, S_CONT, I32(1)), isa.sop("s_cselect_b32", ST[0], ST[0], I32(0)),
                  isa.vop("v_add_u32", EW, ST[0], E_W)] + [isa.vop("v_add_u32", ER[i], ST[0], E_R[i]) for i in range(3)]

        def row_sgprs(mb, cols, rowoff, ld2):
            """cols = columns allowed in row block mb (0: the row / frame is outside the tensor), rowoff = byte offset of its first voxel."""
            hrow = ST[10]
            return [isa.sop("s_add_u32", hrow, row0, I32(mb)),
                    isa.sop("s_cmp_lt_u32", None, hrow, S_H), isa.sop("s_cselect_b32", cols, ST[8], I32(0)),
                    isa.sop("s_mul_i32", ST[12], hrow, S_Wd), isa.sop("s_add_u32", ST[12], ST[12], S_EW0), isa.sop("s_mul_i32", rowoff, ST[12], ld2)]

        def masked(cols, i, ins_list):
            return ([isa.v_cmp("v_cmp_gt_u32", cols, VCOL[i]),
                     Instr("s_and_saveexec_b64", [S_SAVE], [VCC], extra_reads=[EXEC], extra_writes=[EXEC, isa.SCC], cls=isa.SALU)] + ins_list +
                    [Instr("s_mov_b64", [EXEC], [S_SAVE], cls=isa.SALU)])

        OUT = lambda mb, nb: V(mb * 12 + 2 * nb, 2)         # packed outputs in accumulator layout (the fragment registers v0..v95 are free now)
        LQ = lambda mb, i: V(mb * 12 + 4 * i, 4)            # (e3) residual row block mb in memory layout: the same 12 registers
        RB = lambda k, nb: V(96 + 12 * k + 2 * nb, 2)       # (e3) residual pairs back in accumulator layout, two row blocks in flight
        RQ = lambda k, i: V(96 + 12 * k + 4 * i, 4)         # output row block in memory layout, two in flight
        F = lambda k: [V(120 + 4 * k + i) for i in range(4)]

        with_resid = c.epi in (3, 5, 6)

        def to_packed(mb, k):
            """accumulators of row block mb (+ residual pairs RB[k]) -> packed bf16 OUT(mb, .)."""
            r = []
            for nb in range(6):
                f = F(nb % 2)
                acc = ACC(nb, mb)
                for i in range(4):
                    r.append(isa.vop("v_accvgpr_read_b32", f[i], acc.sub(i)))
                if with_resid:
                    r_ = t[16 + (nb % 2)]
                    for i in range(4):
                        src = RB(k, nb).sub(i >> 1)
                        r += [isa.vop("v_lshlrev_b32", r_, I32(16), src) if (i & 1) == 0 else isa.vop("v_and_b32", r_, I32(0xFFFF0000), src),
                              isa.vop("v_add_f32", f[i], f[i], r_)]
                r += [isa.vop("v_cvt_pk_bf16_f32", OUT(mb, nb).sub(0), f[0], f[1]), isa.vop("v_cvt_pk_bf16_f32", OUT(mb, nb).sub(1), f[2], f[3])]
            return r

        def norm_silu(mb):
            """(epi 4) OUT(mb, .) holds the bf16-rounded conv + bias of the lane's 24 channels of one voxel (lanes l, l ^ 16, l ^ 32, l ^ 48 hold the
            same voxel's other channels): x * sqrt(96) / max(||x||, 1e-12) * gamma, SiLU, packed back in place -- the arithmetic of
            rms_silu_kernel (csrc/conv.hip) on what scail_conv4_e0 would have stored (reference RMS_norm = F.normalize * sqrt(C) * gamma,
            wan_vae.py:39-54)."""
            ss, tt = t[18], t[19]
            f0, f1, u0, u1 = F(0)
            r = []
            first = True
            for nb in range(6):
                for h in range(2):
                    src = OUT(mb, nb).sub(h)
                    r += [isa.vop("v_lshlrev_b32", f0, I32(16), src), isa.vop("v_and_b32", f1, I32(0xFFFF0000), src),
                          isa.vop("v_mul_f32", ss, f0, f0) if first else isa.vop("v_fma_f32", ss, f0, f0, ss), isa.vop("v_fma_f32", ss, f1, f1, ss)]
                    first = False
            r += [isa.vop("v_mov_b32", tt, ss), isa.permlane32_swap(ss, tt), isa.vop("v_add_f32", ss, ss, tt),
                  isa.vop("v_mov_b32", tt, ss), isa.permlane16_swap(ss, tt), isa.vop("v_add_f32", ss, ss, tt),
                  isa.vop("v_sqrt_f32", ss, ss), isa.vop("v_max_f32", ss, F32(1e-12), ss), isa.vop("v_rcp_f32", ss, ss),
                  isa.vop("v_mul_f32", ss, F32(96.0 ** 0.5), ss)]
            for nb in range(6):
                for h in range(2):
                    src = OUT(mb, nb).sub(h)
                    g0, g1 = V(EPI_GQ + 4 * nb + 2 * h), V(EPI_GQ + 4 * nb + 2 * h + 1)
                    r += [isa.vop("v_lshlrev_b32", f0, I32(16), src), isa.vop("v_and_b32", f1, I32(0xFFFF0000), src),
                          isa.vop("v_mul_f32", f0, f0, ss), isa.vop("v_mul_f32", f1, f1, ss),
                          isa.vop("v_mul_f32", f0, f0, g0), isa.vop("v_mul_f32", f1, f1, g1),
                          isa.vop("v_mul_f32", u0, F32(-1.4426950408889634), f0), isa.vop("v_mul_f32", u1, F32(-1.4426950408889634), f1),
                          isa.vop("v_exp_f32", u0, u0), isa.vop("v_exp_f32", u1, u1),
                          isa.vop("v_add_f32", u0, F32(1.0), u0), isa.vop("v_add_f32", u1, F32(1.0), u1),
                          isa.vop("v_rcp_f32", u0, u0), isa.vop("v_rcp_f32", u1, u1),
                          isa.vop("v_mul_f32", f0, f0, u0), isa.vop("v_mul_f32", f1, f1, u1),
                          isa.vop("v_cvt_pk_bf16_f32", src, f0, f1)]
            return r

        if with_resid:
            # residual rows: whole lines from memory (3 x 16 bytes per lane and row block, everything requested first), through LDS into the
            # accumulator layout
            for mb in range(8):
                e += row_sgprs(mb, ST[11], ST[7], ldr2)
                for i in range(3):
                    e += masked(ST[11], i, [isa.vop("v_add_u32", voff, ST[7], E_Z[i]),
                                            isa.global_load(4, LQ(mb, i), voff, 0, saddr=S_RF, extra_reads=[EXEC])])
            e.append(isa.waitcnt(vmcnt=0))
            for mb in range(9):
                if mb < 8:
                    e += [isa.ds_write(16, ER[i], LQ(mb, i)) for i in range(3)]
                    e += [isa.ds_read(8, RB(mb % 2, nb), EW, 32 * nb) for nb in range(6)]
                if mb >= 1:
                    e += to_packed(mb - 1, (mb - 1) % 2)
                    if c.epi == 6:
                        e += norm_silu(mb - 1)
        else:
            for mb in range(8):
                e += to_packed(mb, 0)
                if c.epi == 4:
                    e += norm_silu(mb)
        e += self.stamp(2)                                        # phase 2: accumulators -> packed outputs
        # the next tile's first loads have landed by now (the stores of the tile before were issued a whole tile ago): the accumulators can
        # take the next tile's bias as soon as the stores are issued -- these drain behind the next tile's taps
        e.append(isa.waitcnt(vmcnt=0))
        e += self.stamp(3)                                        # phase 3: wait for the next tile's first loads
        # row blocks through LDS into the memory layout: 3 stores of 64 x 16 contiguous bytes instead of 6 of 64 x 8 scattered ones
        def store_pass(base):
            r = []
            for mb in range(9):
                if mb < 8:
                    r += [isa.ds_write(8, EW, OUT(mb, nb), 32 * nb) for nb in range(6)]
                    r += [isa.ds_read_b128(RQ(mb % 2, i), ER[i]) for i in range(3)]
                if mb >= 1:
                    pm = mb - 1
                    r += row_sgprs(pm, ST[11], ST[7], ldc2)
                    for i in range(3):
                        st = isa.global_store(4, voff, RQ(pm % 2, i), 0, saddr=base, extra_reads=[EXEC])
                        if c.nt:
                            st.text += " nt"
                        r += masked(ST[11], i, [isa.vop("v_add_u32", voff, ST[7], E_Y[i]), st])
            return r

        e += store_pass(S_YF)
        if c.epi in (5, 7):
            # second output: the packed sums are still in OUT(., .): normalise them in place and store the row blocks again, y2 - y bytes further
            # (the residual frame base is free since the residual rows arrived)
            e += [isa.vop("v_readfirstlane_b32", ST[0], V(EPI_Y2D)), isa.vop("v_readfirstlane_b32", ST[1], V(EPI_Y2D + 1)),
                  isa.sop("s_add_u32", S_RF.sub(0), S_YF.sub(0), ST[0]), isa.sop("s_addc_u32", S_RF.sub(1), S_YF.sub(1), ST[1])]
            for mb in range(8):
                e += norm_silu(mb)
            e += store_pass(S_RF)
        e += self.stamp(4)                                        # phase 4: stores
        e += [isa.sop("s_bitcmp1_b32", None, self.HAVE_PREV, I32(1)), isa.branch("s_cbranch_scc1", "L_done" if c.prof else "L_exit"), isa.branch("s_branch", "L_start")]
        return sched.pad_hazards(sched.insert_lgkm_waits(e))

    def program(self) -> List[Instr]:
        # entry -> L_tile: setup + first DMAs (or straight to the epilogue when no tile follows) -> [first tile: L_first] / [else: L_epilogue -> L_start]
        # -> slices -> tile_end -> L_tile ...
        prog = (self.entry() + self.tile_setup() + self.epilogue() + self.tile_start() + self.slice_body() + self.tile_end()
                + [isa.label("L_done")] + self.prof_dump() + [isa.label("L_exit"), Instr("s_endpgm", cls=isa.BRANCH)])
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
\t\t.amdhsa_next_free_sgpr {N_SGPR}
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
# the 1x3x3 convolution of Resample (wan_vae.py:76-85: nearest-exact 2x upsample, then Conv2d(dim, dim / 2, 3, padding 1) per frame): its own
# code object (csrc/conv4u.s), so that csrc/conv4.s -- and the traffic measurements stamped with its blob id -- stay as they are
UPSAMPLE = [Cfg(epi=0, kt=1, name="scail_conv4u_e0")]
# narrow outputs (N <= 16): the decoder's RGB head (CausalConv3d(96, 3, 3), wan_vae.py:417-419); same code object as the kt = 1 kernel
NARROW = [Cfg(epi=0, nb=1, name="scail_conv4n_e0")]
# conv -> RMS_norm -> SiLU in one kernel (96 output channels): ResidualBlock.residual[2..4]; also in csrc/conv4u.s
FUSED = [Cfg(epi=4, name="scail_conv4f_e4")]
# tile continuation (Cfg.cont) of the three 96-channel kernels: A/B candidates beside the shipped ones, in csrc/conv4u.s
CONT = [Cfg(epi=0, cont=True, name="scail_conv4c_e0"), Cfg(epi=3, cont=True, name="scail_conv4c_e3"), Cfg(epi=4, cont=True, name="scail_conv4c_e4"),
        Cfg(epi=0, nb=1, cont=True, name="scail_conv4cn_e0")]      # the narrow kernel: no staging strip, the rings simply continue
# round 6: the last convolution of a 96-channel ResidualBlock with the NEXT consumer's RMS_norm + SiLU (Cfg.epi 5: raw sum + normalised copy; 6: normalised only)
RESNORM = [Cfg(epi=5, cont=True, name="scail_conv4c_e5"), Cfg(epi=6, cont=True, name="scail_conv4c_e6"), Cfg(epi=7, kt=1, name="scail_conv4u_e7")]


def variant_cfgs():
    out = []
    for abl in ("dma", "lds", "bar", "dma,lds", "patch"):
        out.append(Cfg(epi=0, abl=abl, name="scail_conv4_e0_abl_" + abl.replace(",", "_")))
    out.append(Cfg(epi=0, cap=2, name="scail_conv4_e0_c2"))
    out.append(Cfg(epi=0, prof=True, name="scail_conv4_e0_prof"))
    out.append(Cfg(epi=0, stagger=14, name="scail_conv4_e0_s14"))
    out.append(Cfg(epi=0, nt=False, name="scail_conv4_e0_t"))
    out.append(Cfg(epi=0, rd_step=2.0, name="scail_conv4_e0_rd2"))
    out.append(Cfg(epi=0, rd_at=6.0, rd_step=2.5, name="scail_conv4_e0_rd6"))
    out.append(Cfg(epi=0, p_at=30.0, dma_at=10.0, name="scail_conv4_e0_p30"))
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
