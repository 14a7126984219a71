"""The hand-scheduled attn4 kernel, checked WITHOUT a GPU: the generated gfx950 instruction stream (scail_amd/asmgen/attn4.py,
the same objects that are printed into csrc/attn4.s) runs in the CPU emulator (tools/asm_emu.py: 4 waves, shared LDS,
in-order memory counters with lazy completion, MFMA result latency, real s_barrier) on small attention problems and is
compared with fp64 softmax(q k^T / sqrt(128)) v of the oracle formula (sat/transformer_defaults.py:67-72).  Covered: every
remainder path of the unrolled tile loop (1 ... 11 key tiles), both ring depths, ragged query blocks, several heads / batch
elements through the XCD-aware workgroup-id decode, key segments, the lazy-rescale subroutine (spiked keys, thr = 0), lazy and
eager completion of loads, plus the static hazard re-check of every scheduled block and `csrc/attn4.s` being up to date."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from scail_amd.asmgen import attn4  # noqa: E402
from tools import attn4_emu_run as R  # noqa: E402


def _rt(x):
    return R.from_bf16_bits(R.to_bf16_bits(x))


def _case(cfg, B, H, Lq, Lk, nseg=1, lazy=True, spike=False, thr=8.0, seed=0, mode=None, qscale=1.0, qshift=0.0):
    rng = np.random.default_rng(seed)
    q = (rng.standard_normal((B, Lq, H * 128)) * qscale + qshift).astype(np.float32)
    ks = [rng.standard_normal((B, Lk, H * 128)).astype(np.float32) for _ in range(nseg)]
    vs = [rng.standard_normal((B, Lk, H * 128)).astype(np.float32) for _ in range(nseg)]
    if spike:                                     # keys that dominate late: the running max must jump (guide rule 26)
        ks[-1][0, Lk - 3, :128] = q[0, 7, :128] * 3.0
        ks[0][0, 70 % Lk, :128] = q[0, 9, :128] * 2.0
    o, st = R.run(cfg, q, ks, vs, H, lazy=lazy, thr_log2=thr, mode=mode)
    if cfg.fold:                                  # the kernel sees q * scale * log2(e) rounded to bf16 (scail_rmsnorm_rope_scaled)
        c = np.float32((1.0 / np.sqrt(128.0)) * 1.4426950408889634)
        q = _rt(q * c) / c
    ref = R.reference(_rt(q), np.concatenate([_rt(x) for x in ks], 1), np.concatenate([_rt(x) for x in vs], 1), H)
    np.testing.assert_allclose(o, ref, rtol=2e-2, atol=6e-3)
    return st


def test_generated_file_is_current():
    text = attn4.assembly(attn4.SHIPPED)
    assert open(os.path.join(ROOT, "scail_amd", "csrc", "attn4.s")).read() == text, "run `python -m scail_amd.asmgen.attn4`"


def test_hazard_table_lane_read_valu_sgpr_and_wide_store_rules():
    """Rules the hardware does not interlock and the emulator (in-order, every result visible at once) cannot see.  Round 6, MI355X:
    v_or_b32 v6, v6, v11 ; v_readfirstlane_b32 s68, v6  returned v6's previous value -- the attention restart flag of wave 3 was lost
    (profiles/r06_attn_restart_readlane_hazard.log).  The static table pads and re-checks: VALU result -> lane read 1 wait state,
    VALU-written SGPR / VCC -> VALU operand 2, VMEM store > 64 bits -> VALU overwrite of its data 1."""
    from scail_amd.asmgen import isa, sched
    from scail_amd.asmgen.isa import V, S, I32
    seq = [isa.vop("v_or_b32", V(6), V(6), V(11)), isa.vop("v_readfirstlane_b32", S(68), V(6))]
    assert any("RAW" in e and "v_readfirstlane_b32" in e for e in sched.check_hazards(seq))
    padded = sched.pad_hazards(seq)
    assert [i.op for i in padded] == ["v_or_b32", "s_nop", "v_readfirstlane_b32"] and sched.check_hazards(padded) == []
    seq = [isa.v_cmp("v_cmp_eq_u32", V(1), I32(0)), isa.v_cndmask(V(2), V(3), V(4))]
    assert sched.check_hazards(seq)
    padded = sched.pad_hazards(seq)
    assert sched.check_hazards(padded) == [] and sum(getattr(i, "count", 1) for i in padded if i.op == "s_nop") == 2
    seq = [isa.vop("v_readfirstlane_b32", S(68), V(6)), isa.vop("v_mov_b32", V(7), S(68))]
    assert sched.check_hazards(seq) and sched.check_hazards(sched.pad_hazards(seq)) == []
    seq = [isa.global_store(4, V(20, 2), V(8, 4)), isa.vop("v_mov_b32", V(9), I32(0))]
    assert any("WAR" in e for e in sched.check_hazards(seq)) and sched.check_hazards(sched.pad_hazards(seq)) == []
    seq = [isa.global_store(2, V(20, 2), V(8, 2)), isa.vop("v_mov_b32", V(9), I32(0))]
    assert sched.check_hazards(seq) == []
    # an MFMA's accumulator INPUT is read after issue: a VALU overwrite keeps its distance (the folded maximum of the attention kernel is such an input)
    from scail_amd.asmgen.isa import A
    mf = isa.Instr("v_mfma_f32_16x16x32_bf16", [V(100, 4)], [A(240, 4), A(128, 4), V(212, 4)], cls=isa.MFMA)
    seq = [mf, isa.vop("v_mov_b32", V(213), I32(0))]
    assert any("WAR" in e for e in sched.check_hazards(seq)) and sched.check_hazards(sched.pad_hazards(seq)) == []


def test_emulator_checks_issue_distances_along_the_executed_path():
    """tools/asm_emu.py Emu.check_hazards (on for every emulator test: tests/conftest.py): the minimum issue distances of asmgen/sched.py are
    checked on the path each wave actually takes -- across labels, branches and subroutine calls.  Regression of round 6's GPU-only bug: the
    shipped kernel generated WITHOUT the lane-read rule computes the right numbers in the emulator (it executes in order), and the dynamic
    check names the instruction pair that lost wave 3's restart flag on the hardware."""
    from scail_amd.asmgen import sched
    rng = np.random.default_rng(21)
    q = rng.standard_normal((1, 256, 128)).astype(np.float32)
    k = rng.standard_normal((1, 64 * 5, 128)).astype(np.float32)
    v = rng.standard_normal((1, 64 * 5, 128)).astype(np.float32)
    keep = sched.READLANE_DIST
    try:
        sched.READLANE_DIST = 1
        prog = attn4.Gen(attn4.M16F).program()
    finally:
        sched.READLANE_DIST = keep
    with pytest.raises(AssertionError, match="v_readfirstlane_b32"):
        R.run(attn4.M16F, q, [k], [v], 1, program=prog)
    R.run(attn4.M16F, q, [k], [v], 1)                      # the shipped program: clean


@pytest.mark.parametrize("rd", [4, 2])
def test_static_hazards_clean(rd):
    assert R.check_static(attn4.Cfg(rd=rd)) == []


@pytest.mark.parametrize("tiles", [1, 2, 3, 4, 5, 6, 7, 9, 11])
def test_every_remainder_path_of_the_tile_loop(tiles):
    st = _case(attn4.DEFAULT, 1, 1, 256, 64 * tiles, lazy=bool(tiles & 1), seed=tiles)
    assert st["mfma"] == 64 * tiles                      # 32 QK^T + 32 P.V MFMAs per tile and wave, nothing recomputed


def test_ring_depth_two_variant():
    cfg = attn4.Cfg(rd=2)
    for tiles in (1, 2, 3, 5, 8):
        _case(cfg, 1, 1, 256, 64 * tiles, lazy=True, seed=10 + tiles)


def test_heads_batch_ragged_queries_and_workgroup_id_decode():
    _case(attn4.DEFAULT, 2, 4, 300, 128, seed=1)                    # 8 (batch, head) pairs: XCD-aware decode
    _case(attn4.DEFAULT, 1, 3, 520, 64, seed=2)                     # pairs % 8 != 0: plain decode; 3 query blocks, ragged last
    _case(attn4.DEFAULT, 2, 4, 256, 64, seed=3, mode=0)


def test_segments_and_lazy_rescale():
    _case(attn4.DEFAULT, 1, 2, 100, 192, nseg=3, spike=True, seed=4)
    _case(attn4.DEFAULT, 1, 1, 256, 640, lazy=False, spike=True, thr=0.0, seed=5)       # rescale whenever a row max moves
    _case(attn4.DEFAULT, 1, 1, 256, 640, lazy=True, spike=True, thr=2.0, seed=6)


# ---- the second shipped kernel: 16x16x32 MFMAs, q in log2 units, running maximum folded into the accumulator init (M16F) ----
def test_m16f_static_hazards_clean():
    assert R.check_static(attn4.M16F) == []


@pytest.mark.parametrize("tiles", [1, 2, 3, 4, 5, 6, 7, 9, 11])
def test_m16f_every_remainder_path(tiles):
    _case(attn4.M16F, 1, 1, 256, 64 * tiles, lazy=bool(tiles & 1), seed=tiles)


def test_m16f_heads_batch_ragged_queries_segments_and_rescale():
    _case(attn4.M16F, 2, 4, 300, 128, seed=1)
    _case(attn4.M16F, 1, 3, 520, 64, seed=2)
    _case(attn4.M16F, 1, 2, 100, 192, nseg=3, spike=True, seed=4)
    _case(attn4.M16F, 1, 1, 256, 640, lazy=False, spike=True, thr=0.0, seed=5)
    _case(attn4.M16F, 1, 1, 256, 640, lazy=True, spike=True, thr=2.0, seed=6)


def test_m16f_first_tile_sets_the_maximum_whatever_its_sign():
    """the running maximum starts at 0 and the first tile's (unconditional) subroutine call moves it to the tile's maximum: scores far
    below 0 must not underflow the whole row (q . k = -300 in log2 units here), scores far above must not overflow"""
    rng = np.random.default_rng(7)
    B, H, Lq, Lk = 1, 1, 64, 192
    k = rng.standard_normal((B, Lk, 128)).astype(np.float32) * 0.05 + 1.0           # all keys ~ +1 per component
    v = rng.standard_normal((B, Lk, 128)).astype(np.float32)
    for sign in (-1.0, 1.0):
        q = (sign * 18.0 + rng.standard_normal((B, Lq, 128)) * 0.05).astype(np.float32)     # q . k / sqrt(128) * log2 e ~ sign * 300
        o, _ = R.run(attn4.M16F, q, [k], [v], H)
        c = np.float32((1.0 / np.sqrt(128.0)) * 1.4426950408889634)
        ref = R.reference(_rt(q * c) / c, _rt(k), _rt(v), H)
        assert np.isfinite(o).all()
        np.testing.assert_allclose(o, ref, rtol=2e-2, atol=6e-3)


@pytest.mark.parametrize("Lk,nseg", [(513, 1), (575, 1), (64 * 9 + 1, 1), (64 * 12 + 63, 1), (100, 2), (64 * 6 + 32, 1), (129, 3)])
def test_m16f_ragged_key_counts(Lk, nseg):
    """any key count: the last tile of a segment holds 1..63 valid keys (513: a single one).  Their K rows are fetched from 64 rows
    earlier and their scores masked; the emulator's memory refuses any read outside an allocated buffer, so a fetch past the end
    of K would fail here even though its scores are masked."""
    _case(attn4.M16F, 1, 2 if Lk < 200 else 1, 70, Lk, nseg=nseg, spike=(Lk > 200), seed=Lk)


# ---- round 3: optimistic hot loop (fixed reference point, verified on the row sums, workgroup restart) and raw-scale callers ----
def _m16f_spiked(factor, row, key, tiles=11, seed=21, lazy=True):
    """One key, far from tile 0, aligned with query ``row``: its score exceeds the row's first-tile maximum by about
    factor * 128 / sqrt(128) * log2(e) = 16.3 * factor log2 units."""
    rng = np.random.default_rng(seed)
    Lq, Lk = 256, 64 * tiles
    q = rng.standard_normal((1, Lq, 128)).astype(np.float32)
    k = rng.standard_normal((1, Lk, 128)).astype(np.float32)
    v = rng.standard_normal((1, Lk, 128)).astype(np.float32)
    k[0, key] = q[0, row] * factor
    o, st = R.run(attn4.M16F, q, [k], [v], 1, lazy=lazy)
    c = np.float32((1.0 / np.sqrt(128.0)) * 1.4426950408889634)
    ref = R.reference(_rt(q * c) / c, _rt(k), _rt(v), 1)
    assert np.isfinite(o).all()
    np.testing.assert_allclose(o, ref, rtol=2e-2, atol=6e-3)
    return st, tiles


def test_m16f_optimistic_pass_needs_no_restart_inside_the_headroom():
    """a key 80 log2 units above the first tile's maximum (P up to 2^40 with 40 units of headroom): one pass, exact"""
    st, tiles = _m16f_spiked(5.0, row=7, key=64 * 2 + 5)
    assert st["mfma"] == 136 * tiles                     # 128 + 8 (row sums) MFMAs per tile and wave, nothing recomputed
    assert st["restarts"] == 0                           # the restart counter of the kernel arguments stays untouched


@pytest.mark.parametrize("row,key", [(7, 64 * 2 + 5), (150, 64 * 3 + 63), (255, 64 * 1)])
def test_m16f_optimistic_pass_overflow_restarts_the_workgroup(row, key):
    """a key ~196 log2 units above the first tile's maximum: exp2 overflows in the optimistic hot loop (no maximum tracking there),
    the row sum of that row becomes inf, the epilogue's check fires in ONE wave only and the whole workgroup (shared K / V^T rings)
    runs again with the lazy-maximum loop -- twice the MFMAs, the right result, nothing stored from the first pass"""
    st, tiles = _m16f_spiked(12.0, row=row, key=key, lazy=bool(row & 1))
    assert st["mfma"] == 2 * 136 * tiles
    assert st["restarts"] == 1                           # ONE count per restarting workgroup (wave 0, one lane), whichever wave saw the overflow


def test_m16f_restart_counter_is_optional():
    """a null counter pointer (every caller but the profiler): the restart path skips the atomic"""
    rng = np.random.default_rng(5)
    q = rng.standard_normal((1, 256, 128)).astype(np.float32)
    k = rng.standard_normal((1, 64 * 11, 128)).astype(np.float32)
    v = rng.standard_normal((1, 64 * 11, 128)).astype(np.float32)
    k[0, 64 * 4 + 1] = q[0, 9] * 12.0
    o, st = R.run(attn4.M16F, q, [k], [v], 1, count_restarts=False)
    o2, st2 = R.run(attn4.M16F, q, [k], [v], 1, count_restarts=True)
    assert st["restarts"] == 0 and st2["restarts"] == 1 and st["mfma"] == st2["mfma"] == 2 * 136 * 11
    assert np.array_equal(o, o2)


def test_m16f_restart_with_key_segments_and_ragged_tail():
    """the K / V^T descriptors walk over the segments: a restart must rewind them (3 segments of 11 tiles + a ragged tail)"""
    rng = np.random.default_rng(33)
    Lq, Lk, nseg = 100, 64 * 10 + 17, 3
    q = rng.standard_normal((1, Lq, 128)).astype(np.float32)
    ks = [rng.standard_normal((1, Lk, 128)).astype(np.float32) for _ in range(nseg)]
    vs = [rng.standard_normal((1, Lk, 128)).astype(np.float32) for _ in range(nseg)]
    ks[1][0, 64 * 2 + 9] = q[0, 3] * 12.0                  # overflow in the hot loop of the SECOND segment
    o, st = R.run(attn4.M16F, q, ks, vs, 1)
    c = np.float32((1.0 / np.sqrt(128.0)) * 1.4426950408889634)
    ref = R.reference(_rt(q * c) / c, np.concatenate([_rt(x) for x in ks], 1), np.concatenate([_rt(x) for x in vs], 1), 1)
    np.testing.assert_allclose(o, ref, rtol=2e-2, atol=6e-3)
    assert st["mfma"] == 2 * 136 * 11 * nseg


@pytest.mark.parametrize("Lk", [64 * 11, 513, 64 * 3])
def test_m16f_raw_scale_callers_scale_q_in_the_prologue(Lk):
    """sl2 != 0 (seam B3, sat/transformer_defaults.py:47-79: q arrives unscaled): the prologue multiplies the Q fragments by
    scale * log2(e) and rounds to bf16 once more -- same kernel, same loop; reference = attention of the queries the loop sees"""
    rng = np.random.default_rng(Lk)
    q = rng.standard_normal((2, 300, 2 * 128)).astype(np.float32)
    k = rng.standard_normal((2, Lk, 2 * 128)).astype(np.float32)
    v = rng.standard_normal((2, Lk, 2 * 128)).astype(np.float32)
    o, _ = R.run(attn4.M16F, q, [k], [v], 2, raw_scale=True)
    c = np.float32((1.0 / np.sqrt(128.0)) * 1.4426950408889634)
    ref = R.reference(_rt(_rt(q) * c) / c, _rt(k), _rt(v), 2)
    np.testing.assert_allclose(o, ref, rtol=2e-2, atol=6e-3)


# ---- round 5: the 192-row form of the shipped kernel (3 query blocks per wave) and the run-per-XCD workgroup-id decode (xcd_mode 2) ----
Q3 = attn4.M16F_Q3


def test_q3_static_hazards_clean():
    assert R.check_static(Q3) == []


@pytest.mark.parametrize("tiles", [1, 2, 3, 4, 5, 6, 7, 9, 11])
def test_q3_every_remainder_path(tiles):
    st = _case(Q3, 1, 1, 192, 64 * tiles, lazy=bool(tiles & 1), seed=tiles)
    assert st["mfma"] == 102 * tiles                     # 96 + 6 (row sums) MFMAs per tile and wave


def test_q3_heads_batch_ragged_queries_segments_and_rescale():
    _case(Q3, 2, 4, 300, 128, seed=1)                    # 8 pairs (round-robin decode), 2 query blocks of 192, the second ragged
    _case(Q3, 1, 3, 520, 64, seed=2)                     # 3 pairs x 3 query blocks = 9 items in runs of 2 (xcd_mode 2), ids 9..15 exit
    _case(Q3, 1, 2, 100, 192, nseg=3, spike=True, seed=4)
    _case(Q3, 1, 1, 192, 640, lazy=False, spike=True, thr=0.0, seed=5)
    _case(Q3, 1, 1, 192, 640, lazy=True, spike=True, thr=2.0, seed=6)


@pytest.mark.parametrize("Lk,nseg", [(513, 1), (64 * 9 + 1, 1), (64 * 12 + 63, 1), (100, 2), (129, 3)])
def test_q3_ragged_key_counts(Lk, nseg):
    _case(Q3, 1, 2 if Lk < 200 else 1, 70, Lk, nseg=nseg, spike=(Lk > 200), seed=Lk)


def test_q3_is_bit_identical_to_the_256_row_kernel():
    """a query row sees the same MFMA sequence over the keys whichever workgroup height computes it: the launch-shape choice of
    scail_flash_attn_bf16 never changes a result bit (incl. an optimistic-pass overflow + restart, which is per workgroup: row 150's
    spike restarts different sets of rows in the two tilings, both exact)"""
    rng = np.random.default_rng(50)
    B, H, Lq, Lk = 1, 2, 400, 64 * 9 + 5
    q = rng.standard_normal((B, Lq, H * 128)).astype(np.float32)
    k = rng.standard_normal((B, Lk, H * 128)).astype(np.float32)
    v = rng.standard_normal((B, Lk, H * 128)).astype(np.float32)
    o4, _ = R.run(attn4.M16F, q, [k], [v], H)
    o3, _ = R.run(Q3, q, [k], [v], H)
    assert np.array_equal(o4, o3)


def test_q3_overflow_restarts_the_workgroup_and_raw_scale():
    rng = np.random.default_rng(51)
    Lq, Lk = 192, 64 * 11
    q = rng.standard_normal((1, Lq, 128)).astype(np.float32)
    k = rng.standard_normal((1, Lk, 128)).astype(np.float32)
    v = rng.standard_normal((1, Lk, 128)).astype(np.float32)
    k[0, 64 * 3 + 63] = q[0, 150] * 12.0
    o, st = R.run(Q3, q, [k], [v], 1)
    c = np.float32((1.0 / np.sqrt(128.0)) * 1.4426950408889634)
    np.testing.assert_allclose(o, R.reference(_rt(q * c) / c, _rt(k), _rt(v), 1), rtol=2e-2, atol=6e-3)
    assert st["mfma"] == 2 * 102 * 11
    q2 = rng.standard_normal((2, 300, 2 * 128)).astype(np.float32)
    k2 = rng.standard_normal((2, 513, 2 * 128)).astype(np.float32)
    v2 = rng.standard_normal((2, 513, 2 * 128)).astype(np.float32)
    o, _ = R.run(Q3, q2, [k2], [v2], 2, raw_scale=True)
    np.testing.assert_allclose(o, R.reference(_rt(_rt(q2) * c) / c, _rt(k2), _rt(v2), 2), rtol=2e-2, atol=6e-3)


@pytest.mark.parametrize("B,H,Lq", [(1, 5, 700), (1, 3, 256), (3, 3, 300), (1, 1, 2100)])
def test_xcd_mode2_decode_covers_every_item_once(B, H, Lq):
    """xcd_mode 2: pair counts that are no multiple of 8 (5 heads = a Ulysses rank of 8) -- every (pair, query block) item is computed
    by exactly one workgroup (a missing item leaves zero rows, a wrong pair a wrong result), the padded ids exit"""
    assert attn4.xcd_mode(B, H) == 2
    _case(attn4.M16F, B, H, Lq, 64, seed=B * 100 + H)


@pytest.mark.parametrize("mode", [0, 2])
def test_split_attention_256_row_launch_then_192_row_launch(mode):
    """scail_flash_attn_bf16's mixed launch (csrc/attn.hip attn4_plan): items [0, a) of the 256-row tiling, then the remaining rows -- from a
    768-row boundary inside a pair on -- as 192-row tiles starting at item0 of THEIR tiling.  3 pairs x 1000 queries: the split falls
    inside pair 1 at row 768 (a = 4 + 3 tiles of 256; the 192-row launch starts at item 6 + 4 = 10 of 18).  Every row is computed exactly
    once (the output buffer starts as zeros) and equals the single-launch result bit for bit."""
    rng = np.random.default_rng(77)
    B, H, Lq, Lk = 1, 3, 1000, 64 * 3 + 9
    q = rng.standard_normal((B, Lq, H * 128)).astype(np.float32)
    k = rng.standard_normal((B, Lk, H * 128)).astype(np.float32)
    v = rng.standard_normal((B, Lk, H * 128)).astype(np.float32)
    whole, _ = R.run(attn4.M16F, q, [k], [v], H, mode=mode)
    nq4, nq3 = 4, 6                                   # ceil(1000 / 256), ceil(1000 / 192)
    a, item0 = 1 * nq4 + 3, 1 * nq3 + 4
    split, _ = R.run(attn4.M16F, q, [k], [v], H, launches=[(attn4.M16F, 0, a, mode), (Q3, item0, H * nq3 - item0, mode)])
    assert np.array_equal(split, whole)
    c = np.float32((1.0 / np.sqrt(128.0)) * 1.4426950408889634)
    np.testing.assert_allclose(split, R.reference(_rt(q * c) / c, _rt(k), _rt(v), H), rtol=2e-2, atol=6e-3)


# ---- round 5: cross attention over two key sets, persistent workgroups (Cfg.x2, scail_attn4_x2) ----
X2 = attn4.X2


def _x2_case(B, H, Lq, Lk1, Lk2, n_wgs, shared2=True, seed=0, lazy=True, spikes=(), raw_scale=False):
    """spikes: (set, batch, key, query row, factor) -- key <- factor x that query (a score ~16.3 x factor log2 units above the rest)"""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((B, Lq, H * 128)).astype(np.float32)
    k1 = rng.standard_normal((B, Lk1, H * 128)).astype(np.float32)
    v1 = rng.standard_normal((B, Lk1, H * 128)).astype(np.float32)
    B2 = 1 if shared2 else B
    k2 = rng.standard_normal((B2, Lk2, H * 128)).astype(np.float32)
    v2 = rng.standard_normal((B2, Lk2, H * 128)).astype(np.float32)
    for st, b, key, row, f in spikes:
        (k1 if st == 0 else k2)[b, key] = q[b, row] * f
    o, stats = R.run_x2(X2, q, k1, v1, k2, v2, H, n_wgs=n_wgs, lazy=lazy, raw_scale=raw_scale)
    c = np.float32((1.0 / np.sqrt(128.0)) * 1.4426950408889634)
    qq = _rt(_rt(q) * c) / c if raw_scale else _rt(q * c) / c
    ref = R.reference_x2(qq, _rt(k1), _rt(v1), _rt(k2), _rt(v2), H)
    assert np.isfinite(o).all()
    np.testing.assert_allclose(o, ref, rtol=2e-2, atol=8e-3)
    return stats


def test_x2_static_hazards_clean():
    assert R.check_static(X2) == []


def test_x2_generated_kernel_matches_two_softmaxes_summed():
    """dit_video_crossattn_sc_xc.py:1107-1203: bf16(bf16(softmax(q K_text^T) V_text) + softmax(q K_clip^T) V_clip); the shipped shape in
    small: 8 + 5 key tiles with the CLIP set's last tile holding ONE key (512 + 257 keys)"""
    st = _x2_case(1, 1, 256, 512, 257, n_wgs=1, seed=1)
    assert st[0]["mfma"] == 136 * 13


@pytest.mark.parametrize("n_wgs", [1, 2, 3])
def test_x2_persistent_workgroups_walk_over_the_items(n_wgs):
    """2 batch elements x 2 heads x 3 query blocks (ragged last: 600 rows) = 12 items over 1 / 2 / 3 workgroups: every item exactly
    once whatever the workgroup count (the output starts as zeros), per-item reload of the kernel arguments, the CLIP set shared by
    the batch (stride 0) like the engine passes it"""
    st = _x2_case(2, 2, 600, 128, 65, n_wgs=n_wgs, seed=2 + n_wgs, lazy=bool(n_wgs & 1))
    assert sum(x["mfma"] for x in st) == 12 * 136 * (2 + 2)


def test_x2_per_batch_second_set_and_ragged_first_set():
    _x2_case(2, 1, 100, 64 * 3 + 9, 64 * 2, n_wgs=2, shared2=False, seed=9)
    _x2_case(1, 3, 256, 64, 64 * 6 + 63, n_wgs=2, seed=10)            # one-tile first set, long ragged second set


@pytest.mark.parametrize("which", [0, 1])
def test_x2_optimistic_overflow_restarts_only_that_set(which):
    """a key ~196 log2 units above its row's first-tile maximum in set `which` (far from tile 0): that set's pass runs again with the
    lazy-maximum loop, the other set's pass runs once; set 0's stored output is read back unchanged either way"""
    st = _x2_case(1, 1, 256, 64 * 11, 64 * 12, n_wgs=1, seed=20 + which, spikes=[(which, 0, 64 * 3 + 7, 150, 12.0)])
    assert st[0]["mfma"] == 136 * (11 + 12) + 136 * (11 if which == 0 else 12)       # (the optimistic hot loop needs >= 10 tiles: shorter sets track the maximum)


def test_x2_raw_scale_callers():
    _x2_case(1, 2, 300, 128, 70, n_wgs=2, seed=30, raw_scale=True)
