"""Ordering of a straight-line gfx950 instruction block around an MFMA spine, counted LDS waits, and a static hazard check.

The generators write a block in PROGRAM ORDER (sequentially correct).  ``schedule`` re-orders it: the MFMAs keep their
relative order (the matrix pipe is the resource being saturated: one v_mfma_f32_32x32x16_bf16 occupies it for 32 cycles =
8 issue slots), every other instruction ("filler") is placed into the gaps between consecutive MFMAs, at most ``cap`` per gap
unless the dependencies of the next MFMA force more, as close as possible to its ``target_gap``.  All register dependencies
of the program order (RAW, WAW, WAR, explicit ``after`` edges) are preserved, with the minimum issue distances the hardware
needs but does not interlock (MI355X: MFMA result -> any non-accumulate reader, VALU -> MFMA operand, transcendental -> VALU,
VALU -> permlane, SALU m0 -> LDS-DMA ...); where a forced placement violates a distance an ``s_nop`` is inserted.
``insert_lgkm_waits`` then adds ``s_waitcnt lgkmcnt(n)`` in front of the first user of every ds_read result with the exact
in-order count.  ``check_hazards`` re-verifies any final sequence (also circularly, for loop bodies)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

from . import isa
from .isa import Instr

MFMA_RESULT_DIST = 16      # slots between an MFMA and a non-accumulate reader / overwriter of its destination
VALU_TO_MFMA_DIST = 3      # VALU (incl. v_accvgpr_write, v_cvt_pk) result -> MFMA operand: 2 wait states
TRANS_DIST = 2             # v_exp / v_rcp result -> non-transcendental VALU: 1 wait state
PERMLANE_DIST = 3          # VALU result -> v_permlane32_swap operand: 2 wait states
READLANE_DIST = 2          # VALU result -> v_readfirstlane / v_readlane source VGPR: 1 wait state, NOT interlocked (measured on MI355X in
                           # round 6: the lane read returned the register's PREVIOUS value -- the attention restart flag of wave 3 was
                           # lost this way, profiles/r06_attn_restart_readlane_hazard.log)
VALU_SGPR_TO_VALU_DIST = 3   # VALU-written SGPR / VCC (v_cmp, v_readfirstlane) -> VALU reading it as an operand (v_cndmask, v_mov ...): 2 wait states
                           # (the gfx940-family rule the vendor compiler pads with s_nop 1; same source as READLANE_DIST)
WIDE_STORE_WAR_DIST = 2    # VMEM store of more than 64 bits -> VALU overwriting its data registers: 1 wait state
M0_DIST = 2                # s_mov / s_add m0 -> LDS-DMA
VCC_BRANCH_DIST = 2        # v_cmp -> s_cbranch_vcc*
SGPR_VALU_TO_VMEM_DIST = 6   # VALU-written SGPR (readfirstlane) -> VMEM / SMEM use: 5 wait states
MFMA_SRC_WAR_DIST = 2      # overwrite of an MFMA's A / B operand registers after the MFMA
MFMA_SRCC_WAR_DIST = 12    # overwrite of an MFMA's srcC registers by a non-MFMA instruction (8 passes + 3 wait states, the larger shape's figure)


def _slots(i: Instr) -> int:
    return getattr(i, "count", 1)


def min_distance(p: Instr, c: Instr, kind: str, unit: Tuple[str, int]) -> int:
    """Minimum issue-slot distance from producer ``p`` to consumer ``c`` for a dependency of ``kind`` on register ``unit``."""
    if kind in ("raw", "waw"):
        if p.cls == isa.MFMA:
            if c.cls == isa.MFMA and kind == "raw":
                # accumulate chain (same 16-register tuple as srcC) is forwarded; as an A/B operand it is not
                c_src = c.src[2]
                if isinstance(c_src, isa.Reg) and unit in c_src.units() and c_src == p.dst[0]:
                    return 1
                return MFMA_RESULT_DIST
            if c.cls == isa.MFMA and kind == "waw":
                return 1 if c.dst[0] == p.dst[0] else MFMA_RESULT_DIST
            return MFMA_RESULT_DIST
        if kind == "raw":
            if c.cls == isa.MFMA and p.cls in (isa.VALU, isa.TRANS):
                return VALU_TO_MFMA_DIST
            if p.cls == isa.TRANS and c.cls in (isa.VALU, isa.DS_WRITE, isa.VMEM_STORE):
                return TRANS_DIST
            if c.op in ("v_permlane32_swap_b32", "v_permlane16_swap_b32") and p.cls in (isa.VALU, isa.TRANS):
                return PERMLANE_DIST
            if c.op in ("v_readfirstlane_b32", "v_readlane_b32") and unit[0] == "v" and p.cls in (isa.VALU, isa.TRANS):
                return max(READLANE_DIST, TRANS_DIST if p.cls == isa.TRANS else 0)
            if unit[0] == "m0" and c.cls == isa.LDS_DMA:
                return M0_DIST
            if unit[0] == "vcc" and c.cls == isa.BRANCH and p.cls in (isa.VALU, isa.TRANS):
                return VCC_BRANCH_DIST
            if unit[0] == "s" and p.cls in (isa.VALU,) and c.cls in (isa.VMEM_LOAD, isa.VMEM_STORE, isa.LDS_DMA, isa.SALU):
                return SGPR_VALU_TO_VMEM_DIST
            if unit[0] in ("s", "vcc") and p.cls in (isa.VALU, isa.TRANS) and c.cls in (isa.VALU, isa.TRANS, isa.MFMA):
                return VALU_SGPR_TO_VALU_DIST
        return 1
    # war
    if p.cls == isa.MFMA:
        # the accumulator input (srcC) is read by the matrix pipe after issue: a VALU overwrite of it must keep its distance (the vendor
        # compiler's rule for VGPR srcC: passes + 3 wait states; 4 passes for the 16x16x32 shape, 8 for 32x32x16)
        c_src = p.src[2] if len(p.src) > 2 else None
        if isinstance(c_src, isa.Reg) and unit in c_src.units() and c.cls != isa.MFMA:
            return MFMA_SRCC_WAR_DIST
        return MFMA_SRC_WAR_DIST
    if p.cls == isa.VMEM_STORE and c.cls in (isa.VALU, isa.TRANS) and unit[0] == "v" and ("dwordx4" in p.op or "dwordx3" in p.op):
        return WIDE_STORE_WAR_DIST
    return 1


def build_deps(block: Sequence[Instr]) -> List[Dict[int, int]]:
    """deps[i] = {j: min distance} over all j < i that instruction i must follow."""
    last_write: Dict[Tuple[str, int], int] = {}
    readers: Dict[Tuple[str, int], List[int]] = {}
    deps: List[Dict[int, int]] = []
    ids = {id(x): k for k, x in enumerate(block)}
    for i, ins in enumerate(block):
        d: Dict[int, int] = {}

        def add(j, dist):
            if j != i:
                d[j] = max(d.get(j, 0), dist)

        for u in ins.reads():
            if u in last_write:
                j = last_write[u]
                add(j, min_distance(block[j], ins, "raw", u))
        for u in ins.writes():
            if u in last_write:
                j = last_write[u]
                add(j, min_distance(block[j], ins, "waw", u))
            for j in readers.get(u, ()):
                add(j, min_distance(block[j], ins, "war", u))
        for other in getattr(ins, "after", ()):
            if id(other) in ids and ids[id(other)] < i:
                add(ids[id(other)], 1)
        deps.append(d)
        for u in ins.writes():
            last_write[u] = i
            readers[u] = []
        for u in ins.reads():
            readers.setdefault(u, []).append(i)
    return deps


def schedule(block: Sequence[Instr], cap: int = 5, lookahead: float = 1.0, trailing_cap: int = 0, late_extra: float = None) -> List[Instr]:
    """See module docstring.  ``target_gap`` of a filler = index of the MFMA after which it would like to sit (fractional
    values order fillers inside a gap); a filler is not pulled earlier than ``target_gap - lookahead`` gaps.
    ``late_extra``: a gap takes ONE filler beyond ``cap`` when the next candidate is already more than ``late_extra`` gaps behind
    its target -- a backlog is worked off one extra instruction per gap instead of piling up until a dependent MFMA forces the
    whole cluster into a single gap."""
    block = list(block)
    n = len(block)
    deps = build_deps(block)
    mf = [i for i in range(n) if block[i].cls == isa.MFMA]
    fillers = sorted((i for i in range(n) if block[i].cls != isa.MFMA), key=lambda i: (block[i].target_gap, i))
    pos: Dict[int, int] = {}
    out: List[Instr] = []
    cur = 0

    def slack(i):
        """slots still missing before i may issue (all deps must be emitted)."""
        need = 0
        for j, dist in deps[i].items():
            need = max(need, pos[j] + dist - cur)
        return need

    def emit(i):
        nonlocal cur
        miss = slack(i)
        while miss > 0:
            k = min(miss, 16)
            out.append(isa.nop(k - 1, comment="hazard pad"))
            cur += k
            miss -= k
        pos[i] = cur
        out.append(block[i])
        cur += _slots(block[i])

    def force(i):
        """emit i after all of its not-yet-emitted ancestors (program order among them)."""
        stack, order, seen = [i], [], set()
        while stack:
            x = stack.pop()
            if x in seen or x in pos:
                continue
            seen.add(x)
            order.append(x)
            stack.extend(j for j in deps[x] if j not in pos)
        for x in sorted(order):
            if x not in pos:
                emit(x)

    for g, m in enumerate(mf):
        force(m)
        placed = 0
        for i in fillers:
            if placed >= cap and not (late_extra is not None and placed == cap and block[i].target_gap <= g - late_extra):
                break
            if i in pos or block[i].target_gap > g + lookahead:
                continue
            if any(j not in pos for j in deps[i]) or slack(i) > 0:
                continue
            emit(i)
            placed += 1
    for i in range(n):
        if i not in pos:
            force(i)
    return out


def insert_lgkm_waits(seq: Sequence[Instr], carry_in: Sequence[Instr] = (), carry_out: List[Instr] = None) -> List[Instr]:
    """Counted ``s_waitcnt lgkmcnt(n)`` before the first instruction that touches the destination of an outstanding
    ds_read / s_load (LDS operations retire in order; n = number of younger LGKM operations allowed to stay in flight).
    ``carry_in``: LGKM operations still outstanding when the block is entered (issued by the preceding block, oldest first);
    ``carry_out`` (a list) receives the operations outstanding at the end of this block."""
    out: List[Instr] = []
    q: List[Instr] = list(carry_in)          # outstanding LGKM ops, oldest first
    for ins in seq:
        if ins.op == "s_waitcnt" and getattr(ins, "lgkmcnt", None) is not None:
            del q[:max(0, len(q) - ins.lgkmcnt)]
            out.append(ins)
            continue
        touched = set(ins.reads()) | set(ins.writes())
        need = None
        for k, o in enumerate(q):
            if touched & set(o.writes()):
                need = k
        if need is not None:
            n_keep = min(len(q) - 1 - need, 15)       # youngest operations that may stay in flight (field maximum 15)
            out.append(isa.waitcnt(lgkmcnt=n_keep))
            del q[:len(q) - n_keep]
        out.append(ins)
        if ins.cls in (isa.DS_READ, isa.DS_WRITE) or ins.op.startswith("s_load"):
            q.append(ins)
    if carry_out is not None:
        carry_out[:] = q
    return out


def check_hazards(seq: Sequence[Instr], circular: bool = False) -> List[str]:
    """Static re-check of the minimum distances on a final linear sequence (labels / waits are transparent).  With
    ``circular`` the sequence is checked as a loop body following itself.  Returns a list of violations (empty = clean)."""
    seq = [i for i in seq if i.op != "label"]
    rounds = 2 if circular else 1
    last_write: Dict[Tuple[str, int], Tuple[int, Instr]] = {}
    readers: Dict[Tuple[str, int], List[Tuple[int, Instr]]] = {}
    errs: List[str] = []
    cur = 0
    for rnd in range(rounds):
        for ins in seq:
            for u in ins.reads():
                if u in last_write:
                    p, pi = last_write[u]
                    need = min_distance(pi, ins, "raw", u)
                    if cur - p < need:
                        errs.append(f"RAW {u}: '{pi.render().strip()}' -> '{ins.render().strip()}' distance {cur - p} < {need}")
            for u in ins.writes():
                if u in last_write:
                    p, pi = last_write[u]
                    need = min_distance(pi, ins, "waw", u)
                    if cur - p < need:
                        errs.append(f"WAW {u}: '{pi.render().strip()}' -> '{ins.render().strip()}' distance {cur - p} < {need}")
                for p, pi in readers.get(u, ()):
                    need = min_distance(pi, ins, "war", u)
                    if cur - p < need:
                        errs.append(f"WAR {u}: '{pi.render().strip()}' -> '{ins.render().strip()}' distance {cur - p} < {need}")
            for u in ins.writes():
                last_write[u] = (cur, ins)
                readers[u] = []
            for u in ins.reads():
                readers.setdefault(u, []).append((cur, ins))
            cur += _slots(ins)
    return errs


def pad_hazards(seq: Sequence[Instr]) -> List[Instr]:
    """Sequential code (prologue, epilogue, rare paths): keep the order, insert ``s_nop`` where a minimum distance is not met."""
    out: List[Instr] = []
    block = [i for i in seq]
    deps = build_deps(block)
    pos: Dict[int, int] = {}
    cur = 0
    for i, ins in enumerate(block):
        miss = 0
        for j, dist in deps[i].items():
            miss = max(miss, pos[j] + dist - cur)
        while miss > 0:
            k = min(miss, 16)
            out.append(isa.nop(k - 1, comment="hazard pad"))
            cur += k
            miss -= k
        pos[i] = cur
        out.append(ins)
        cur += _slots(ins)
    return out
