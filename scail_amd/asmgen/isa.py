"""A minimal gfx950 assembly DSL: registers, instructions with explicit read / write sets, text rendering.

Used by the kernel generators in this package (attn4.py) to emit hand-scheduled CDNA4 code: the generator builds lists of
``Instr`` objects, ``sched.py`` orders them around an MFMA spine and inserts the counted waits, and ``render`` prints the
``.s`` file that ``scail_amd/build.py`` assembles with clang (-x assembler, -mcpu=gfx950).  The same ``Instr`` objects are
executed by the CPU emulator of the test suite (tools/asm_emu.py), so a kernel is checked functionally before it reaches a
GPU.  Only the instruction subset the generators need is described here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union


@dataclass(frozen=True)
class Reg:
    """A register or an aligned register tuple: kind 'v' (arch VGPR), 'a' (AGPR), 's' (SGPR), or a special
    ('vcc', 'exec', 'm0', 'scc')."""
    kind: str
    idx: int = 0
    n: int = 1

    def __str__(self):
        if self.kind in ("vcc", "exec", "m0", "scc"):
            return self.kind
        return f"{self.kind}{self.idx}" if self.n == 1 else f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    def units(self) -> List[Tuple[str, int]]:
        if self.kind in ("vcc", "exec"):
            return [(self.kind, 0), (self.kind, 1)]
        if self.kind in ("m0", "scc"):
            return [(self.kind, 0)]
        return [(self.kind, self.idx + i) for i in range(self.n)]

    def sub(self, i: int, n: int = 1) -> "Reg":
        assert 0 <= i and i + n <= self.n
        return Reg(self.kind, self.idx + i, n)


def V(i, n=1): return Reg("v", i, n)
def A(i, n=1): return Reg("a", i, n)
def S(i, n=1): return Reg("s", i, n)


VCC, EXEC, M0, SCC = Reg("vcc"), Reg("exec"), Reg("m0"), Reg("scc")


@dataclass(frozen=True)
class Imm:
    """Literal / inline constant.  ``f`` floats render as hex bit patterns unless they are inline constants."""
    val: Union[int, float]
    is_float: bool = False

    def __str__(self):
        if self.is_float:
            v = float(self.val)
            if v in (0.0, 0.5, 1.0, 2.0, 4.0, -0.5, -1.0, -2.0, -4.0):
                return repr(v)
            import struct
            return hex(struct.unpack("<I", struct.pack("<f", v))[0])
        v = int(self.val)
        return str(v) if -16 <= v <= 64 else hex(v & 0xFFFFFFFF)


def I32(v): return Imm(int(v))
def F32(v): return Imm(float(v), True)


@dataclass(frozen=True)
class Neg:
    """Source modifier -x (VOP3 float sources)."""
    reg: Reg

    def __str__(self):
        return f"-{self.reg}"


Operand = Union[Reg, Imm, Neg]

# classes used by the scheduler / hazard checker
MFMA, VALU, TRANS, SALU, DS_READ, DS_WRITE, VMEM_LOAD, VMEM_STORE, LDS_DMA, WAIT, BARRIER, BRANCH, NOP, LABEL = (
    "mfma", "valu", "trans", "salu", "ds_read", "ds_write", "vmem_load", "vmem_store", "lds_dma", "wait", "barrier", "branch", "nop", "label")

_SALU_PREFIX = ("s_",)
_TRANS_OPS = {"v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32"}


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return MFMA
    if op in ("s_waitcnt",):
        return WAIT
    if op == "s_barrier":
        return BARRIER
    if op in ("s_nop",):
        return NOP
    if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64", "s_endpgm"):
        return BRANCH
    if op.startswith("s_"):
        return SALU
    if op.startswith("ds_read"):
        return DS_READ
    if op.startswith("ds_write"):
        return DS_WRITE
    if op.startswith("buffer_load") or op.startswith("global_load"):
        return VMEM_LOAD
    if op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("global_atomic"):
        return VMEM_STORE
    if op in _TRANS_OPS:
        return TRANS
    return VALU


@dataclass
class Instr:
    op: str
    dst: List[Reg] = field(default_factory=list)        # registers written
    src: List[Operand] = field(default_factory=list)    # operands read, in assembly order
    mods: str = ""                                       # trailing modifiers: "offset:16", "offen lds", "off" ...
    extra_reads: List[Reg] = field(default_factory=list)   # implicit reads (m0, vcc, exec, scc)
    extra_writes: List[Reg] = field(default_factory=list)  # implicit writes (vcc, scc)
    label: Optional[str] = None                          # for LABEL pseudo-instructions and branch targets
    comment: str = ""
    cls: str = ""
    tag: str = ""                                        # free-form tag the generators use (phase, role)
    offset: int = 0                                      # numeric copy of an "offset:" modifier for the emulator
    target_gap: float = 0.0                              # scheduling hint: preferred MFMA gap

    def __post_init__(self):
        if not self.cls:
            self.cls = LABEL if self.op == "label" else (LDS_DMA if "lds" in self.mods.split() else classify(self.op))

    # ---- register sets --------------------------------------------------------------------------
    def reads(self) -> List[Tuple[str, int]]:
        out = []
        for o in list(self.src) + list(self.extra_reads):
            r = o.reg if isinstance(o, Neg) else o
            if isinstance(r, Reg):
                out.extend(r.units())
        return out

    def writes(self) -> List[Tuple[str, int]]:
        out = []
        for r in list(self.dst) + list(self.extra_writes):
            out.extend(r.units())
        return out

    # ---- text ------------------------------------------------------------------------------------
    def render(self) -> str:
        if self.op == "label":
            if self.label is None:                       # an assembler directive riding as a pseudo-label (".p2align 6")
                return "\t" + self.text
            return f"{self.label}:"
        if getattr(self, "text", None):
            body = self.text
            if self.comment:
                body = f"{body:<76}// {self.comment}"
            return "\t" + body
        if self.op == "s_waitcnt" or self.op == "s_nop" or self.op == "s_setprio":
            body = f"{self.op} {self.mods}"
        elif self.cls == BRANCH and self.label is not None:
            body = f"{self.op} {self.label}"
        else:
            ops = [str(d) for d in self.dst] + [str(s) for s in self.src]
            if self.op.startswith("ds_write") or self.op.startswith("global_store") or self.op.startswith("buffer_store"):
                ops = [str(s) for s in self.src]
            body = self.op + (" " + ", ".join(ops) if ops else "")
            if self.mods:
                body += " " + self.mods
        if self.comment:
            body = f"{body:<76}// {self.comment}"
        return "\t" + body


def label(name: str) -> Instr:
    return Instr("label", label=name)


def render(instrs: Sequence[Instr]) -> str:
    return "\n".join(i.render() for i in instrs) + "\n"


# ------------------------------------------------------------------------------------------------
# constructors (one per instruction form the generators use; they fix operand order and implicit registers)
# ------------------------------------------------------------------------------------------------
_SCC_WRITERS = {"s_add_u32", "s_sub_u32", "s_add_i32", "s_sub_i32", "s_addc_u32", "s_and_b32", "s_or_b32", "s_xor_b32", "s_lshl_b32",
                "s_lshr_b32", "s_ashr_i32", "s_min_u32", "s_min_i32", "s_max_u32", "s_max_i32", "s_and_b64", "s_or_b64", "s_lshl_b64",
                "s_andn2_b64", "s_and_saveexec_b64", "s_bfe_u32"}
_SCC_READERS = {"s_cselect_b32", "s_addc_u32", "s_cselect_b64"}


def mfma(D: Reg, a: Reg, b: Reg, c, **kw) -> Instr:
    """D(16) = A(4: 32 rows x 16 k, bf16) x B(4: 16 k x 32 cols) + C(16 | 0)  -- v_mfma_f32_32x32x16_bf16."""
    assert D.n == 16 and a.n == 4 and b.n == 4
    return Instr("v_mfma_f32_32x32x16_bf16", [D], [a, b, c], **kw)


def mfma16(D: Reg, a: Reg, b: Reg, c, **kw) -> Instr:
    """D(4) = A(4: 16 rows x 32 k, bf16) x B(4: 32 k x 16 cols) + C(4 | 0)  -- v_mfma_f32_16x16x32_bf16 (4 passes = 16 cycles).
    Lane l holds A[l % 16][8 (l / 16) .. +7], B[8 (l / 16) .. +7][l % 16] and D[4 (l / 16) + e][l % 16], e = 0..3."""
    assert D.n == 4 and a.n == 4 and b.n == 4
    return Instr("v_mfma_f32_16x16x32_bf16", [D], [a, b, c], **kw)


def vop(op: str, dst, *srcs, **kw) -> Instr:
    d = [dst] if isinstance(dst, Reg) else list(dst)
    return Instr(op, d, list(srcs), **kw)


def permlane16_swap(a: Reg, b: Reg, **kw) -> Instr:
    """v_permlane16_swap_b32 a, b: the odd 16-lane rows of ``a`` (lanes 16-31, 48-63) are exchanged with the even rows of ``b``
    (lanes 0-15, 32-47); both read and written."""
    i = Instr("v_permlane16_swap_b32", [a, b], [a, b], **kw)
    i.text = f"v_permlane16_swap_b32 {a}, {b}"
    return i


def permlane32_swap(a: Reg, b: Reg, **kw) -> Instr:
    """v_permlane32_swap_b32 a, b: lanes 32-63 of ``a`` are exchanged with lanes 0-31 of ``b`` (both read and written)."""
    i = Instr("v_permlane32_swap_b32", [a, b], [a, b], **kw)
    i.text = f"v_permlane32_swap_b32 {a}, {b}"
    return i


def v_cmp(op: str, a, b, dst: Reg = VCC, **kw) -> Instr:
    return Instr(op, [dst], [a, b], **kw)


def v_cndmask(dst: Reg, a, b, mask: Reg = VCC, **kw) -> Instr:
    """dst = mask ? b : a (per lane)."""
    return Instr("v_cndmask_b32", [dst], [a, b, mask], **kw)


def sop(op: str, dst, *srcs, **kw) -> Instr:
    d = [] if dst is None else ([dst] if isinstance(dst, Reg) else list(dst))
    ew = [SCC] if (op in _SCC_WRITERS or op.startswith("s_cmp")) else []
    er = [SCC] if op in _SCC_READERS else []
    return Instr(op, d, list(srcs), extra_reads=er, extra_writes=ew, **kw)


def s_load(n: int, dst: Reg, base: Reg, off: int, **kw) -> Instr:
    op = {1: "s_load_dword", 2: "s_load_dwordx2", 4: "s_load_dwordx4", 8: "s_load_dwordx8"}[n]
    assert dst.n == n
    return Instr(op, [dst], [base, Imm(off)], offset=off, **kw)


def ds_read_b128(dst: Reg, addr: Reg, offset: int = 0, **kw) -> Instr:
    assert dst.n == 4 and 0 <= offset <= 65535
    return Instr("ds_read_b128", [dst], [addr], mods=f"offset:{offset}" if offset else "", offset=offset, **kw)


def ds_write(nbytes: int, addr: Reg, data: Reg, offset: int = 0, **kw) -> Instr:
    op = {4: "ds_write_b32", 8: "ds_write_b64", 16: "ds_write_b128"}[nbytes]
    assert data.n * 4 == nbytes and 0 <= offset <= 65535
    return Instr(op, [], [addr, data], mods=f"offset:{offset}" if offset else "", offset=offset, **kw)


def ds_read(nbytes: int, dst: Reg, addr: Reg, offset: int = 0, **kw) -> Instr:
    op = {4: "ds_read_b32", 8: "ds_read_b64", 16: "ds_read_b128"}[nbytes]
    assert dst.n * 4 == nbytes
    return Instr(op, [dst], [addr], mods=f"offset:{offset}" if offset else "", offset=offset, **kw)


def buffer_load_lds(voff: Reg, rsrc: Reg, soff, offset: int = 0, **kw) -> Instr:
    """LDS-DMA: 16 bytes per lane from buffer[voff + soff + offset] to LDS[m0 + offset + 16 lane]."""
    assert rsrc.n == 4 and 0 <= offset <= 4095
    return Instr("buffer_load_dwordx4", [], [voff, rsrc, soff], mods=f"offen offset:{offset} lds" if offset else "offen lds",
                 extra_reads=[M0], offset=offset, **kw)


def buffer_load(ndw: int, dst: Reg, voff: Reg, rsrc: Reg, soff, offset: int = 0, **kw) -> Instr:
    """dst(ndw) = buffer[voff + soff + offset]  (raw buffer, offen)."""
    op = {1: "buffer_load_dword", 2: "buffer_load_dwordx2", 4: "buffer_load_dwordx4"}[ndw]
    assert dst.n == ndw and rsrc.n == 4 and 0 <= offset <= 4095
    i = Instr(op, [dst], [voff, rsrc, soff], offset=offset, **kw)
    i.text = f"{op} {dst}, {voff}, {rsrc}, {soff} offen" + (f" offset:{offset}" if offset else "")
    return i


def global_load(ndw: int, dst: Reg, vaddr: Reg, offset: int = 0, saddr: Optional[Reg] = None, sc: bool = False, **kw) -> Instr:
    """vaddr(2) + offset, or -- with ``saddr`` -- saddr(2 SGPRs) + zero-extended vaddr(1) + offset.  sc: system-coherent read (sc0 sc1:
    served by L2, not by a possibly stale line of the CU's vector L1) -- for data this wave stored earlier in the kernel."""
    op = {1: "global_load_dword", 2: "global_load_dwordx2", 4: "global_load_dwordx4"}[ndw]
    assert dst.n == ndw and -4096 <= offset <= 4095 and vaddr.n == (1 if saddr is not None else 2)
    base = str(saddr) if saddr is not None else "off"
    i = Instr(op, [dst], [vaddr] + ([saddr] if saddr is not None else []), offset=offset, **kw)
    i.text = f"{op} {dst}, {vaddr}, {base}" + (f" offset:{offset}" if offset else "") + (" sc0 sc1" if sc else "")
    return i


def global_store(ndw: int, vaddr: Reg, data: Reg, offset: int = 0, saddr: Optional[Reg] = None, **kw) -> Instr:
    op = {1: "global_store_dword", 2: "global_store_dwordx2", 4: "global_store_dwordx4"}[ndw]
    assert data.n == ndw and vaddr.n == (1 if saddr is not None else 2)
    base = str(saddr) if saddr is not None else "off"
    i = Instr(op, [], [vaddr, data] + ([saddr] if saddr is not None else []), offset=offset, **kw)
    i.text = f"{op} {vaddr}, {data}, {base}" + (f" offset:{offset}" if offset else "")
    return i


def global_atomic_add(vaddr: Reg, data: Reg, saddr: Reg, offset: int = 0, **kw) -> Instr:
    """global_atomic_add (u32, no return value): mem[saddr + vaddr + offset] += data, per active lane."""
    assert data.n == 1 and vaddr.n == 1 and saddr.n == 2
    i = Instr("global_atomic_add", [], [vaddr, data, saddr], offset=offset, **kw)
    i.text = f"global_atomic_add {vaddr}, {data}, {saddr}" + (f" offset:{offset}" if offset else "")
    return i


def s_call(ret: Reg, target: str, **kw) -> Instr:
    """s_call_b64: ret(2) = address of the next instruction; jump to ``target`` (return with s_setpc_b64 ret)."""
    assert ret.n == 2
    i = Instr("s_call_b64", [ret], [], label=target, cls=BRANCH, **kw)
    i.text = f"s_call_b64 {ret}, {target}"
    return i


def waitcnt(vmcnt: Optional[int] = None, lgkmcnt: Optional[int] = None, **kw) -> Instr:
    parts = []
    if vmcnt is not None:
        parts.append(f"vmcnt({vmcnt})")
    if lgkmcnt is not None:
        parts.append(f"lgkmcnt({lgkmcnt})")
    i = Instr("s_waitcnt", mods=" ".join(parts), **kw)
    i.vmcnt, i.lgkmcnt = vmcnt, lgkmcnt
    return i


def nop(n: int = 0, **kw) -> Instr:
    assert 0 <= n <= 15
    i = Instr("s_nop", mods=str(n), **kw)
    i.count = n + 1
    return i


def barrier(**kw) -> Instr:
    return Instr("s_barrier", **kw)


def branch(op: str, target: str, **kw) -> Instr:
    er = {"s_cbranch_scc0": [SCC], "s_cbranch_scc1": [SCC], "s_cbranch_vccz": [VCC], "s_cbranch_vccnz": [VCC],
          "s_cbranch_execz": [EXEC], "s_cbranch_execnz": [EXEC]}.get(op, [])
    return Instr(op, label=target, extra_reads=er, **kw)
