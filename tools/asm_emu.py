"""CPU emulator for the gfx950 instruction subset of scail_amd/asmgen (TEST INFRASTRUCTURE: only tests/ and tools/ use it).

It executes the very ``Instr`` objects a generator emits -- one workgroup, all its waves, round-robin one instruction at
a time, with a shared LDS and a flat "global memory" arena -- so a hand-scheduled kernel is checked functionally on the CPU
before it ever reaches a GPU.  Beyond plain functional execution it models the parts of the machine a hand-written
stream gets wrong silently:

  * memory counters: ``ds_read`` / ``s_load`` results and LDS-DMA / global loads are NOT visible when issued.  In
    ``lazy`` mode (default) they are committed as late as the architecture allows -- only when an ``s_waitcnt`` with a small
    enough count retires them (in order) -- so a missing or too-large wait shows up as a wrong result; ``eager`` mode commits
    at once (catches the opposite mistake: data landing early over something still in use).  Run a kernel in both.
  * MFMA results are committed ``MFMA_LATENCY`` issue slots after the instruction (more than the hardware's 12 wait states
    for an 8-pass v_mfma_f32_32x32x16_bf16), except towards an MFMA that takes the register as its accumulator (srcC
    chain), which the hardware forwards.  A VALU / LDS / store that reads the destination earlier sees the OLD value.
  * ``s_barrier`` really synchronises the waves of the workgroup (a wave that skips a barrier dead-locks the emulator).

Numerics: fp32 IEEE through numpy; bf16 conversions round to nearest even; v_exp_f32 = exp2; v_rcp_f32 = 1/x."""
from __future__ import annotations

import struct
from typing import Dict, List, Sequence

import os

import numpy as np

from scail_amd.asmgen import isa

MFMA_LATENCY = 18
ARENA_BASE = 0x7F0000000000


def _f(u):
    return u.view(np.float32)


def _u(f):
    return np.asarray(f, dtype=np.float32).view(np.uint32)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bits (uint32 holding 16 bits), round to nearest even."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (r & 0xFFFF).astype(np.uint32)


def bf16_to_f32(h: np.ndarray) -> np.ndarray:
    return (h.astype(np.uint32) << 16).view(np.float32)


_SCHED = None


def _sched():
    global _SCHED
    if _SCHED is None:
        from scail_amd.asmgen import sched as m
        _SCHED = m
    return _SCHED


class Memory:
    """Flat global memory: named numpy buffers placed in one arena; pointers are ARENA_BASE + offset."""

    def __init__(self, size=1 << 28):
        self.mem = np.zeros(size, dtype=np.uint8)
        self.top = 4096
        self.bufs: Dict[str, tuple] = {}

    def alloc(self, name: str, arr: np.ndarray, pad: int = 0) -> int:
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        off = (self.top + 255) // 256 * 256
        assert off + raw.size + pad <= self.mem.size, "arena too small"
        self.mem[off:off + raw.size] = raw
        self.top = off + raw.size + pad
        self.bufs[name] = (off, raw.size, arr.dtype, arr.shape)
        return ARENA_BASE + off

    def read_back(self, name: str) -> np.ndarray:
        off, n, dt, shape = self.bufs[name]
        return self.mem[off:off + n].view(dt).reshape(shape).copy()

    def _chk(self, addr, n):
        off = addr - ARENA_BASE
        if off < 0 or off + n > self.top:
            raise RuntimeError(f"global access out of the arena: 0x{addr:x} (+{n})")
        # every access must lie inside ONE allocated buffer: a kernel that reads past the end of an operand (into the alignment gap
        # or into its neighbour) is caught here even when the values are masked later
        for lo, size, _, _ in self.bufs.values():
            if lo <= off and off + n <= lo + size:
                return off
        raise RuntimeError(f"global access outside every allocated buffer: arena offset {off} (+{n})")

    def load(self, addr: int, n: int) -> np.ndarray:
        off = self._chk(addr, n)
        return self.mem[off:off + n]

    def store(self, addr: int, data: np.ndarray):
        off = self._chk(addr, data.size)
        self.mem[off:off + data.size] = data


class Wave:
    def __init__(self, wid: int):
        self.id = wid
        self.v = np.zeros((512, 64), dtype=np.uint32)       # arch VGPRs
        self.a = np.zeros((512, 64), dtype=np.uint32)       # AGPRs
        self.s = np.zeros(128, dtype=np.uint32)
        self.vcc = np.zeros(64, dtype=bool)
        self.exec = np.ones(64, dtype=bool)
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.lgkm: List = []        # pending (commit_fn)
        self.vm: List = []
        self.mfma_pending: List = []    # [slots_left, reg, values]
        self.issued = 0
        self.hz_last_write = {}     # (Emu.check_hazards) register unit -> (issue slot, instruction) of its last writer
        self.hz_readers = {}        # ... -> the last few readers since that write
        self.stats = {"mfma": 0, "instr": 0}


class Emu:
    def __init__(self, program: Sequence[isa.Instr], mem: Memory, n_waves: int = 4, lds_bytes: int = 160 * 1024, lazy: bool = True,
                 check_hazards: bool = None):
        # DYNAMIC hazard check (round 6): every executed instruction is checked against the minimum issue distances of asmgen/sched.py
        # (min_distance: the same table the static padding uses) along the path the wave ACTUALLY took -- across labels, branches and
        # subroutine calls, where the static check of a linear block cannot look.  Violations are collected in ``self.hazards``; the emulator
        # itself executes in order with every result visible at once, so it would not notice them in the numbers (that is how the lost restart
        # flag of round 6 passed every CPU test).  Default: on when SCAIL_EMU_HAZARDS=1 (the emulator tests switch it on for their rare-path cases).
        self.check_hazards = (os.environ.get("SCAIL_EMU_HAZARDS", "0") != "0") if check_hazards is None else check_hazards
        self.hazards: List[str] = []
        self.prog = list(program)
        self.labels = {i.label: k for k, i in enumerate(self.prog) if i.op == "label"}
        self.mem = mem
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.waves = [Wave(w) for w in range(n_waves)]
        self.lazy = lazy
        self.label_addr = {}          # label -> fake byte address for s_getpc / s_setpc based calls

    # ---------------------------------------------------------------------------------------------
    def launch(self, kernarg: bytes, block_id=(0, 0, 0), max_steps: int = 50_000_000):
        ka = self.mem.alloc("__kernarg", np.frombuffer(kernarg, dtype=np.uint8))
        for w in self.waves:
            w.s[0], w.s[1] = ka & 0xFFFFFFFF, ka >> 32
            w.s[2], w.s[3], w.s[4] = block_id
            w.v[0] = np.arange(64, dtype=np.uint32) + 64 * w.id
        steps = 0
        while not all(w.done for w in self.waves):
            progressed = False
            for w in self.waves:
                if w.done:
                    continue
                if w.at_barrier:
                    continue
                self.step(w)
                progressed = True
                steps += 1
            if all(w.done or w.at_barrier for w in self.waves):
                live = [w for w in self.waves if not w.done]
                if live and all(w.at_barrier for w in live):
                    if len(live) != len(self.waves):
                        raise RuntimeError("barrier dead-lock: some waves ended while others wait at s_barrier")
                    for w in live:
                        w.at_barrier = False
                    progressed = True
            if not progressed:
                raise RuntimeError("emulator dead-lock")
            if steps > max_steps:
                raise RuntimeError("emulator step limit")
        if self.check_hazards and self.hazards:
            uniq = sorted(set(h.split(" ", 2)[2] for h in self.hazards))
            raise AssertionError(f"dynamic hazard check: {len(self.hazards)} violation(s) of the minimum issue distances, e.g. " + " | ".join(uniq[:4]))
        return steps

    # ---------------------------------------------------------------------------------------------
    def _rd(self, w: Wave, o, as_f=False):
        """operand -> 64-lane uint32 vector (scalars broadcast)."""
        neg = False
        if isinstance(o, isa.Neg):
            neg, o = True, o.reg
        if isinstance(o, isa.Imm):
            val = np.full(64, struct.unpack("<I", struct.pack("<f", float(o.val)))[0] if o.is_float else int(o.val) & 0xFFFFFFFF, dtype=np.uint32)
        elif o.kind == "v":
            val = w.v[o.idx].copy()
        elif o.kind == "a":
            val = w.a[o.idx].copy()
        elif o.kind == "s":
            val = np.full(64, w.s[o.idx], dtype=np.uint32)
        elif o.kind == "m0":
            val = np.full(64, w.m0, dtype=np.uint32)
        else:
            raise NotImplementedError(o)
        if neg:
            val = val ^ np.uint32(0x80000000)
        return val

    def _rds(self, w: Wave, o) -> int:
        if isinstance(o, isa.Imm):
            return struct.unpack("<I", struct.pack("<f", float(o.val)))[0] if o.is_float else int(o.val) & 0xFFFFFFFF
        if o.kind == "s":
            return int(w.s[o.idx])
        if o.kind == "m0":
            return int(w.m0)
        if o.kind == "vcc":
            return int(sum(1 << i for i in range(32) if w.vcc[i]))
        raise NotImplementedError(o)

    def _rds64(self, w: Wave, o) -> int:
        if isinstance(o, isa.Imm):
            return int(o.val) & 0xFFFFFFFFFFFFFFFF
        if o.kind == "s":
            return int(w.s[o.idx]) | (int(w.s[o.idx + 1]) << 32)
        if o.kind == "vcc":
            return int(sum(1 << i for i in range(64) if w.vcc[i]))
        if o.kind == "exec":
            return int(sum(1 << i for i in range(64) if w.exec[i]))
        raise NotImplementedError(o)

    def _wrs(self, w: Wave, r: isa.Reg, val: int):
        val &= 0xFFFFFFFF
        if r.kind == "s":
            w.s[r.idx] = val
        elif r.kind == "m0":
            w.m0 = val
        else:
            raise NotImplementedError(r)

    def _wrs64(self, w: Wave, r: isa.Reg, val: int):
        if r.kind == "s":
            w.s[r.idx], w.s[r.idx + 1] = val & 0xFFFFFFFF, (val >> 32) & 0xFFFFFFFF
        elif r.kind == "vcc":
            w.vcc = np.array([(val >> i) & 1 for i in range(64)], dtype=bool)
        elif r.kind == "exec":
            w.exec = np.array([(val >> i) & 1 for i in range(64)], dtype=bool)
        else:
            raise NotImplementedError(r)

    def _wrv(self, w: Wave, r: isa.Reg, val: np.ndarray, lane_mask=None):
        bank = w.v if r.kind == "v" else w.a
        m = w.exec if lane_mask is None else lane_mask
        bank[r.idx][m] = np.asarray(val, dtype=np.uint32)[m]

    def _regs(self, w: Wave, r: isa.Reg) -> np.ndarray:
        bank = w.v if r.kind == "v" else w.a
        return bank[r.idx:r.idx + r.n]

    # ---------------------------------------------------------------------------------------------
    def _retire_mfma(self, w: Wave, slots: int):
        keep = []
        for p in w.mfma_pending:
            p[0] -= slots
            if p[0] <= 0:
                self._regs(w, p[1])[:] = p[2]
            else:
                keep.append(p)
        w.mfma_pending = keep

    def _flush_mfma_for(self, w: Wave, reg: isa.Reg):
        """srcC forwarding: commit pending results overlapping ``reg``."""
        keep = []
        for p in w.mfma_pending:
            if p[1].kind == reg.kind and not (p[1].idx + p[1].n <= reg.idx or reg.idx + reg.n <= p[1].idx):
                self._regs(w, p[1])[:] = p[2]
            else:
                keep.append(p)
        w.mfma_pending = keep

    def _hazard_check(self, w: Wave, ins: isa.Instr):
        sched = _sched()
        cur = w.issued
        lw, rd = w.hz_last_write, w.hz_readers
        hz = getattr(ins, "_hz_units", None)
        if hz is None:                       # the unit lists of an instruction object never change: computed once
            hz = ins._hz_units = (tuple(ins.reads()), tuple(ins.writes()))
        reads, writes = hz
        # a (writer, reader) pair sits at a fixed distance on a straight path: re-checking it on every trip of a hot loop adds nothing, so a pair
        # is looked at only when the distance is short enough to matter (every rule of the table is <= 16 slots)
        if len(self.hazards) < 50:
            for u in reads:
                p = lw.get(u)
                if p is not None and cur - p[0] < 16:
                    need = sched.min_distance(p[1], ins, "raw", u)
                    if cur - p[0] < need:
                        self.hazards.append(f"wave {w.id} RAW {u}: '{p[1].render().strip()}' -> '{ins.render().strip()}' distance {cur - p[0]} < {need}")
            for u in writes:
                p = lw.get(u)
                if p is not None and cur - p[0] < 16:
                    need = sched.min_distance(p[1], ins, "waw", u)
                    if cur - p[0] < need:
                        self.hazards.append(f"wave {w.id} WAW {u}: '{p[1].render().strip()}' -> '{ins.render().strip()}' distance {cur - p[0]} < {need}")
                for q in rd.get(u, ()):
                    if cur - q[0] >= 16:
                        continue
                    need = sched.min_distance(q[1], ins, "war", u)
                    if need > 1 and cur - q[0] < need:
                        self.hazards.append(f"wave {w.id} WAR {u}: '{q[1].render().strip()}' -> '{ins.render().strip()}' distance {cur - q[0]} < {need}")
        for u in writes:
            lw[u] = (cur, ins)
            rd[u] = []
        for u in reads:
            lst = rd.setdefault(u, [])
            lst.append((cur, ins))
            if len(lst) > 4:
                del lst[0]

    def _queue(self, w: Wave, q: str, fn):
        if self.lazy:
            getattr(w, q).append(fn)
        else:
            fn()

    def _drain(self, w: Wave, q: str, n: int):
        lst = getattr(w, q)
        while len(lst) > n:
            lst.pop(0)()

    # ---------------------------------------------------------------------------------------------
    def step(self, w: Wave):
        ins = self.prog[w.pc]
        w.pc += 1
        op = ins.op
        if op == "label":
            return
        slots = getattr(ins, "count", 1)
        if self.check_hazards:
            self._hazard_check(w, ins)
        w.issued += slots
        w.stats["instr"] += 1
        d = ins.dst[0] if ins.dst else None
        s = ins.src
        R = lambda k: self._rd(w, s[k])
        F = lambda k: _f(self._rd(w, s[k]))

        if op == "v_mfma_f32_32x32x16_bf16":
            self._retire_mfma(w, 1)
            if isinstance(s[2], isa.Reg):
                self._flush_mfma_for(w, s[2])
            A = self._regs(w, s[0])       # (4, 64) uint32
            B = self._regs(w, s[1])
            lane = np.arange(64)
            Am = np.zeros((32, 16), dtype=np.float32)
            Bm = np.zeros((16, 32), dtype=np.float32)
            for j in range(8):
                av = (A[j >> 1] >> (16 * (j & 1))) & 0xFFFF
                bv = (B[j >> 1] >> (16 * (j & 1))) & 0xFFFF
                Am[lane & 31, 8 * (lane >> 5) + j] = bf16_to_f32(av)
                Bm[8 * (lane >> 5) + j, lane & 31] = bf16_to_f32(bv)
            P = Am.astype(np.float64) @ Bm.astype(np.float64)
            if isinstance(s[2], isa.Reg):
                C = _f(self._regs(w, s[2]).copy())
            else:
                C = np.zeros((16, 64), dtype=np.float32)
            out = np.zeros((16, 64), dtype=np.float32)
            for r in range(16):
                rows = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                out[r] = (C[r].astype(np.float64) + P[rows, lane & 31]).astype(np.float32)
            w.mfma_pending.append([MFMA_LATENCY, d, out.view(np.uint32).copy()])
            w.stats["mfma"] += 1
            return
        if op == "v_mfma_f32_16x16x32_bf16":
            self._retire_mfma(w, 1)
            if isinstance(s[2], isa.Reg):
                self._flush_mfma_for(w, s[2])
            A = self._regs(w, s[0])       # (4, 64) uint32
            B = self._regs(w, s[1])
            lane = np.arange(64)
            Am = np.zeros((16, 32), dtype=np.float32)
            Bm = np.zeros((32, 16), dtype=np.float32)
            for j in range(8):
                av = (A[j >> 1] >> (16 * (j & 1))) & 0xFFFF
                bv = (B[j >> 1] >> (16 * (j & 1))) & 0xFFFF
                Am[lane & 15, 8 * (lane >> 4) + j] = bf16_to_f32(av)
                Bm[8 * (lane >> 4) + j, lane & 15] = bf16_to_f32(bv)
            P = Am.astype(np.float64) @ Bm.astype(np.float64)
            C = _f(self._regs(w, s[2]).copy()) if isinstance(s[2], isa.Reg) else np.zeros((4, 64), dtype=np.float32)
            out = np.zeros((4, 64), dtype=np.float32)
            for r in range(4):
                out[r] = (C[r].astype(np.float64) + P[4 * (lane >> 4) + r, lane & 15]).astype(np.float32)
            w.mfma_pending.append([MFMA_LATENCY // 2, d, out.view(np.uint32).copy()])
            w.stats["mfma"] += 1
            return
        self._retire_mfma(w, slots)

        # ---- waits / control ----------------------------------------------------------------------
        if op == "s_waitcnt":
            if getattr(ins, "vmcnt", None) is not None:
                self._drain(w, "vm", ins.vmcnt)
            if getattr(ins, "lgkmcnt", None) is not None:
                self._drain(w, "lgkm", ins.lgkmcnt)
            return
        if op in ("s_nop", "s_setprio", "s_sleep"):
            return
        if op == "s_barrier":
            w.at_barrier = True
            return
        if op == "s_endpgm":
            self._drain(w, "vm", 0)
            self._drain(w, "lgkm", 0)
            self._retire_mfma(w, 1000)
            w.done = True
            return
        if op == "s_branch":
            w.pc = self.labels[ins.label]
            return
        if op in ("s_cbranch_scc0", "s_cbranch_scc1"):
            if w.scc == (1 if op.endswith("1") else 0):
                w.pc = self.labels[ins.label]
            return
        if op in ("s_cbranch_vccnz", "s_cbranch_vccz"):
            nz = bool(w.vcc.any())
            if nz == op.endswith("nz"):
                w.pc = self.labels[ins.label]
            return
        if op in ("s_cbranch_execz", "s_cbranch_execnz"):
            nz = bool(w.exec.any())
            if nz == op.endswith("nz"):
                w.pc = self.labels[ins.label]
            return
        if op == "s_call_b64":
            self._wrs64(w, ins.dst[0], w.pc)
            w.pc = self.labels[ins.label]
            return
        if op == "s_setpc_b64":
            w.pc = self._rds64(w, s[0])
            return

        # ---- SALU ---------------------------------------------------------------------------------
        if op.startswith("s_"):
            self._salu(w, ins)
            return

        # ---- LDS ----------------------------------------------------------------------------------
        if op.startswith("ds_read"):
            nb = d.n * 4
            addr = (self._rd(w, s[0]).astype(np.int64) + ins.offset)
            if (addr + nb > self.lds.size).any():
                raise RuntimeError(f"LDS read out of range at pc {w.pc - 1}: max {int(addr.max())}")
            # the data is sampled at ISSUE time in eager mode; in lazy mode at retire time (worst case for a racing writer)
            def commit(addr=addr, d=d, nb=nb, mask=w.exec.copy()):
                data = np.stack([self.lds[a:a + nb] for a in addr]).view(np.uint32)      # (64, n)
                bank = w.v if d.kind == "v" else w.a
                for i in range(d.n):
                    bank[d.idx + i][mask] = data[:, i][mask]
            self._queue(w, "lgkm", commit)
            return
        if op.startswith("ds_write"):
            # LDS instructions of one wave execute in order: the wave's older ds_reads sample their data before this write lands
            self._drain(w, "lgkm", 0)
            data = self._regs(w, s[1])
            nb = s[1].n * 4
            addr = (self._rd(w, s[0]).astype(np.int64) + ins.offset)
            if (addr + nb > self.lds.size).any():
                raise RuntimeError("LDS write out of range")
            vals = np.ascontiguousarray(data.T).view(np.uint8).reshape(64, nb)
            for l in range(64):
                if w.exec[l]:
                    self.lds[addr[l]:addr[l] + nb] = vals[l]
            self._queue(w, "lgkm", lambda: None)
            return

        # ---- VMEM ---------------------------------------------------------------------------------
        if op.startswith("buffer_load") and ins.cls == isa.LDS_DMA:
            rs = self._regs_s(w, s[1])
            base = rs[0] | ((rs[1] & 0xFFFF) << 32)
            nrec = rs[2]
            voff = self._rd(w, s[0]).astype(np.int64)
            soff = self._rds(w, s[2])
            lds_addr = w.m0 + ins.offset + 16 * np.arange(64)
            gaddr = base + voff + soff + ins.offset
            inrange = (voff + ins.offset + 16) <= nrec
            def commit(gaddr=gaddr, lds_addr=lds_addr, inrange=inrange):
                for l in range(64):
                    self.lds[lds_addr[l]:lds_addr[l] + 16] = self.mem.load(int(gaddr[l]), 16) if inrange[l] else 0
            self._queue(w, "vm", commit)
            return
        if op.startswith("buffer_load"):
            nb = d.n * 4
            rs = self._regs_s(w, s[1])
            base = rs[0] | ((rs[1] & 0xFFFF) << 32)
            voff = self._rd(w, s[0]).astype(np.int64)
            addr = base + voff + self._rds(w, s[2]) + ins.offset
            inrange = (voff + ins.offset + nb) <= rs[2]            # raw buffer: offsets past num_records read zeros
            def commit(addr=addr, d=d, nb=nb, mask=w.exec.copy(), inrange=inrange):
                bank = w.v if d.kind == "v" else w.a
                for l in range(64):
                    if mask[l]:
                        data = self.mem.load(int(addr[l]), nb).view(np.uint32) if inrange[l] else np.zeros(nb // 4, np.uint32)
                        for i in range(d.n):
                            bank[d.idx + i][l] = data[i]
            self._queue(w, "vm", commit)
            return
        if op.startswith("global_load"):
            nb = d.n * 4
            addr = self._gaddr(w, s[0], s[1] if len(s) > 1 else None) + ins.offset
            def commit(addr=addr, d=d, nb=nb, mask=w.exec.copy()):
                bank = w.v if d.kind == "v" else w.a
                for l in range(64):
                    if mask[l]:
                        data = self.mem.load(int(addr[l]), nb).view(np.uint32)
                        for i in range(d.n):
                            bank[d.idx + i][l] = data[i]
            self._queue(w, "vm", commit)
            return
        if op == "global_atomic_add":
            addr = self._gaddr(w, s[0], s[2]) + ins.offset
            data = w.v[s[1].idx]
            for l in range(64):
                if w.exec[l]:
                    cur = self.mem.load(int(addr[l]), 4).view(np.uint32)[0]
                    self.mem.store(int(addr[l]), np.array([(int(cur) + int(data[l])) & 0xFFFFFFFF], np.uint32).view(np.uint8))
            self._queue(w, "vm", lambda: None)
            return
        if op.startswith("global_store"):
            nb = s[1].n * 4
            addr = self._gaddr(w, s[0], s[2] if len(s) > 2 else None) + ins.offset
            data = np.ascontiguousarray(self._regs(w, s[1]).T).view(np.uint8).reshape(64, nb)
            for l in range(64):
                if w.exec[l]:
                    self.mem.store(int(addr[l]), data[l])
            self._queue(w, "vm", lambda: None)
            return

        # ---- VALU ---------------------------------------------------------------------------------
        self._valu(w, ins)

    def _regs_s(self, w: Wave, r: isa.Reg):
        return [int(w.s[r.idx + i]) for i in range(r.n)]

    def _gaddr(self, w: Wave, vaddr: isa.Reg, saddr) -> np.ndarray:
        if saddr is None:
            return self._addr64(w, vaddr).astype(np.int64)
        return (w.v[vaddr.idx].astype(np.int64) + self._rds64(w, saddr))

    def _addr64(self, w: Wave, r: isa.Reg) -> np.ndarray:
        bank = w.v
        return bank[r.idx].astype(np.uint64) | (bank[r.idx + 1].astype(np.uint64) << np.uint64(32))

    # ---------------------------------------------------------------------------------------------
    def _salu(self, w: Wave, ins: isa.Instr):
        op, s = ins.op, ins.src
        d = ins.dst[0] if ins.dst else None
        M = 0xFFFFFFFF
        if op.startswith("s_load_dword"):
            n = d.n
            base = self._rds64(w, s[0])
            data = self.mem.load(base + ins.offset, 4 * n).view(np.uint32).copy()
            def commit(d=d, data=data):
                for i in range(d.n):
                    w.s[d.idx + i] = data[i]
            self._queue(w, "lgkm", commit)
            return
        if op == "s_mov_b32":
            self._wrs(w, d, self._rds(w, s[0])); return
        if op == "s_mov_b64":
            self._wrs64(w, d, self._rds64(w, s[0])); return
        a = self._rds(w, s[0]) if s else 0
        b = self._rds(w, s[1]) if len(s) > 1 else 0
        sgn = lambda x: x - (1 << 32) if x & 0x80000000 else x
        if op == "s_add_u32":
            r = a + b; self._wrs(w, d, r); w.scc = int(r > M)
        elif op == "s_addc_u32":
            r = a + b + w.scc; self._wrs(w, d, r); w.scc = int(r > M)
        elif op == "s_sub_u32":
            r = a - b; self._wrs(w, d, r); w.scc = int(b > a)
        elif op == "s_add_i32":
            r = sgn(a) + sgn(b); self._wrs(w, d, r); w.scc = int(not (-(1 << 31) <= r < (1 << 31)))
        elif op == "s_sub_i32":
            r = sgn(a) - sgn(b); self._wrs(w, d, r); w.scc = int(not (-(1 << 31) <= r < (1 << 31)))
        elif op == "s_mul_i32":
            self._wrs(w, d, (sgn(a) * sgn(b)))
        elif op == "s_mul_hi_u32":
            self._wrs(w, d, (a * b) >> 32)
        elif op == "s_lshl_b32":
            r = (a << (b & 31)) & M; self._wrs(w, d, r); w.scc = int(r != 0)
        elif op == "s_lshr_b32":
            r = a >> (b & 31); self._wrs(w, d, r); w.scc = int(r != 0)
        elif op == "s_and_b32":
            r = a & b; self._wrs(w, d, r); w.scc = int(r != 0)
        elif op == "s_or_b32":
            r = a | b; self._wrs(w, d, r); w.scc = int(r != 0)
        elif op == "s_xor_b32":
            r = a ^ b; self._wrs(w, d, r); w.scc = int(r != 0)
        elif op == "s_min_u32":
            r = min(a, b); self._wrs(w, d, r); w.scc = int(a <= b)
        elif op == "s_max_u32":
            r = max(a, b); self._wrs(w, d, r); w.scc = int(a >= b)
        elif op == "s_min_i32":
            r = min(sgn(a), sgn(b)); self._wrs(w, d, r); w.scc = int(sgn(a) <= sgn(b))
        elif op == "s_cselect_b32":
            self._wrs(w, d, a if w.scc else b)
        elif op == "s_bitcmp1_b32":
            w.scc = int((a >> (b & 31)) & 1)
        elif op in ("s_cmp_eq_u64", "s_cmp_lg_u64"):
            x, y = self._rds64(w, s[0]), self._rds64(w, s[1])
            w.scc = int((x == y) == op.startswith("s_cmp_eq"))
        elif op.startswith("s_cmp_"):
            u = op.endswith("u32")
            x, y = (a, b) if u else (sgn(a), sgn(b))
            c = op[len("s_cmp_"):-4]
            w.scc = int({"eq": x == y, "lg": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[c])
        elif op == "s_lshl_b64":
            r = (self._rds64(w, s[0]) << (b & 63)) & 0xFFFFFFFFFFFFFFFF; self._wrs64(w, d, r); w.scc = int(r != 0)
        elif op == "s_and_saveexec_b64":
            old = self._rds64(w, isa.EXEC)
            self._wrs64(w, d, old)
            new = old & self._rds64(w, s[0])
            self._wrs64(w, isa.EXEC, new); w.scc = int(new != 0)
        elif op == "s_or_b64":
            r = self._rds64(w, s[0]) | self._rds64(w, s[1]); self._wrs64(w, d, r); w.scc = int(r != 0)
        else:
            raise NotImplementedError(op)

    # ---------------------------------------------------------------------------------------------
    def _valu(self, w: Wave, ins: isa.Instr):
        op, s = ins.op, ins.src
        d = ins.dst[0] if ins.dst else None
        R = lambda k: self._rd(w, s[k])
        F = lambda k: _f(self._rd(w, s[k]))
        I = lambda k: self._rd(w, s[k]).view(np.int32)
        old = np.seterr(all="ignore")
        try:
            if op == "v_mov_b32":
                self._wrv(w, d, R(0))
            elif op == "v_accvgpr_read_b32" or op == "v_accvgpr_write_b32":
                self._wrv(w, d, R(0))
            elif op == "v_add_f32":
                self._wrv(w, d, _u(F(0) + F(1)))
            elif op == "v_pk_add_f32":                             # two fp32 adds: register pairs
                for h in range(2):
                    self._wrv(w, d.sub(h), _u(_f(self._rd(w, s[0].sub(h))) + _f(self._rd(w, s[1].sub(h)))))
            elif op == "v_sub_f32":
                self._wrv(w, d, _u(F(0) - F(1)))
            elif op == "v_mul_f32":
                self._wrv(w, d, _u(F(0) * F(1)))
            elif op == "v_fma_f32":
                self._wrv(w, d, _u((F(0).astype(np.float64) * F(1).astype(np.float64) + F(2).astype(np.float64)).astype(np.float32)))
            elif op == "v_max_f32":
                self._wrv(w, d, _u(np.fmax(F(0), F(1))))
            elif op == "v_max3_f32":
                self._wrv(w, d, _u(np.fmax(np.fmax(F(0), F(1)), F(2))))
            elif op == "v_exp_f32":
                self._wrv(w, d, _u(np.exp2(F(0).astype(np.float64)).astype(np.float32)))
            elif op == "v_sqrt_f32":
                self._wrv(w, d, _u(np.sqrt(F(0).astype(np.float64)).astype(np.float32)))
            elif op == "v_rcp_f32":
                self._wrv(w, d, _u((1.0 / F(0).astype(np.float64)).astype(np.float32)))
            elif op == "v_cvt_f32_u32":
                self._wrv(w, d, _u(R(0).astype(np.float32)))
            elif op == "v_cvt_u32_f32":
                x = np.nan_to_num(F(0).astype(np.float64), nan=0.0, posinf=4294967295.0, neginf=0.0)
                self._wrv(w, d, np.clip(np.trunc(x), 0, 4294967295).astype(np.uint64).astype(np.uint32))
            elif op == "v_cvt_pk_bf16_f32":
                self._wrv(w, d, bf16_round(F(0)) | (bf16_round(F(1)) << 16))
            elif op == "v_permlane32_swap_b32":
                ra, rb_ = ins.dst[0], ins.dst[1]                     # vdst upper half <-> vsrc lower half
                a, b = w.v[ra.idx].copy(), w.v[rb_.idx].copy()
                na, nb = a.copy(), b.copy()
                na[32:], nb[:32] = b[:32], a[32:]
                w.v[ra.idx], w.v[rb_.idx] = na, nb
            elif op == "v_permlane16_swap_b32":
                ra, rb_ = ins.dst[0], ins.dst[1]                     # vdst odd rows <-> vsrc even rows (rows of 16 lanes)
                a, b = w.v[ra.idx].copy(), w.v[rb_.idx].copy()
                na, nb = a.copy(), b.copy()
                na[16:32], nb[0:16] = b[0:16], a[16:32]
                na[48:64], nb[32:48] = b[32:48], a[48:64]
                w.v[ra.idx], w.v[rb_.idx] = na, nb
            elif op == "v_add_u32":
                self._wrv(w, d, (R(0).astype(np.uint64) + R(1)).astype(np.uint32))
            elif op == "v_subrev_u32":
                self._wrv(w, d, (R(1).astype(np.int64) - R(0)).astype(np.uint32))
            elif op == "v_sub_u32":
                self._wrv(w, d, (R(0).astype(np.int64) - R(1)).astype(np.uint32))
            elif op == "v_mul_lo_u32":
                self._wrv(w, d, (R(0).astype(np.uint64) * R(1).astype(np.uint64)).astype(np.uint32))
            elif op == "v_mul_u32_u24":
                self._wrv(w, d, ((R(0) & 0xFFFFFF).astype(np.uint64) * (R(1) & 0xFFFFFF)).astype(np.uint32))
            elif op == "v_lshlrev_b32":
                self._wrv(w, d, (R(1).astype(np.uint64) << (R(0) & 31).astype(np.uint64)).astype(np.uint32))
            elif op == "v_lshrrev_b32":
                self._wrv(w, d, R(1) >> (R(0) & 31))
            elif op == "v_and_b32":
                self._wrv(w, d, R(0) & R(1))
            elif op == "v_or_b32":
                self._wrv(w, d, R(0) | R(1))
            elif op == "v_or3_b32":
                self._wrv(w, d, R(0) | R(1) | R(2))
            elif op == "v_xor_b32":
                self._wrv(w, d, R(0) ^ R(1))
            elif op == "v_lshl_add_u32":
                self._wrv(w, d, ((R(0).astype(np.uint64) << (R(1) & 31).astype(np.uint64)) + R(2)).astype(np.uint32))
            elif op == "v_lshl_or_b32":
                self._wrv(w, d, ((R(0).astype(np.uint64) << (R(1) & 31).astype(np.uint64)).astype(np.uint32) | R(2)))
            elif op == "v_and_or_b32":
                self._wrv(w, d, (R(0) & R(1)) | R(2))
            elif op == "v_add3_u32":
                self._wrv(w, d, (R(0).astype(np.uint64) + R(1) + R(2)).astype(np.uint32))
            elif op == "v_mad_u32_u24":
                self._wrv(w, d, ((R(0) & 0xFFFFFF).astype(np.uint64) * (R(1) & 0xFFFFFF) + R(2)).astype(np.uint32))
            elif op == "v_min_u32":
                self._wrv(w, d, np.minimum(R(0), R(1)))
            elif op == "v_min_i32":
                self._wrv(w, d, np.minimum(I(0), I(1)).view(np.uint32))
            elif op == "v_mbcnt_lo_u32_b32" or op == "v_mbcnt_hi_u32_b32":
                # used only as the lane-id idiom: mbcnt_lo(-1, 0) then mbcnt_hi(-1, prev)
                lane = np.arange(64, dtype=np.uint32)
                self._wrv(w, d, np.minimum(lane, 32) if op.endswith("lo_u32_b32") else lane)
            elif op == "v_add_co_u32":            # dst, vcc = a + b
                r = R(0).astype(np.uint64) + R(1)
                self._wrv(w, d, r.astype(np.uint32)); w.vcc = (r >> 32) != 0
            elif op == "v_addc_co_u32":           # dst, vcc = a + b + vcc_in   (src: a, b, vcc)
                r = R(0).astype(np.uint64) + R(1) + w.vcc.astype(np.uint64)
                self._wrv(w, d, r.astype(np.uint32)); w.vcc = (r >> 32) != 0
            elif op.startswith("v_cmp_"):
                body = op[len("v_cmp_"):]
                cmp, ty = body.rsplit("_", 1)
                x, y = (F(0), F(1)) if ty == "f32" else ((I(0), I(1)) if ty == "i32" else (R(0), R(1)))
                r = {"gt": x > y, "lt": x < y, "ge": x >= y, "le": x <= y, "eq": x == y, "ne": x != y, "lg": x != y,
                     "ngt": ~(x > y), "nlt": ~(x < y), "nge": ~(x >= y), "nle": ~(x <= y)}[cmp]        # n*: true for NaN operands
                r = np.asarray(r) & w.exec
                if d.kind == "vcc":
                    w.vcc = r
                else:
                    self._wrs64(w, d, int(sum(1 << i for i in range(64) if r[i])))
            elif op == "v_cndmask_b32":
                m = w.vcc if s[2].kind == "vcc" else np.array([(self._rds64(w, s[2]) >> i) & 1 for i in range(64)], dtype=bool)
                self._wrv(w, d, np.where(m, R(1), R(0)))
            elif op == "v_readfirstlane_b32":
                first = int(np.argmax(w.exec)) if w.exec.any() else 0
                self._wrs(w, d, int(self._rd(w, s[0])[first]))
            else:
                raise NotImplementedError(op)
        finally:
            np.seterr(**old)
