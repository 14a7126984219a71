"""Sequence parallelism over the token axis for the SCAIL DiT.

Reference behaviour being replaced: DeepSpeed-Ulysses (sat/mpu/ulysses_attn_layer.py:41-110,
all_to_all.py:15-108) over an H- or W-split of the latent (diffusion_video.py:495-552), with the
RoPE window shifted by the rank (dit_video_crossattn_sc_xc.py:1578-1585) and a final gather to
SP rank 0 (diffusion_video.py:571-585).

MI355X design: the same latent split and rank-shifted RoPE (so a rank's tokens are an aligned slab
of ref, noise and pose tokens), but instead of 4 all-to-alls per attention (12 per layer, also for
the two cross-attentions whose K/V are replicated anyway) each layer does ONE exchange: an
all-gather of the post-norm, post-RoPE K rows and of the V rows over xGMI (RCCL; direct peer links, no
ring dependence), after which every rank runs full attention of its local queries against the gathered
keys -- one contiguous (L, D) matrix in rank-major token order, so the launch is the ordinary
single-segment one.  Softmax is permutation invariant over keys, so rank-major key order is harmless.  Everything else in the
block is per-token and needs no communication; weights are replicated (32 GB << 288 GB HBM).
The K/V projection is issued first so the all-gather overlaps the Q projection + Q norm/RoPE; the two CFG batch
elements are independent sequences, so the exchange of one runs under the attention of the other (both modes).
"""
from __future__ import annotations

import contextlib
import os
import threading
from typing import List, Optional

import torch

from . import lib as L
from . import ops


class _Handle:
    def __init__(self, work=None):
        self.work = work

    def wait(self):
        if self.work is not None:
            self.work.wait()


class TorchDistBackend:
    """torch.distributed process group (backend 'nccl' == RCCL on ROCm; 'gloo' in CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self._src = dist.get_global_rank(group, 0) if group is not None else 0
        # gloo moves host memory only: device tensors are staged through the host.  That is a TEST vehicle (several
        # ranks sharing one GPU, where RCCL refuses duplicate devices), never the production path (nccl == RCCL).
        self.host_staged = dist.get_backend(group) == "gloo"

    def _staged(self, t):
        return self.host_staged and t.is_cuda

    def broadcast(self, t):
        if self._staged(t):
            h = t.cpu()
            self.dist.broadcast(h, src=self._src, group=self.group)
            t.copy_(h)
            return
        self.dist.broadcast(t, src=self._src, group=self.group)

    def all_gather_into(self, out, inp, async_op=True):
        """out: (size, *inp.shape) contiguous."""
        if self._staged(inp):
            hi = inp.reshape(-1).cpu()
            ho = [torch.empty_like(hi) for _ in range(self.size)]     # gloo has no all_gather_into_tensor
            self.dist.all_gather(ho, hi, group=self.group)
            out.view(self.size, -1).copy_(torch.stack(ho))
            return _Handle(None)
        w = self.dist.all_gather_into_tensor(out.view(-1), inp.reshape(-1), group=self.group, async_op=async_op)
        return _Handle(w if async_op else None)

    def all_to_all(self, out, inp, async_op=False):
        """out[s] <- rank s's inp[my rank]; out, inp: (size, ...) contiguous.  async_op: returns a handle whose wait()
        makes the current stream wait (RCCL runs the exchange on its own stream meanwhile)."""
        if self.host_staged:                                          # gloo has no all_to_all_single: N gathers
            hi = inp.reshape(self.size, -1).cpu() if inp.is_cuda else inp.reshape(self.size, -1)
            rows = []
            for dst in range(self.size):
                lst = [torch.empty_like(hi[dst]) for _ in range(self.size)] if self.rank == dst else None
                self.dist.gather(hi[dst].contiguous(), gather_list=lst,
                                 dst=self.dist.get_global_rank(self.group, dst) if self.group is not None else dst,
                                 group=self.group)
                if self.rank == dst:
                    rows = lst
            out.view(self.size, -1).copy_(torch.stack(rows))
            return _Handle(None)
        w = self.dist.all_to_all_single(out.view(-1), inp.reshape(-1), group=self.group, async_op=async_op)
        return _Handle(w if async_op else None)

    def gather_cat(self, t, dim):
        """Concatenate along ``dim`` on group rank 0 (reference: dist.gather + concat, :578-585);
        other ranks get their own shard back."""
        if self._staged(t):
            h = t.cpu()
            lst = [torch.zeros_like(h) for _ in range(self.size)] if self.rank == 0 else None
            self.dist.gather(h, gather_list=lst, dst=self._src, group=self.group)
            return torch.cat(lst, dim=dim).to(t.device) if self.rank == 0 else t
        if self.rank == 0:
            lst = [torch.zeros_like(t) for _ in range(self.size)]
        else:
            lst = None
        self.dist.gather(t, gather_list=lst, dst=self._src, group=self.group)
        return torch.cat(lst, dim=dim) if self.rank == 0 else t


class ThreadBackend:
    """N virtual ranks as threads of ONE process sharing one GPU (or the CPU): lets the multi-rank
    data path -- including the multi-segment attention kernel -- be exercised on a 1-GPU box."""

    class Shared:
        def __init__(self, size):
            self.size = size
            self.barrier = threading.Barrier(size)
            self.slots: List[Optional[torch.Tensor]] = [None] * size

    def __init__(self, shared: "ThreadBackend.Shared", rank: int):
        self.shared, self.rank, self.size = shared, rank, shared.size

    def _sync(self):
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        self.shared.barrier.wait()

    def broadcast(self, t):
        if self.rank == 0:
            self.shared.slots[0] = t
        self._sync()
        if self.rank != 0:
            t.copy_(self.shared.slots[0])
        self._sync()

    def all_gather_into(self, out, inp, async_op=True):
        self.shared.slots[self.rank] = inp
        self._sync()
        for r in range(self.size):
            out[r].copy_(self.shared.slots[r].reshape(out[r].shape))
        self._sync()
        return _Handle(None)

    def all_to_all(self, out, inp, async_op=False):
        self.shared.slots[self.rank] = inp
        self._sync()
        for r in range(self.size):
            out[r].copy_(self.shared.slots[r][self.rank])
        self._sync()
        return _Handle(None)

    def gather_cat(self, t, dim):
        self.shared.slots[self.rank] = t
        self._sync()
        res = torch.cat([s for s in self.shared.slots], dim=dim) if self.rank == 0 else t
        self._sync()
        return res


class LocalCopyBackend:
    """MEASUREMENT backend: rank 0 of an N-rank group on ONE GPU, every peer's contribution replaced by a local copy of this rank's own
    data.  The launch sequence and every kernel shape are those of a real rank (results are meaningless), so a step through it is the
    compute side of an N-rank step: (T_1 / N) / T_N is the compute-only strong-scaling efficiency; the exchange time comes on top
    (bench.py ``config.sp_compute_side``, tools/sp_rank_compute.py)."""

    def __init__(self, size):
        self.rank, self.size = 0, size

    def broadcast(self, t):
        pass

    def all_gather_into(self, out, inp, async_op=True):
        for r in range(self.size):
            out[r].copy_(inp.reshape(out[r].shape))
        return _Handle(None)

    def all_to_all(self, out, inp, async_op=False):
        out.copy_(inp)
        return _Handle(None)

    def gather_cat(self, t, dim):
        return t


class ConcurrentCopyBackend(LocalCopyBackend):
    """MEASUREMENT backend: LocalCopyBackend whose "collectives" behave like RCCL's on the chip -- they run on their OWN stream as a kernel of
    ``workgroups`` resident workgroups (one CU each, like RCCL's channels) that moves the real message and stays resident for the time the
    xGMI links would need for the bytes that leave the GPU (``link_gbps`` per direction and peer link, N - 1 links in parallel), while the
    compute stream carries on until the executor's WAIT callback.  What it shows on one GPU: how much of a rank's step the CU occupancy of
    its collectives costs (the attention launches one workgroup per CU; a CU held by a channel is missing from its round), for a channel
    count K -- the measurement behind the decision NOT to cap RCCL's channels (bench.py header; tools/sp_rank_compute.py --comm-wgs K,
    profiles/r06_sp_comm_standin.log)."""

    def __init__(self, size, workgroups=16, link_gbps=48.0, device="cuda"):
        super().__init__(size)
        self.workgroups, self.link_gbps = int(workgroups), float(link_gbps)
        self.stream = torch.cuda.Stream(device=device)

    def _launch(self, out, inp, link_bytes):
        """link_bytes: what the busiest peer link carries in one direction; the kernel stays resident for link_bytes / link_gbps"""
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)                     # the message is complete on the compute stream (stream order)
        min_ns = int(link_bytes / self.link_gbps)        # bytes / (GB/s) = ns
        nb = inp.numel() * inp.element_size()
        L.call("scail_comm_standin", inp.data_ptr(), out.data_ptr(), nb, self.workgroups, min_ns, self.stream.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(self.stream)

        class _H:
            def wait(self_inner):
                torch.cuda.current_stream().wait_event(ev)
        return _H()

    def all_to_all(self, out, inp, async_op=False):
        nb = inp.numel() * inp.element_size()
        h = self._launch(out, inp, nb / self.size)       # one of the N equal chunks to every peer, each over its own link
        if not async_op:
            h.wait()
        return h

    def all_gather_into(self, out, inp, async_op=True):
        nb = inp.numel() * inp.element_size()
        hs = [self._launch(out[r], inp, nb if r == self.size - 1 else 0) for r in range(self.size)]     # the whole message to (from) every peer
        return hs[-1]


# ulysses exchange: one side stream per CFG element up to this many ranks (CExchange).  0 since round 5: with the planned attention launch
# shape (whole 256-row rounds + 192-row tiles, csrc/attn.hip attn4_plan) one stream is faster at 4 ranks -- compute-side efficiency 95.3 %
# against 94.2 % with the two side streams, same box, same process order (profiles/r05_sp4_side_streams_ab.log); rounds 3-4, before the
# plan: 93.9 % with, 91.4 % without.  SCAIL_SP_SIDE_STREAMS=1 brings them back.
SIDE_STREAM_MAX_RANKS = 0


class CExchange:
    """Host side of the C executor's exchange callback (include/scail_dit.h "sequence-parallel execution"): owns the send / recv /
    ofull / back buffers of one (B, Ltok) shape and starts / awaits the collectives through the group's backend when the executor
    asks for them.  The executor enqueues every kernel of a block itself; per layer this object only sees 4 (all-gather) or 8
    (ulysses) callbacks, which start 2 / 4 collectives (one all-gather, or one all-to-all each way, per CFG element)."""

    def __init__(self, sp: "SequenceParallel", heads: int, D: int, B: int, Ltok: int, device):
        from . import cstep
        self._cs = cstep
        self.backend, self.size = sp.backend, sp.size
        self.mode = sp.resolve_mode(heads)
        self.mode_code = cstep.SP_ULYSSES if self.mode == "ulysses" else cstep.SP_ALLGATHER
        N = self.size
        e = lambda *sh: torch.empty(*sh, device=device, dtype=torch.bfloat16)
        if self.mode == "ulysses":
            Dn = D // N
            self.send, self.recv = e(B, N, Ltok, 3 * Dn), e(B, N, Ltok, 3 * Dn)      # one message per peer: q | k | v side by side
            self.ofull, self.back = e(B, N, Ltok, Dn), e(B, N, Ltok, Dn)
        else:
            self.send, self.recv = e(B, Ltok, 2 * D), e(B, N, Ltok, 2 * D)             # k | v side by side
            self.ofull = self.back = None
        # optional: two side streams for the two CFG elements (see SIDE_STREAM_MAX_RANKS)
        # (SCAIL_SP_SIDE_STREAMS = 0 / 1 overrides the rule: same-process A/B against the planned single-stream launch shape)
        want = os.environ.get("SCAIL_SP_SIDE_STREAMS")
        use_side = (N <= SIDE_STREAM_MAX_RANKS) if want is None else want == "1"
        self.side = [torch.cuda.Stream(device=device) for _ in range(2)] if (self.mode == "ulysses" and use_side and torch.device(device).type == "cuda") else None
        self.device = torch.device(device)
        self.handles = {}
        self.error = None
        self._ext = {}
        self._cb = cstep.EXCHANGE_FN(self._callback)            # must outlive every call that uses it

    def descriptor(self):
        cs = self._cs
        p = lambda t: t.data_ptr() if t is not None else None
        side = (cs._p * 2)(*( [s.cuda_stream for s in self.side] if self.side else [None, None]))
        return cs.DitSp(self.size, self.mode_code, p(self.send), p(self.recv), p(self.ofull), p(self.back), self._cb, None, side)

    def _on(self, stream):
        """context that makes ``stream`` (a hipStream_t value) torch's current stream, so that the collective orders itself
        against the kernels the executor enqueued there"""
        ptr = stream or 0
        if ptr == torch.cuda.current_stream(self.device).cuda_stream:
            return contextlib.nullcontext()
        if ptr not in self._ext:
            self._ext[ptr] = torch.cuda.ExternalStream(ptr, device=self.device)
        return torch.cuda.stream(self._ext[ptr])

    def _callback(self, user, op, layer, b, stream):
        cs = self._cs
        try:
            with self._on(stream):
                if op == cs.SP_FWD_START:
                    if self.mode == "ulysses":
                        self.handles["f", b] = [self.backend.all_to_all(self.recv[b], self.send[b], async_op=True)]
                    else:
                        self.handles["f", b] = [self.backend.all_gather_into(self.recv[b], self.send[b])]
                elif op == cs.SP_FWD_WAIT:
                    for h in self.handles.pop(("f", b)):
                        h.wait()
                elif op == cs.SP_BACK_START:
                    self.handles["b", b] = [self.backend.all_to_all(self.back[b], self.ofull[b], async_op=True)]
                elif op == cs.SP_BACK_WAIT:
                    for h in self.handles.pop(("b", b)):
                        h.wait()
                else:
                    raise L.ScailHipError(f"unknown exchange op {op}")
            return 0
        except BaseException as e:              # never let an exception cross the C frame: hand it to CStep._sp_call
            self.error = e
            return 1


class SequenceParallel:
    """What the engine and the DiT need from a sequence-parallel group."""

    def __init__(self, backend, mode: str = "auto"):
        """mode: 'allgather' (K / V^T all-gather, any head count), 'ulysses' (head <-> sequence all-to-all of
        q, k, v and o like the reference, 4x less traffic at 8 ranks, needs heads % size == 0) or 'auto'
        (ulysses when it divides and size >= 4: received bytes per rank and layer at config 2 are
        2 GB (N-1)/N for the all-gather vs 4 GB (N-1)/N^2 for the all-to-alls)."""
        self.backend = backend
        self.rank, self.size = backend.rank, backend.size
        self.mode = mode
        self._buf = {}
        self._streams = None
        self._xch = {}

    def c_exchange(self, heads: int, D: int, B: int, Ltok: int, device) -> CExchange:
        """The exchange object the C executor calls back into (one per (B, Ltok) shape; one shape is kept)."""
        key = (heads, D, B, Ltok, str(device))
        if key not in self._xch:
            self._xch = {key: CExchange(self, heads, D, B, Ltok, device)}
        return self._xch[key]

    def resolve_mode(self, heads: int) -> str:
        if self.mode != "auto":
            if self.mode == "ulysses" and heads % self.size:
                raise ValueError(f"ulysses needs heads ({heads}) divisible by the group size ({self.size})")
            return self.mode
        return "ulysses" if (heads % self.size == 0 and self.size >= 4) else "allgather"

    # ---- host-side sharding (diffusion_video.py:495-552) ----
    @staticmethod
    def chunk_dim_for(shape_hw) -> int:
        h, w = shape_hw
        return 3 if h < w else 4

    def chunk(self, t: torch.Tensor, dim: int) -> torch.Tensor:
        if t.shape[dim] % self.size:
            raise ValueError(f"dim {dim} of size {t.shape[dim]} does not split over {self.size} ranks")
        return torch.chunk(t, self.size, dim=dim)[self.rank].contiguous()

    def check_latent(self, H: int, W: int, chunk_dim: int):
        """Each rank needs an even number of patch rows/cols of the half-size pose latent:
        (H / size) % 4 == 0 along the split axis (reference needs the same for avg_pool2d, :630)."""
        ext = H if chunk_dim == 3 else W
        if ext % (4 * self.size):
            raise ValueError(f"latent extent {ext} along the split axis must be a multiple of 4*sp_size={4 * self.size}")

    def broadcast(self, t):
        self.backend.broadcast(t)

    def gather_to_rank0(self, t, dim):
        return self.backend.gather_cat(t.contiguous(), dim)

    def self_check(self, device) -> dict:
        """First contact with the fabric: run each collective the layer exchange uses on rank-stamped data and verify the
        result on every rank BEFORE any timed or production work (a wrong-but-silent exchange would only show up as a bad
        video).  Returns what a log line needs to validate a scaling run: group size, backend, RCCL version, mode.
        Raises RuntimeError naming the collective that failed."""
        N, r = self.size, self.rank
        bf = torch.bfloat16
        # stamps are 16-bit INTEGER patterns carried in the bf16 payload (viewed, not converted): every (source, destination,
        # element) triple is a distinct value, so a chunk routed to the wrong rank or rows permuted inside a chunk cannot compare
        # equal (bf16-rounded float stamps above 256 collapse neighbouring ranks / elements onto the same value)
        assert N * N * 8 < 32768
        stamp = lambda src, dst: ((src * N + dst) * 8 + torch.arange(8, device=device, dtype=torch.int32) + 1).to(torch.int16)
        inp = torch.stack([stamp(r, d) for d in range(N)]).view(bf)
        got = torch.empty_like(inp)
        self.backend.all_to_all(got, inp, async_op=True).wait()
        want = torch.stack([stamp(s, r) for s in range(N)])
        if not torch.equal(got.view(torch.int16), want):
            raise RuntimeError(f"sequence-parallel self-check: all_to_all returned wrong data on rank {r} of {N}")
        loc = (torch.arange(32, device=device, dtype=torch.int32).view(4, 8) + 64 * r + 1).to(torch.int16).view(bf)
        allg = torch.empty(N, 4, 8, device=device, dtype=bf)
        self.backend.all_gather_into(allg, loc).wait()
        want = (torch.arange(32, device=device, dtype=torch.int32).view(1, 4, 8)
                + 64 * torch.arange(N, device=device, dtype=torch.int32).view(N, 1, 1) + 1).to(torch.int16)
        if not torch.equal(allg.view(torch.int16), want):
            raise RuntimeError(f"sequence-parallel self-check: all_gather returned wrong data on rank {r} of {N}")
        t = torch.full((4,), float(r == 0) * 7.0, device=device, dtype=torch.float32)
        self.backend.broadcast(t)
        if not bool((t == 7.0).all()):
            raise RuntimeError(f"sequence-parallel self-check: broadcast returned wrong data on rank {r} of {N}")
        info = {"ranks": N, "collectives_verified": ["all_to_all", "all_gather", "broadcast"]}
        d = getattr(self.backend, "dist", None)
        if d is not None:
            info["backend"] = d.get_backend(self.backend.group)
            if info["backend"] == "nccl":
                info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        return info

    # ---- the one per-layer exchange ----
    def self_attention(self, net, lw, xn, qkv, vt_loc, cos, sin, att, Ltok, eps):
        if self.resolve_mode(net.num_attention_heads) == "ulysses":
            return self.self_attention_ulysses(net, lw, xn, qkv, cos, sin, att, Ltok, eps)
        return self.self_attention_allgather(net, lw, xn, qkv, vt_loc, cos, sin, att, Ltok, eps)

    def self_attention_ulysses(self, net, lw, xn, qkv, cos, sin, att, Ltok, eps):
        """The reference's exchange (sat/mpu/ulysses_attn_layer.py:65-107): scatter heads / gather sequence for q, k, v,
        attention over the full sequence for heads/size heads, all-to-all back.  Norm + RoPE are applied before the
        exchange (the q/k RMSNorm spans all heads of a token).
        Layout: q, k and v travel in ONE all-to-all of (dst rank, Ltok, 3 Dn) messages (q | k | v of the destination's heads side
        by side in a row; the reference issues three collectives), so what a rank receives is (src rank, Ltok, 3 Dn) = one
        (size * Ltok, 3 Dn) matrix in rank-major token order whose column thirds are q, k, v.  The attention is then the
        ordinary single-segment full-length launch on heads/size heads -- size * Ltok is a multiple of 64 whenever the
        unsharded length is, so the 4-wave kernel (scail_flash_attn_kernel_for == 4) serves it although the per-rank slabs
        are ragged (6 104 tokens at 8 ranks).  Key order is rank-major, not raster order: softmax is permutation invariant
        over keys, and q / o use the same order, undone by the way back.
        The CFG batch elements are independent sequences, so they are pipelined: the exchange of element b+1 runs (on
        RCCL's stream) under the attention of element b, and the way back of b under the attention of b+1."""
        D, nh, N = net.hidden_size, net.num_attention_heads, self.size
        B = xn.shape[0]
        Hn = nh // N
        Dn = Hn * 128
        Lf = N * Ltok
        key = ("u", B, Ltok)
        if key not in self._buf:
            dev = xn.device
            e = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.bfloat16)
            self._buf = {key: dict(send=e(B, N, Ltok, 3 * Dn), recv=e(B, N, Ltok, 3 * Dn), vt=e(B, 1, Hn, 128, (Lf + 63) // 64 * 64),
                                   ofull=e(B, N, Ltok, Dn), back=e(B, N, Ltok, Dn))}
        bf = self._buf[key]
        send, recv, vt, ofull, back = bf["send"], bf["recv"], bf["vt"], bf["ofull"], bf["back"]
        # One side stream per CFG element: the launches of an element leave a partial last round of the chip (10 heads x 191 query
        # blocks = 7.5 rounds at 4 ranks); on separate streams the other element's kernels fill those CUs.  Measured with
        # tools/sp_rank_compute.py: 4 ranks 91.4 -> 93.9 % compute-side efficiency, 8 ranks 90.7 -> 89.4 % (two resident kernels
        # halve each other's L2 share; with a high-priority first stream 87 %), so: two streams up to 4 ranks, one beyond.
        # The collectives are still ENQUEUED in the order fwd(0), fwd(1), back(0), back(1) on every rank.
        main = torch.cuda.current_stream() if (xn.is_cuda and N <= 4) else None
        if main is not None and self._streams is None:
            self._streams = [torch.cuda.Stream(device=xn.device) for _ in range(2)]
        ctx = (lambda b: torch.cuda.stream(self._streams[b % 2])) if main is not None else (lambda b: contextlib.nullcontext())
        if main is not None:
            for st in self._streams[:min(B, 2)]:
                st.wait_stream(main)
        fwd = []
        for b in range(B):
            with ctx(b):
                # projection + norm + RoPE per element: the exchange of element b runs under the projection of element b+1
                ops.gemm(xn[b], lw["qkv_w"], lw["qkv_b"], out=qkv[b])
                # the norm + RoPE kernels write the SEND layout (dst rank, Ltok, Dn) themselves -- one column slab per destination
                # rank -- instead of normalising in place and packing afterwards (the reference's `.permute().contiguous()` before
                # all_to_all_single, sat/mpu/ulysses_attn_layer.py:65-80); v is a plain copy into the same layout
                self.pack_rows(qkv[b, :, D:2 * D], send[b, :, :, Dn:2 * Dn], lw["kn"], cos, sin, Ltok, eps, 1.0)
                self.pack_rows(qkv[b, :, :D], send[b, :, :, :Dn], lw["qn"], cos, sin, Ltok, eps, ops.ATTN_LOG2_SCALE)     # q in log2 units
                self.pack_rows(qkv[b, :, 2 * D:], send[b, :, :, 2 * Dn:], None, None, None, Ltok, eps, 1.0)
                fwd.append([self.backend.all_to_all(recv[b], send[b], async_op=True)])
        bwd = []
        for b in range(B):
            with ctx(b):
                for h in fwd[b]:
                    h.wait()
                qf, kf, vf = recv[b].view(1, Lf, 3 * Dn).split(Dn, dim=2)                # all ranks' tokens, my heads
                ops.transpose_v(vf, Hn, out=vt[b])
                net._timed("self_attn", ops.flash_attn, qf, kf, vt[b], out=ofull[b].view(1, Lf, Dn), q_prescaled=True)
                bwd.append(self.backend.all_to_all(back[b], ofull[b], async_op=True))    # back[b][g] = my tokens, head group g
        for b in range(B):
            with ctx(b):
                bwd[b].wait()
                att[b].view(Ltok, N, Dn).copy_(back[b].permute(1, 0, 2))
        if main is not None:
            for st in self._streams[:min(B, 2)]:
                main.wait_stream(st)
        return att

    @staticmethod
    def pack_rows(x, out, w, cos, sin, Ltok, eps, out_scale):
        """x (Ltok, D) view -> out (N, Ltok, D / N) (a column third of the (N, Ltok, 3 D / N) message): RMSNorm + RoPE (w given) or plain copy (w None) straight into the send layout
        of the head <-> sequence all-to-all: one kernel (scail_rmsnorm_rope_slabs).  GPU tensors only -- there is no CPU path; the layout
        is pinned by tests/test_kernels_gpu.py::test_rmsnorm_rope_slabs: out == rows.view(Ltok, N, D / N).permute(1, 0, 2) of the
        row-major kernel's result, bit for bit."""
        if x.is_cuda:
            return ops.rmsnorm_rope_slabs(x, w, out, cos, sin, rows_per_batch=Ltok, eps=eps, out_scale=out_scale)
        raise L.ScailHipError("sequence-parallel self-attention needs GPU tensors (scail_amd has no CPU path)")

    def self_attention_allgather(self, net, lw, xn, qkv, vt_loc, cos, sin, att, Ltok, eps):
        """xn (B, Lloc, D) -> att (B, Lloc, D): per CFG batch element K / V projection, K norm + RoPE and the all-gather of
        both (so the gather of element b runs under the projection of element b+1), Q projection + norm + RoPE under the
        last gather, then attention of the local queries over all ranks' keys element by element.
        Layout: K and V ROWS are gathered (not per-rank V^T images) side by side in ONE collective, so they arrive as one
        (size * Ltok, 2 D) matrix in rank-major token order (column halves k, v) and the attention is the ordinary single-segment launch (4-wave kernel whenever
        the unsharded length is a multiple of 64, however ragged the per-rank slabs are); V^T is staged from the gathered
        rows (one pass over 2 L D bytes per element: 0.1 % of the attention it feeds)."""
        D, nh, N = net.hidden_size, net.num_attention_heads, self.size
        B = xn.shape[0]
        Lf = N * Ltok
        q = qkv[..., :D]
        key = (B, Ltok)
        if key not in self._buf:
            dev = xn.device
            e = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.bfloat16)
            self._buf = {key: dict(kvloc=e(B, Ltok, 2 * D), kvg=e(B, N, Ltok, 2 * D), vt=e(B, 1, nh, 128, (Lf + 63) // 64 * 64))}
        bufs = self._buf[key]
        kvloc, kvg, vt = bufs["kvloc"], bufs["kvg"], bufs["vt"]          # k | v side by side in a row: ONE all-gather per element
        hs = []
        for b in range(B):      # K / V projection per element: the gather of element b runs under the projection of element b+1
            ops.gemm(xn[b], lw["qkv_w"][D:2 * D], lw["qkv_b"][D:2 * D], out=qkv[b, :, D:2 * D])
            ops.gemm(xn[b], lw["qkv_w"][2 * D:], lw["qkv_b"][2 * D:], out=kvloc[b, :, D:])
            ops.rmsnorm_rope(qkv[b:b + 1, :, D:2 * D], lw["kn"], cos, sin, out=kvloc[b:b + 1, :, :D], rows_per_batch=Ltok, eps=eps)
            hs.append(self.backend.all_gather_into(kvg[b], kvloc[b]))
        ops.gemm(xn, lw["qkv_w"][:D], lw["qkv_b"][:D], out=q)                       # overlaps the exchange
        ops.rmsnorm_rope(q, lw["qn"], cos, sin, rows_per_batch=Ltok, eps=eps, out_scale=ops.ATTN_LOG2_SCALE)        # q in log2 units
        for b in range(B):
            hs[b].wait()
            kg, vg = kvg[b].view(1, Lf, 2 * D).split(D, dim=2)
            ops.transpose_v(vg, nh, out=vt[b])
            net._timed("self_attn", ops.flash_attn, q[b:b + 1], kg, vt[b], out=att[b:b + 1], q_prescaled=True)
        return att


def init_from_env(backend: str = "nccl") -> Optional[SequenceParallel]:
    """One process per GPU, launched by torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* from the environment.  The whole world is one sequence-parallel group (the reference's
    CLI asserts dp == 1, sample_video.py:229).  Returns None for a single process."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return None
    if not dist.is_initialized():
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return SequenceParallel(TorchDistBackend(None), mode=os.environ.get("SCAIL_SP_MODE", "auto"))
