"""Sequence parallelism over the token axis for the SCAIL DiT.

Reference behaviour being replaced: DeepSpeed-Ulysses (sat/mpu/ulysses_attn_layer.py:41-110,
all_to_all.py:15-108) over an H- or W-split of the latent (diffusion_video.py:495-552), with the
RoPE window shifted by the rank (dit_video_crossattn_sc_xc.py:1578-1585) and a final gather to
SP rank 0 (diffusion_video.py:571-585).

MI355X design: the same latent split and rank-shifted RoPE (so a rank's tokens are an aligned slab
of ref, noise and pose tokens), but instead of 4 all-to-alls per attention (12 per layer, also for
the two cross-attentions whose K/V are replicated anyway) each layer does ONE exchange: an
all-gather of the post-norm, post-RoPE K and of the V^T staging buffer over xGMI (RCCL; direct
peer links, no ring dependence), after which every rank runs full attention of its local queries
against the n_seg gathered key segments (``scail_flash_attn_bf16`` n_seg > 1).  Softmax is
permutation invariant over keys, so rank-major key order is harmless.  Everything else in the
block is per-token and needs no communication; weights are replicated (32 GB << 288 GB HBM).
The K/V projection is issued first so the all-gather overlaps the Q projection + Q norm/RoPE; the two CFG batch
elements are independent sequences, so the exchange of one runs under the attention of the other (both modes).
"""
from __future__ import annotations

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

    # ---- the one per-layer exchange ----
    def self_attention(self, net, lw, xn, qkv, vt_loc, cos, sin, att, Ltok, eps):
        if self.resolve_mode(net.num_attention_heads) == "ulysses":
            return self.self_attention_ulysses(net, lw, xn, qkv, cos, sin, att, Ltok, eps)
        return self.self_attention_allgather(net, lw, xn, qkv, vt_loc, cos, sin, att, Ltok, eps)

    def self_attention_ulysses(self, net, lw, xn, qkv, cos, sin, att, Ltok, eps):
        """The reference's exchange (sat/mpu/ulysses_attn_layer.py:65-107): scatter heads / gather sequence for
        q, k, v in ONE all-to-all, attention over the full sequence for heads/size heads, all-to-all back.
        Norm + RoPE are applied before the exchange (the q/k RMSNorm spans all heads of a token).
        The CFG batch elements are independent sequences, so they are pipelined: the exchange of element b+1
        runs (on RCCL's stream) under the attention of element b, and the way back of b under the attention of b+1."""
        D, nh, N = net.hidden_size, net.num_attention_heads, self.size
        B = xn.shape[0]
        Hn = nh // N
        Dn = Hn * 128
        key = ("u", B, Ltok)
        if key not in self._buf:
            dev = xn.device
            Lp = (Ltok + 63) // 64 * 64
            e = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.bfloat16)
            self._buf = {key: dict(send=e(B, N, 3, Ltok, Dn), recv=e(B, N, 3, Ltok, Dn), vtg=e(B, N, 1, Hn, 128, Lp),
                                   oseg=e(B, N, Ltok, Dn), back=e(B, N, Ltok, Dn))}
        bf = self._buf[key]
        send, recv, vtg, oseg, back = bf["send"], bf["recv"], bf["vtg"], bf["oseg"], bf["back"]
        fwd = []
        for b in range(B):                                                  # (dst rank, q|k|v, L, Dn) per batch element
            # projection + norm + RoPE per element: the exchange of element b runs under the projection of element b+1
            ops.gemm(xn[b], lw["qkv_w"], lw["qkv_b"], out=qkv[b])
            ops.rmsnorm_rope(qkv[b:b + 1, :, D:2 * D], lw["kn"], cos, sin, rows_per_batch=Ltok, eps=eps)
            ops.rmsnorm_rope(qkv[b:b + 1, :, :D], lw["qn"], cos, sin, rows_per_batch=Ltok, eps=eps)
            send[b].copy_(qkv[b].view(Ltok, 3, N, Dn).permute(2, 1, 0, 3))
            fwd.append(self.backend.all_to_all(recv[b], send[b], async_op=True))   # recv[b][src] = its tokens, my heads
        bwd = []
        for b in range(B):
            fwd[b].wait()
            ops.transpose_v(recv[b, :, 2], Hn, out=vtg[b].view(N, Hn, 128, -1))   # all source ranks' V in one launch
            # ONE launch: the source rank is the kernel's batch index for q / o (N x Ltok query rows), K / V^T are the N
            # gathered segments broadcast over it -- N * Ltok / 256 x heads/N workgroups instead of N launches that
            # each fill less than the chip
            net._timed("self_attn", ops.flash_attn, recv[b, :, 0], recv[b, 0:1, 1], vtg[b, 0], out=oseg[b],
                       n_seg=N, k_seg_stride=recv.stride(1), vt_seg_stride=vtg.stride(1))
            bwd.append(self.backend.all_to_all(back[b], oseg[b], async_op=True))    # back[b][g] = my tokens, head group g
        for b in range(B):
            bwd[b].wait()
            att[b].view(Ltok, N, Dn).copy_(back[b].permute(1, 0, 2))
        return att

    def self_attention_allgather(self, net, lw, xn, qkv, vt_loc, cos, sin, att, Ltok, eps):
        """xn (B, Lloc, D) -> att (B, Lloc, D): per CFG batch element K/V projection, K norm+RoPE, V^T staging and the
        all-gather of both (so the gather of element b runs under the projection of element b+1), Q projection + norm +
        RoPE under the last gather, then attention over all ranks' keys element by element."""
        D, nh = net.hidden_size, net.num_attention_heads
        B = xn.shape[0]
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        key = (B, Ltok)
        if key not in self._buf:
            Lp = vt_loc.shape[-1]
            dev = xn.device
            self._buf = {key: dict(
                kloc=torch.empty(B, Ltok, D, device=dev, dtype=torch.bfloat16),
                kg=torch.empty(B, self.size, 1, Ltok, D, device=dev, dtype=torch.bfloat16),
                vtg=torch.empty(B, self.size, 1, nh, 128, Lp, device=dev, dtype=torch.bfloat16))}
        bufs = self._buf[key]
        kloc, kg, vtg = bufs["kloc"], bufs["kg"], bufs["vtg"]
        hs = []
        for b in range(B):      # K / V projection per element: the gather of element b runs under the projection of element b+1
            ops.gemm(xn[b], lw["qkv_w"][D:], lw["qkv_b"][D:], out=qkv[b, :, D:])    # K and V columns
            ops.rmsnorm_rope(k[b:b + 1], lw["kn"], cos, sin, out=kloc[b:b + 1], rows_per_batch=Ltok, eps=eps)
            ops.transpose_v(v[b:b + 1], nh, out=vt_loc[b:b + 1])
            hs.append((self.backend.all_gather_into(kg[b], kloc[b:b + 1]), self.backend.all_gather_into(vtg[b], vt_loc[b:b + 1])))
        ops.gemm(xn, lw["qkv_w"][:D], lw["qkv_b"][:D], out=q)                       # overlaps the exchange
        ops.rmsnorm_rope(q, lw["qn"], cos, sin, rows_per_batch=Ltok, eps=eps)
        for b in range(B):
            hs[b][0].wait()
            hs[b][1].wait()
            net._timed("self_attn", ops.flash_attn, q[b:b + 1], kg[b, 0], vtg[b, 0], out=att[b:b + 1], n_seg=self.size,
                       k_seg_stride=kg.stride(1), vt_seg_stride=vtg.stride(1))
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
