"""GPU tests of the sequence-parallel path at the sizes BASELINE configs 3 and 5 run it, and first contact with RCCL.

  * 8 (Ulysses) and 2 (all-gather) virtual ranks -- threads sharing the one GPU through scail_amd.parallel.ThreadBackend -- run the REAL
    multi-rank data path at full size (14B width, 512x896x81f: L = 48 832 tokens; 2 layers so the run stays short; rank slabs of 6 104 rows,
    5 heads per rank at 8 ranks): H-chunked latents, rank-shifted RoPE, the C executor's sequence-parallel block (scail_dit_step_sp:
    norm + RoPE writing the send layout, head <-> sequence all-to-alls through the exchange callback, full-length attention on the rank's
    heads, way back, slabs -> rows), gather to rank 0 -- against the single-rank evaluation of the same network.  The multi-character
    extension (BASELINE config 5: 2 ref + 2 pose streams, L = 60 032) runs the same way at 8 ranks.
  * the C executor's sequence-parallel path is BIT-IDENTICAL to the per-op host path (scail_amd.parallel.SequenceParallel.self_attention),
    at toy and at full size.
  * RCCL: init_process_group("nccl", world_size = 1) on the box + SequenceParallel.self_check + the two collectives the layer exchange uses,
    with HSA_ENABLE_IPC_MODE_LEGACY as bench.py sets it (reference: sat/mpu/ulysses_attn_layer.py:41-110, diffusion_video.py:495-585).
"""
import json
import os
import subprocess
import sys
import threading

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DEV = "cuda"

P14B = dict(hidden_size=5120, num_attention_heads=40, inner_hidden_size=13824, text_dim=4096, time_freq_dim=256, time_embed_dim=5120)


def _mk(params, layers, seed=1234):
    from scail_amd.dit import DiffusionTransformer
    return DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=81, latent_width=300, latent_height=300,
                                share_adaln=True, use_i2v_clip=True, device=DEV, init_seed=seed, num_layers=layers, **params)


def _run_ranks(world, mode, mk, inputs, chunk_dim, use_c, cfg_pair=False):
    """one network evaluation on `world` virtual ranks; returns the result gathered on rank 0"""
    from scail_amd.parallel import SequenceParallel, ThreadBackend
    x, t, ctx, ref, pose, clip = inputs
    shared = ThreadBackend.Shared(world)
    outs, errs = [None] * world, []

    def run(rk):
        try:
            torch.cuda.set_device(0)
            n = mk()
            n.use_c_step = use_c
            sp = SequenceParallel(ThreadBackend(shared, rk), mode=mode)
            n.sp = sp
            sp.check_latent(x.shape[3], x.shape[4], chunk_dim)
            ch = lambda tt: sp.chunk(tt, chunk_dim)
            o = n.forward_f32(ch(x), t, ctx, None, concat_images=torch.zeros(1, device=DEV), image_clip_features=clip,
                              ref_concat=ch(ref), concat_smpl_render=ch(pose), chunk_dim=chunk_dim, cfg_pair=cfg_pair)
            if use_c:
                assert n._cstep is not None, "the C executor must have run"
            outs[rk] = sp.gather_to_rank0(o, chunk_dim)
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=run, args=(rk,)) for rk in range(world)]
    [tt.start() for tt in th]
    [tt.join() for tt in th]
    assert not errs, errs
    torch.cuda.synchronize()
    return outs[0]


def _inputs(T, H, W, n_char, text_dim, Lt, Lc, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, T, 16, H, W, generator=g).to(DEV)
    ref = torch.randn(1, n_char, 16, H, W, generator=g).to(DEV).to(torch.bfloat16)
    pose = torch.randn(1, n_char * T, 16, H // 2, W // 2, generator=g).to(DEV).to(torch.bfloat16)
    ctx = torch.randn(2, Lt, text_dim, generator=g).to(DEV).to(torch.bfloat16)
    clip = torch.randn(1, Lc, 1280, generator=g).to(DEV).to(torch.bfloat16)
    t = torch.tensor([700.0, 700.0], device=DEV)
    return x, t, ctx, ref, pose, clip


@pytest.mark.parametrize("world,mode,n_char", [(8, "ulysses", 1), (2, "allgather", 1), (8, "ulysses", 2), (4, "ulysses", 1)])
def test_sequence_parallel_fullsize_virtual_ranks(world, mode, n_char):
    T, H, W = 21, 64, 112
    inputs = _inputs(T, H, W, n_char, 4096, 512, 257)
    x, t, ctx, ref, pose, clip = inputs
    mk = lambda: _mk(P14B, 2)
    net = mk()
    single = net.forward_f32(x, t, ctx, None, concat_images=torch.zeros(1, device=DEV), image_clip_features=clip, ref_concat=ref, concat_smpl_render=pose)
    torch.cuda.synchronize()
    del net
    got = _run_ranks(world, mode, mk, inputs, 3, use_c=True)
    d = (got - single).abs()
    scale = float(single.abs().mean())
    print(f"SP {world} x {mode}, {n_char} character(s), L = {(n_char + T) * (H // 2) * (W // 2) + n_char * T * (H // 4) * (W // 4)}: "
          f"max |d| {float(d.max()):.4f}, mean |d| {float(d.mean()):.5f}, |ref| mean {scale:.3f}")
    # bf16 re-association only (rank-major key order, per-rank GEMM tiling): measured max 0.031, mean 0.0025 on |ref| mean 0.87
    assert torch.isfinite(got).all()
    assert float(d.mean()) <= 6e-3 * max(scale, 1.0) and float(d.max()) <= 0.125
    a, b = got.flatten().double(), single.flatten().double()
    assert float((a @ b) / (a.norm() * b.norm())) >= 0.9999
    if n_char == 1:
        # the executor's sequence-parallel block enqueues the kernels of the per-op host path in the same order: identical bits
        host = _run_ranks(world, mode, mk, inputs, 3, use_c=False)
        assert torch.equal(got, host), f"C executor vs per-op host path: max |d| {float((got - host).abs().max())}"


@pytest.mark.parametrize("world,mode,chunk_dim,side", [(2, "allgather", 3, None), (2, "ulysses", 3, None), (4, "ulysses", 4, None), (4, "allgather", 3, None),
                                                       (2, "ulysses", 3, "1"), (4, "ulysses", 4, "1")])
def test_sp_c_executor_equals_per_op_path_small(world, mode, chunk_dim, side, monkeypatch):
    """toy width (4 heads), both exchange modes, H- and W-split, without and with the two side streams of the ulysses exchange
    (SCAIL_SP_SIDE_STREAMS=1: fork / join inside scail_dit_block_sp; off by default since round 5)"""
    if side is not None:
        monkeypatch.setenv("SCAIL_SP_SIDE_STREAMS", side)
    cfgd = dict(hidden_size=512, num_attention_heads=4, inner_hidden_size=1024, text_dim=64, time_freq_dim=256, time_embed_dim=512)
    T, H, W = (2, 16, 32) if chunk_dim == 3 else (2, 32, 16)
    inputs = _inputs(T, H, W, 1, 64, 12, 5, seed=11)
    mk = lambda: _mk(cfgd, 3, seed=77)
    c = _run_ranks(world, mode, mk, inputs, chunk_dim, use_c=True)
    host = _run_ranks(world, mode, mk, inputs, chunk_dim, use_c=False)
    assert torch.equal(c, host)
    x, t, ctx, ref, pose, clip = inputs
    single = mk().forward_f32(x, t, ctx, None, concat_images=torch.zeros(1, device=DEV), image_clip_features=clip, ref_concat=ref, concat_smpl_render=pose)
    torch.testing.assert_close(c, single, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("world,mode,use_c", [(4, "ulysses", True), (4, "ulysses", False), (2, "allgather", True), (2, "allgather", False)])
def test_exchange_is_one_collective_each_way(world, mode, use_c, monkeypatch):
    """The layer exchange is ONE collective per direction and CFG element: q | k | v travel side by side in one all-to-all (the
    reference issues three, sat/mpu/ulysses_attn_layer.py:65-80) + one back = 4 per layer at B = 2 (round 5: 8); all-gather mode:
    one gather of k | v per element = 2 per layer (round 5: 4).  Counted on every virtual rank, C executor and per-op path."""
    from scail_amd.parallel import ThreadBackend
    counts, lock = {}, threading.Lock()

    def counted(name):
        orig = getattr(ThreadBackend, name)

        def f(self, *a, **k):
            with lock:
                counts[name, self.rank] = counts.get((name, self.rank), 0) + 1
            return orig(self, *a, **k)
        return f

    monkeypatch.setattr(ThreadBackend, "all_to_all", counted("all_to_all"))
    monkeypatch.setattr(ThreadBackend, "all_gather_into", counted("all_gather_into"))
    cfgd = dict(hidden_size=512, num_attention_heads=4, inner_hidden_size=1024, text_dim=64, time_freq_dim=256, time_embed_dim=512)
    layers = 3
    _run_ranks(world, mode, lambda: _mk(cfgd, layers, seed=77), _inputs(2, 16, 32, 1, 64, 12, 5, seed=11), 3, use_c=use_c)
    for rk in range(world):
        a2a, ag = counts.get(("all_to_all", rk), 0), counts.get(("all_gather_into", rk), 0)
        assert (a2a, ag) == ((layers * 2 * 2, 0) if mode == "ulysses" else (0, layers * 2)), (rk, a2a, ag)


@pytest.mark.parametrize("world,mode", [(1, None), (8, "ulysses"), (4, "ulysses"), (2, "allgather")])
def test_cfg_pair_is_bit_identical_at_full_size(world, mode):
    """SCAIL_DIT_CFG_PAIR (include/scail_dit.h; guiders.py:41-57, dit...:1009-1042): the sampler's batch is one latent twice, so layer 0 up to
    its first cross attention is evaluated once and copied -- same bits as the plain B = 2 evaluation, on one rank and on the
    sequence-parallel executor (where layer 0 then exchanges element 0 only), at 14B width and L = 48 832 with 3 layers (so that layer 0,
    a middle layer and the row-pruned last layer all run)."""
    T, H, W = 21, 64, 112
    x, t, ctx, ref, pose, clip = _inputs(T, H, W, 1, 4096, 512, 257, seed=5)
    x = torch.cat([x[:1], x[:1]]).contiguous()
    inputs = (x, t, ctx, ref, pose, clip)
    mk = lambda: _mk(P14B, 3)
    outs = []
    for pair in (False, True):
        if world == 1:
            net = mk()
            outs.append(net.forward_f32(x, t, ctx, None, concat_images=torch.zeros(1, device=DEV), image_clip_features=clip, ref_concat=ref,
                                        concat_smpl_render=pose, cfg_pair=pair))
            assert net._cstep is not None
            torch.cuda.synchronize()
            del net
        else:
            outs.append(_run_ranks(world, mode, mk, inputs, 3, use_c=True, cfg_pair=pair))
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().mean()) > 1e-3
    assert not torch.equal(outs[0][0], outs[0][1]), "the two elements must differ (different text conditioning)"
    assert torch.equal(outs[0], outs[1]), f"cfg_pair changed the result: max |d| {float((outs[0] - outs[1]).abs().max())}"


def test_exchange_callback_error_reaches_the_caller():
    """an exception inside the exchange callback aborts the executor call and is re-raised by the binding (not swallowed by ctypes)"""
    from scail_amd.parallel import SequenceParallel

    class Broken:
        rank, size = 0, 2

        def all_gather_into(self, out, inp, async_op=True):
            raise RuntimeError("fabric down")

        all_to_all = all_gather_into

    cfgd = dict(hidden_size=256, num_attention_heads=2, inner_hidden_size=512, text_dim=64, time_freq_dim=256, time_embed_dim=256)
    net = _mk(cfgd, 1)
    net.sp = SequenceParallel(Broken(), mode="allgather")
    x, t, ctx, ref, pose, clip = _inputs(2, 16, 16, 1, 64, 12, 5)
    with pytest.raises(RuntimeError, match="fabric down"):
        net.forward_f32(x[:, :, :, :8], t, ctx, None, concat_images=torch.zeros(1, device=DEV), image_clip_features=clip,
                        ref_concat=ref[:, :, :, :8], concat_smpl_render=pose[:, :, :, :4], chunk_dim=3)


def test_slabs_to_rows_inverts_the_slab_layout():
    from scail_amd import lib as L
    L.load()
    g = torch.Generator(device=DEV).manual_seed(0)
    for rows, D, n in ((37, 512, 4), (6104, 5120, 8), (100, 256, 1)):
        x = torch.randn(rows, D, device=DEV, generator=g).to(torch.bfloat16)
        slabs = x.view(rows, n, D // n).permute(1, 0, 2).contiguous()
        y = torch.full((rows, D + 8), 7.0, device=DEV, dtype=torch.bfloat16)
        L.call("scail_slabs_to_rows", slabs.data_ptr(), D // n, rows * (D // n), y.data_ptr(), D + 8, rows, D, torch.cuda.current_stream().cuda_stream)
        assert torch.equal(y[:, :D], x) and bool((y[:, D:] == 7.0).all())


_RCCL_SCRIPT = r"""
import json, os, sys
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # as bench.py sets it (the pool's host driver supports only dmabuf IPC)
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from scail_amd import lib
from scail_amd.parallel import SequenceParallel, TorchDistBackend
lib.load()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sp = SequenceParallel(TorchDistBackend(None))
info = sp.self_check(dev)
# the two collectives of the layer exchange, at one rank's config-3 message size (6 104 tokens x 640 columns, bf16), async like the product
a = torch.randn(1, 6104, 640, device=dev).to(torch.bfloat16)
b = torch.empty_like(a)
sp.backend.all_to_all(b, a, async_op=True).wait()
c = torch.empty(1, 6104, 640, device=dev, dtype=torch.bfloat16)
sp.backend.all_gather_into(c, a[0]).wait()
t = torch.ones(4, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
info["a2a_ok"] = bool(torch.equal(a, b))
info["allgather_ok"] = bool(torch.equal(c, a))
info["ipc_mode_legacy"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
print(json.dumps(info))
dist.destroy_process_group()
"""


def test_rccl_single_rank_init_and_self_check():
    """RCCL has to load, create a communicator on this box and move data before a scaling job depends on it: a one-rank process group is
    the part of that a 1-GPU box can run (ncclCommInitRank, the collectives' launch path, stream ordering with torch)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT % ROOT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    info = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print("RCCL single-rank check:", info)
    assert info["backend"] == "nccl" and info["ranks"] == 1 and info["rccl_version"]
    assert info["collectives_verified"] == ["all_to_all", "all_gather", "broadcast"] and info["a2a_ok"] and info["allgather_ok"]


_RCCL2_SCRIPT = r"""
import json, os, sys
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from scail_amd import lib
from scail_amd.dit import DiffusionTransformer
from scail_amd.parallel import SequenceParallel, TorchDistBackend
lib.load()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("SCAIL_TEST_BACKEND", "nccl")
# nccl: one GPU per rank, the exchange is RCCL over xGMI.  gloo: the TEST vehicle of scail_amd.parallel.TorchDistBackend -- both ranks on GPU 0,
# the exchange staged through the host -- so that everything but the transport runs on a 1-GPU box
local = rank if backend == "nccl" else 0
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
LAYERS = int(os.environ.get("SCAIL_TEST_LAYERS", "2"))
P = dict(hidden_size=5120, num_attention_heads=40, inner_hidden_size=13824, text_dim=4096, time_freq_dim=256, time_embed_dim=5120)
mk = lambda: DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=81, latent_width=300, latent_height=300,
                                  share_adaln=True, use_i2v_clip=True, device=dev, init_seed=1234, num_layers=LAYERS, **P)
T, H, W = 21, 64, 112
g = torch.Generator().manual_seed(1)
x = torch.randn(1, T, 16, H, W, generator=g).repeat(2, 1, 1, 1, 1).to(dev)
ref = torch.randn(1, 1, 16, H, W, generator=g).to(dev).to(torch.bfloat16)
pose = torch.randn(1, T, 16, H // 2, W // 2, generator=g).to(dev).to(torch.bfloat16)
ctx = torch.randn(2, 512, 4096, generator=g).to(dev).to(torch.bfloat16)
clip = torch.randn(1, 257, 1280, generator=g).to(dev).to(torch.bfloat16)
t = torch.tensor([700.0, 700.0], device=dev)
res = {}
for mode in ("ulysses", "allgather"):
    net = mk()
    sp = SequenceParallel(TorchDistBackend(None), mode=mode)
    info = sp.self_check(dev)
    net.sp = sp
    ch = lambda tt: sp.chunk(tt, 3)
    o = net.forward_f32(ch(x), t, ctx, None, concat_images=torch.zeros(1, device=dev), image_clip_features=clip, ref_concat=ch(ref),
                        concat_smpl_render=ch(pose), chunk_dim=3, cfg_pair=True)
    assert net._cstep is not None
    full = sp.gather_to_rank0(o, 3)
    torch.cuda.synchronize()
    if rank == 0:
        net.sp = None
        single = net.forward_f32(x, t, ctx, None, concat_images=torch.zeros(1, device=dev), image_clip_features=clip, ref_concat=ref,
                                 concat_smpl_render=pose)
        d = (full - single).abs()
        a, b = full.flatten().double(), single.flatten().double()
        res[mode] = dict(max=float(d.max()), mean=float(d.mean()), scale=float(single.abs().mean()),
                         cos=float((a @ b) / (a.norm() * b.norm())), rccl=info.get("rccl_version"), backend=info.get("backend"))
    del net
    dist.barrier()
if rank == 0:
    print(json.dumps(res))
dist.destroy_process_group()
"""


def _two_process_ranks(tmp_path, backend, port):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", SCAIL_TEST_BACKEND=backend)
    script = tmp_path / "ranks2.py"
    script.write_text(_RCCL2_SCRIPT % ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(f"2 ranks ({backend}):", res)
    for mode in ("ulysses", "allgather"):
        m = res[mode]
        assert m["backend"] == backend
        assert m["mean"] <= 6e-3 * max(m["scale"], 1.0) and m["max"] <= 0.125 and m["cos"] >= 0.9999, (mode, m)
    return res


def test_two_process_ranks_fullsize_layers_host_staged_exchange(tmp_path):
    """TWO PROCESSES (torch.distributed.run, as the driver launches a scaling job) run scail_dit_step_sp with SCAIL_DIT_CFG_PAIR on 2 full-width
    layers at L = 48 832, both exchange modes, against the single-rank evaluation -- with the gloo vehicle (both ranks on GPU 0, exchange
    staged through the host), so that every line of the script below except the transport is exercised on a 1-GPU box."""
    _two_process_ranks(tmp_path, "gloo", 29673)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: two real ranks exchanging over RCCL / xGMI")
def test_two_real_ranks_over_rccl_fullsize_layers_and_bench_line(tmp_path):
    """The first multi-GPU box that runs this suite exercises what a 1-GPU box cannot: the SAME script with backend nccl -- scail_dit_step_sp
    on two GPUs whose per-layer exchange is a real RCCL all_to_all_single / all_gather_into_tensor (sat/mpu/ulysses_attn_layer.py:41-110,
    all_to_all.py:15-108, diffusion_video.py:495-585); then bench.py --gpus 2 on the tiny config through torch.distributed.run exactly as
    the driver launches it."""
    res = _two_process_ranks(tmp_path, "nccl", 29671)
    assert res["ulysses"]["rccl"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29672", "bench.py", "--gpus", "2", "--config", "tiny", "--steps", "2", "--warmup", "1",
                        "--no-vae", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["path"].startswith("scail_dit_step_sp")
