"""world_size-2 gloo test (CPU) of the sequence-parallel HOST logic: latent chunking, rank-shifted RoPE
window, K/V all-gather through scail_amd.parallel.TorchDistBackend, gather to rank 0.  The per-rank
compute is the CPU oracle (checker) -- the HIP kernels cannot run here; the same exchange with the
real kernels is covered on the GPU by tests/test_dit_gpu.py::test_sequence_parallel_emulated_*."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, golden, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import scail_oracle as O
    from scail_amd.parallel import SequenceParallel, TorchDistBackend
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(golden).items()}
    cfg = O.DiTConfig(**O.TINY)
    sd = O.make_state_dict(cfg, seed=int(g["seed"]))
    sp = SequenceParallel(TorchDistBackend(None))
    info = sp.self_check(torch.device("cpu"))              # every collective of the layer exchange, on rank-stamped data
    assert info["ranks"] == world and info["backend"] == "gloo", info
    x = g["x"].clone()
    if rank != 0:
        x.zero_()
    sp.broadcast(x)                                        # noise broadcast, diffusion_video.py:486-493
    cd = sp.chunk_dim_for(x.shape[-2:]) if x.shape[-2] != x.shape[-1] else 3
    sp.check_latent(x.shape[3], x.shape[4], cd)

    def kv_gather(k, v):                                   # the per-layer exchange (parallel.py)
        kv = torch.cat([k, v], dim=-1).contiguous()        # ONE collective: k | v side by side, like the product's message layout
        g2 = torch.empty(world, *kv.shape)
        sp.backend.all_gather_into(g2, kv, async_op=False)
        kg, vg = torch.cat(list(g2), dim=2).split(k.shape[-1], dim=-1)
        return kg, vg

    shift = rank * (x.shape[cd] // world // 2)            # rank-shifted RoPE window along the split axis (dit...:1578-1585)
    out = O.dit_forward(cfg, sd, sp.chunk(x, cd), g["t"], g["ctx"], sp.chunk(g["ref"], cd), sp.chunk(g["pose"], cd),
                        g["clip"], H_shift=shift if cd == 3 else 0, W_shift=shift if cd == 4 else 0, kv_gather=kv_gather)
    full = sp.gather_to_rank0(out, cd)
    # the ulysses exchange primitive: out[s] <- rank s's inp[my rank]
    inp = torch.arange(world * 3, dtype=torch.float32).reshape(world, 3) + 100 * rank
    got = torch.empty_like(inp)
    sp.backend.all_to_all(got, inp)
    assert torch.equal(got, torch.stack([torch.arange(3, dtype=torch.float32) + 3 * rank + 100 * s for s in range(world)]))
    if rank == 0:
        q.put((full - g["out"]).abs().max().item())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fixture", ["dit_tiny.npz", "dit_shapes.npz"])     # square latent: H split; portrait 12 x 8: W split
def test_sp2_gloo_matches_reference_golden(golden_dir, fixture):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + len(fixture)) % 400
    procs = [ctx.Process(target=_worker, args=(r, 2, port, os.path.join(golden_dir, fixture), q)) for r in range(2)]
    [p.start() for p in procs]
    err = q.get(timeout=240)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert err < 5e-5, err          # reference itself: SP=2 vs SP=1 differ by 7.2e-7


def test_check_latent_rejects_bad_split():
    from scail_amd.parallel import SequenceParallel

    class B:
        rank, size = 0, 8
    sp = SequenceParallel(B())
    sp.check_latent(64, 112, 3)                 # config 3: 64 rows / 8 ranks = 8 rows each
    with pytest.raises(ValueError):
        sp.check_latent(60, 112, 3)
