"""GPU parity tests, op level: every HIP kernel of libscail_hip.so (called through the C ABI via
scail_amd.ops) against the CPU oracle (oracle/scail_oracle.py) on seeded inputs.

Tolerances: bf16 storage + fp32 accumulate vs the fp32 oracle ->
rtol 2e-2 / atol 2e-2 on O(1) activations; tighter where only one rounding separates the two."""
import math

import pytest
import torch

from oracle import scail_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from scail_amd import lib, ops as _ops
    lib.load()          # loud failure when the HIP library is missing
    assert torch.cuda.is_available()
    return _ops


def bfr(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bfr(torch.randn(*shape, generator=g) * scale)


def gpu_bf16(x):
    return x.to(torch.bfloat16).to(DEV)


def close(a, b, rtol=2e-2, atol=2e-2, msg=""):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), f"{msg} mismatches={int(bad.sum())}/{bad.numel()} max_err={float(err.max()):.4g} " \
                          f"at {tuple(torch.nonzero(bad)[0].tolist()) if bad.any() else None} ref_absmax={float(b.abs().max()):.4g}"


# ------------------------------------------------------------------------------------------------
@pytest.fixture(params=[0, 128, 256, 257, 260, 261, 262])
def gemm_tile(request):
    """0 = the product dispatch (by shape).  The forced tile instantiations exist only in the measurement build
    (SCAIL_ABLATIONS=1, include/scail_hip_ablation.h) and are skipped without it."""
    from scail_amd import lib as L
    if request.param == 0:
        yield 0
        return
    if not L.ABLATIONS:
        pytest.skip("kernel variant of the measurement build (run with SCAIL_ABLATIONS=1)")
    L.tune_set("gemm_tile", request.param)
    yield request.param
    L.tune_set("gemm_tile", 0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 128), (1000, 384, 256), (77, 64, 320), (256, 512, 5120)])
@pytest.mark.parametrize("epi", ["bias", "gelu_tanh", "gelu_erf"])
def test_gemm(ops, gemm_tile, M, N, K, epi):
    from scail_amd import lib as L
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
    ref = x @ w.t() + b
    code = {"bias": L.EPI_BIAS, "gelu_tanh": L.EPI_GELU_TANH, "gelu_erf": L.EPI_GELU_ERF}[epi]
    if epi == "gelu_tanh":
        ref = O.gelu_tanh(ref)
    elif epi == "gelu_erf":
        ref = torch.nn.functional.gelu(ref)
    y = ops.gemm(gpu_bf16(x), gpu_bf16(w), b.to(DEV), epilogue=code)
    close(y, ref, rtol=1e-2, atol=1e-2, msg=f"gemm {M}x{N}x{K} {epi}")


def test_gemm_transpose_detecting(ops, gemm_tile):
    """A = I with an asymmetric W catches an output written transposed (guide rule 16)."""
    K = N = 128
    x = torch.eye(K)
    w = bfr(torch.arange(N * K, dtype=torch.float32).reshape(N, K) / (N * K))
    y = ops.gemm(gpu_bf16(x), gpu_bf16(w))
    close(y, w.t(), rtol=1e-2, atol=1e-3, msg="gemm identity")


def test_gemm_strided_and_residual(ops, gemm_tile):
    from scail_amd import lib as L
    B, Lr, K, N = 2, 150, 128, 256
    xbig = rnd(B, Lr, 3 * K, seed=4)
    x = xbig[..., K:2 * K]                                   # column-slice view, lda = 3K
    w, b = rnd(N, K, seed=5, scale=0.1), rnd(N, seed=6)
    resid = rnd(B, Lr, N, seed=7)
    gate = rnd(B, 6 * N, seed=8)
    ref = resid + gate[:, None, 2 * N:3 * N] * (x @ w.t() + b)
    xg = gpu_bf16(xbig)[..., K:2 * K]
    h = gpu_bf16(resid)
    g = gate.to(DEV)
    ops.gemm(xg, gpu_bf16(w), b.to(DEV), out=h, epilogue=L.EPI_RESID, resid=h, gate=g[:, 2 * N:3 * N], rows_per_batch=Lr)
    close(h, ref, msg="gemm resid+gate in place")
    h2 = gpu_bf16(resid)
    ops.gemm(xg, gpu_bf16(w), b.to(DEV), out=h2, epilogue=L.EPI_RESID, resid=h2)
    close(h2, resid + (x @ w.t() + b), msg="gemm ungated residual")


@pytest.mark.parametrize("M,N,K", [(2048 + 136, 512, 192), (4096, 768, 320), (2600 + 8, 256, 128), (2 * 1100, 256, 448)])
@pytest.mark.parametrize("epi", ["bias", "nobias", "gelu_tanh", "resid+gate", "resid"])
def test_gemm4_vs_oracle(ops, M, N, K, epi):
    """the generated 4-wave kernels (csrc/gemm4.s; scail_gemm_kernel_for == 4): ragged last m-tile (136 / 8 valid rows), 1-3 n-tiles,
    odd and even k-tile counts (3, 5, 2, 7), all four epilogues + NULL bias, strided x / y views, gate rows per batch not a multiple of
    the tile; against fp32 and against the kernels of csrc/gemm.hip on the same inputs (same accumulation order: equal to the last bit
    up to the epilogue's rounding)."""
    from scail_amd import lib as L
    xbig = rnd(M, K + 64, seed=1)
    x = xbig[:, 64:]                                                  # lda = K + 64
    w, b = rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
    resid, gate = rnd(M, N, seed=4), rnd(2, N, seed=5)
    rpb = M // 2
    bias = None if epi == "nobias" else b
    ref = x @ w.t() + (0 if bias is None else bias)
    code = L.EPI_BIAS
    kw = {}
    if epi == "gelu_tanh":
        ref, code = O.gelu_tanh(ref), L.EPI_GELU_TANH
    if epi.startswith("resid"):
        code = L.EPI_RESID
        if epi == "resid+gate":
            ref = ref * gate.repeat_interleave(rpb, 0)
        ref = resid + ref
    xg, wg = gpu_bf16(xbig)[:, 64:], gpu_bf16(w)
    outs = []
    for on in (1, 0):
        L.set_option("gemm4", on)
        try:
            ybig = torch.full((M + 1, N + 8), 3.0, device=DEV, dtype=torch.bfloat16)
            y = ybig[:M, :N]                                           # ldc = N + 8; one guard row, 8 guard columns
            if epi.startswith("resid"):
                y.copy_(gpu_bf16(resid))
                kw = dict(resid=y, gate=gate.to(DEV) if epi == "resid+gate" else None, rows_per_batch=rpb if epi == "resid+gate" else 0)
            which = L.load().scail_gemm_kernel_for(xg.stride(0), y.stride(0), y.stride(0) if kw else 0, M, N, K, code)
            assert which == (4 if on else 0)
            ops.gemm(xg, wg, None if bias is None else bias.to(DEV), out=y, epilogue=code, **kw)
        finally:
            L.set_option("gemm4", 1)
        close(y, ref, rtol=1e-2, atol=1e-2, msg=f"gemm4={on} {M}x{N}x{K} {epi}")
        assert (ybig[M] == 3.0).all() and (ybig[:, N:] == 3.0).all(), "writes outside the M x N result"
        outs.append(y.float().clone())
    close(outs[0], outs[1], rtol=8e-3, atol=1e-3, msg="gemm4 vs csrc/gemm.hip")


def test_gemm_errors_are_loud(ops):
    from scail_amd import lib as L
    x, w = gpu_bf16(rnd(8, 72)), gpu_bf16(rnd(16, 72))
    with pytest.raises(L.ScailHipError, match="multiple of 64"):
        ops.gemm(x, w)
    with pytest.raises(L.ScailHipError, match="GPU"):
        ops.gemm(rnd(8, 64).to(torch.bfloat16), rnd(16, 64).to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [128, 256, 5120])
def test_ln_modulate(ops, D):
    B, Ls = 2, 37
    x, sh, sc = rnd(B, Ls, D, seed=1, scale=2.0) + 0.5, rnd(B, D, seed=2), rnd(B, D, seed=3, scale=0.3)
    ref = O.modulate(O.layer_norm(x, 1e-6), sh[:, None], sc[:, None])
    y = ops.ln_modulate(gpu_bf16(x), sh.to(DEV), sc.to(DEV))
    close(y, ref, msg=f"ln_modulate D={D}")
    # noise-token slice variant used by the final layer
    y2 = ops.ln_modulate(gpu_bf16(x), sh.to(DEV), sc.to(DEV), rows_out=20, src_row_offset=5)
    close(y2, ref[:, 5:25], msg="ln_modulate slice")


def test_layernorm_affine(ops):
    rows, D = 53, 1280
    x, w, b = rnd(rows, D, seed=1, scale=3.0), 1 + 0.1 * rnd(D, seed=2), rnd(D, seed=3)
    y = ops.layernorm_affine(gpu_bf16(x), w.to(DEV), b.to(DEV), eps=1e-5)
    close(y, O.layer_norm(x, 1e-5, w, b), msg="layernorm_affine")


@pytest.mark.parametrize("heads", [1, 2, 40])
def test_rmsnorm_rope(ops, heads):
    cfg = O.DiTConfig(hidden_size=128 * heads, num_attention_heads=heads, latent_height=32, latent_width=32, num_frames=13)
    T, Hp, Wp = 3, 6, 4
    cos, sin = O.rope_tables(cfg, T, Hp, Wp)
    Ltok = cos.shape[0]
    B, D = 2, 128 * heads
    x, w = rnd(B, Ltok, 3 * D, seed=1), 1 + 0.1 * rnd(D, seed=2)
    q = x[..., D:2 * D]
    ref = O.rms_norm(q, w, 1e-6)
    ref_rope = O._merge(O.apply_rope(O._heads(ref, heads), cos, sin))
    xg = gpu_bf16(x)
    out = torch.empty(B, Ltok, D, device=DEV, dtype=torch.bfloat16)
    ops.rmsnorm_rope(xg[..., D:2 * D], w.to(DEV), out=out)
    close(out, ref, msg="rmsnorm (no rope)")
    ops.rmsnorm_rope(xg[..., D:2 * D], w.to(DEV), cos[:, 0::2].contiguous().to(DEV), sin[:, 0::2].contiguous().to(DEV))
    close(xg[..., D:2 * D], ref_rope, msg="rmsnorm+rope in place")
    close(xg[..., :D], x[..., :D], rtol=0, atol=0, msg="neighbour columns untouched")
    # out_scale (queries in log2 units for the attention): one rounding of scale * result; 1.0 is bit-identical to the plain entry
    o1 = torch.empty_like(out)
    ops.rmsnorm_rope(gpu_bf16(x)[..., D:2 * D], w.to(DEV), cos[:, 0::2].contiguous().to(DEV), sin[:, 0::2].contiguous().to(DEV), out=o1,
                     out_scale=ops.ATTN_LOG2_SCALE)
    close(o1, ref_rope * ops.ATTN_LOG2_SCALE, rtol=1e-2, atol=2e-3, msg="rmsnorm+rope, scaled output")
    o2 = torch.empty_like(out)
    ops.rmsnorm_rope(gpu_bf16(x)[..., D:2 * D], w.to(DEV), cos[:, 0::2].contiguous().to(DEV), sin[:, 0::2].contiguous().to(DEV), out=o2, out_scale=1.0)
    assert torch.equal(o2, xg[..., D:2 * D])


@pytest.mark.parametrize("D", [1536, 5120])
def test_row_wave_kernels_agree_with_the_block_kernels(ops, D):
    """option row_wave (include/scail_hip.h): the one-wave-per-row LayerNorm / RMSNorm + RoPE kernels against the block-per-row kernels on the
    same inputs -- same arithmetic up to the order of the fp32 sums: at most one bf16 step, on a small fraction of the elements; ragged row
    counts (rows % 4 != 0), strided input, slab output."""
    from scail_amd import lib as L
    heads = D // 128
    B, Ls = 2, 203
    x = gpu_bf16(rnd(B, Ls, 3 * D, seed=4, scale=1.5))
    sh, sc = rnd(B, D, seed=5).to(DEV), rnd(B, D, seed=6, scale=0.3).to(DEV)
    w, b = (1 + 0.1 * rnd(D, seed=7)).to(DEV), rnd(D, seed=8).to(DEV)
    ang = torch.rand(Ls, 64, generator=torch.Generator().manual_seed(9)) * 6.28
    cos, sin = torch.cos(ang).to(DEV).contiguous(), torch.sin(ang).to(DEV).contiguous()
    xd = x[..., :D].contiguous()

    def run():
        slabs = torch.empty(heads // 4, B * Ls, 4 * 128, device=DEV, dtype=torch.bfloat16)
        ops.rmsnorm_rope_slabs(x.view(B * Ls, 3 * D)[:, D:2 * D], w, slabs, cos, sin, rows_per_batch=Ls, out_scale=0.1275)
        return (ops.ln_modulate(xd, sh, sc), ops.ln_modulate(xd, sh, sc, rows_out=101, src_row_offset=7), ops.layernorm_affine(xd.view(B * Ls, D), w, b),
                ops.rmsnorm_rope(x[..., D:2 * D], w, cos, sin, out=torch.empty(B, Ls, D, device=DEV, dtype=torch.bfloat16), rows_per_batch=Ls), slabs)

    L.set_option("row_wave", 0)
    try:
        old = run()
    finally:
        L.set_option("row_wave", 1)
    new = run()
    for a, c in zip(old, new):
        d = (a.float() - c.float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * max(1.0, float(a.float().abs().max())), float(d.max())
        assert float((d > 0).float().mean()) < 2e-3


@pytest.mark.parametrize("heads,n_slabs", [(4, 2), (8, 8), (4, 1)])
def test_rmsnorm_rope_slabs(ops, heads, n_slabs):
    """scail_rmsnorm_rope_slabs: the q / k norm + RoPE (and the plain copy of v) written as one dense column slab per destination rank
    = the send layout of the Ulysses all-to-all (sat/mpu/ulysses_attn_layer.py:65-80: permute + contiguous), bit-identical to the
    row-major kernel followed by that permute."""
    cfg = O.DiTConfig(hidden_size=128 * heads, num_attention_heads=heads, latent_height=32, latent_width=32, num_frames=13)
    cos, sin = O.rope_tables(cfg, 3, 6, 4)
    cg, sg = cos[:, 0::2].contiguous().to(DEV), sin[:, 0::2].contiguous().to(DEV)
    Ltok, D = cos.shape[0], 128 * heads
    Dn = D // n_slabs
    x, w = rnd(Ltok, 3 * D, seed=1), (1 + 0.1 * rnd(D, seed=2)).to(DEV)
    xg = gpu_bf16(x)
    for col, wt, tabs, scale in ((0, w, (cg, sg), ops.ATTN_LOG2_SCALE), (1, w, (cg, sg), 1.0), (1, w, (None, None), 1.0), (2, None, (None, None), 1.0)):
        src = xg[:, col * D:(col + 1) * D]
        slabs = torch.full((n_slabs, Ltok, Dn), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.rmsnorm_rope_slabs(src, wt, slabs, tabs[0], tabs[1], out_scale=scale)
        if wt is None:
            rows = src.clone()
        else:
            rows = torch.empty(Ltok, D, device=DEV, dtype=torch.bfloat16)
            ops.rmsnorm_rope(src, wt, tabs[0], tabs[1], out=rows, out_scale=scale)
        assert torch.equal(slabs, rows.view(Ltok, n_slabs, Dn).permute(1, 0, 2)), (col, scale)
        # the ONE-message layout of the fused exchange (include/scail_dit.h): (dst rank, token, q | k | v) -- this third's columns of
        # every message row hold the same values, the other two thirds are left alone
        msg = torch.full((n_slabs, Ltok, 3 * Dn), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.rmsnorm_rope_slabs(src, wt, msg[:, :, col * Dn:(col + 1) * Dn], tabs[0], tabs[1], out_scale=scale)
        assert torch.equal(msg[:, :, col * Dn:(col + 1) * Dn], slabs), (col, scale)
        keep = [j for j in range(3) if j != col]
        assert all(bool((msg[:, :, j * Dn:(j + 1) * Dn] == 7.0).all()) for j in keep)
    from scail_amd import lib as L
    with pytest.raises(L.ScailHipError, match="contiguous last dim"):
        ops.rmsnorm_rope_slabs(xg[:, :D], w, torch.empty(n_slabs, Ltok + 1, Dn, device=DEV, dtype=torch.bfloat16))


def test_options_flipped_by_another_thread_while_launching(ops):
    """scail_set_option state is std::atomic (include/scail_hip.h): a thread that flips "attn4_rows" / "attn4_xcd" / "attn4_cus" while
    another thread launches attentions is defined behaviour -- every launch reads each option once, and all tile heights give the
    same bits (no exp2 overflow on this data), so every result of the launching thread is identical."""
    import threading
    from scail_amd import lib as L
    heads, Lq, Lk = 3, 1000, 2100
    D = heads * 128
    g = torch.Generator(device=DEV).manual_seed(5)
    q = (torch.randn(1, Lq, D, device=DEV, generator=g) * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
    k = torch.randn(1, Lk, D, device=DEV, generator=g).to(torch.bfloat16)
    vt = ops.transpose_v(torch.randn(1, Lk, D, device=DEV, generator=g).to(torch.bfloat16), heads)
    base = ops.flash_attn(q, k, vt, q_prescaled=True)
    stop, errs = threading.Event(), []

    def flipper():
        i = 0
        while not stop.is_set():
            L.set_option("attn4_rows", (0, 192, 256)[i % 3])
            L.set_option("attn4_xcd", i & 1)
            L.set_option("attn4_cus", (0, 200, 17)[i % 3])
            i += 1

    def launcher():
        try:
            torch.cuda.set_device(0)
            for _ in range(300):
                o = ops.flash_attn(q, k, vt, q_prescaled=True)
                if not torch.equal(o, base):
                    errs.append("result changed under a concurrent option flip")
                    break
        except Exception as e:  # pragma: no cover
            errs.append(e)

    t1, t2 = threading.Thread(target=flipper), threading.Thread(target=launcher)
    try:
        t1.start(); t2.start()
        t2.join()
    finally:
        stop.set()
        t1.join()
        for name, v in (("attn4_rows", 0), ("attn4_xcd", 1), ("attn4_cus", 0)):
            L.set_option(name, v)
    assert not errs, errs


def test_explicit_attn4_rows_wins_over_the_executor_hint(ops):
    """the executor's thread-local hint (two attentions side by side -> one 256-row launch each) applies to the automatic
    choice only; an operator's explicit option "attn4_rows" wins.  scail_flash_attn_rows_for reports the shape a launch would take."""
    from scail_amd import lib as L
    lib = L.load()
    try:
        L.set_option("attn4_rows", 192)
        assert lib.scail_flash_attn_rows_for(1, 5, 48832) == 192
        L.set_option("attn4_rows", 256)
        assert lib.scail_flash_attn_rows_for(1, 5, 48832) == 256
    finally:
        L.set_option("attn4_rows", 0)


@pytest.mark.parametrize("rows", [256, 192])
def test_restart_decision_reaches_every_wave(ops, rows):
    """The decision to run a workgroup again is taken from four per-wave flags exchanged through LDS; each wave ORs the four words and reads
    the result with v_readfirstlane.  Round 6 found on MI355X that a VALU result read by v_readfirstlane in the NEXT issue slot returns the
    register's previous value (not interlocked; asmgen/sched.py READLANE_DIST): the flag of wave 3 -- the last OR -- was lost, waves
    disagreed and stored inf / inf rows (profiles/r06_attn_restart_readlane_hazard.log).  Here: the overflowing query row sits in each of
    the four waves in turn, for both tile heights, several launches each (the failure was intermittent); every launch must count exactly
    one restart and match fp32 softmax on ALL rows."""
    from scail_amd import lib as L
    Lq, Lk = rows, 64 * 40
    g = torch.Generator(device=DEV).manual_seed(3)
    q = torch.randn(1, Lq, 128, device=DEV, generator=g)
    k0 = torch.randn(1, Lk, 128, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(1, Lk, 128, device=DEV, generator=g).to(torch.bfloat16)
    vt = ops.transpose_v(v, 1)
    qb = (q * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
    ctr = torch.zeros(1, device=DEV, dtype=torch.int32)
    per_wave = rows // 4
    L.set_option("attn4_rows", rows)
    try:
        for wave in range(4):
            for r, key in ((wave * per_wave, 64 * 10), (wave * per_wave + per_wave // 2 + 1, 64 * 3 + 17), ((wave + 1) * per_wave - 1, 64 * 20 + 33)):      # keys of hot-loop tiles (the remainder tiles track the maximum)
                k = k0.clone()
                k[0, key] = (q[0, r] * 14.0).to(torch.bfloat16)          # ~230 log2 units above the first tile's maximum: exp2 overflows
                s = qb[0].float() @ k[0].float().t()
                ref = torch.softmax(s * math.log(2.0), dim=-1) @ v[0].float()
                for rep in range(3):
                    ctr.zero_()
                    L.call("scail_flash_attn_count_restarts", ctr.data_ptr())
                    try:
                        o = ops.flash_attn(qb, k, vt, q_prescaled=True)
                        torch.cuda.synchronize()
                    finally:
                        L.call("scail_flash_attn_count_restarts", None)
                    assert int(ctr.item()) == 1, (wave, r, key, rep, int(ctr.item()))
                    assert torch.isfinite(o.float()).all(), (wave, r, key, rep)
                    torch.testing.assert_close(o[0].float(), ref, rtol=2e-2, atol=2e-2, msg=lambda m: f"wave {wave} row {r} key {key} rep {rep}: {m}")
    finally:
        L.set_option("attn4_rows", 0)


def test_restart_counter_counts_workgroups(ops):
    """scail_flash_attn_count_restarts (include/scail_hip.h): a device counter gets + 1 per workgroup of scail_attn4_m16f that leaves the
    optimistic pass (a score > ~167 log2 units above the first key tile's row maximum) and runs again; random data never restarts.  Two
    spiked query rows in different 256-row tiles of one head -> exactly 2; results equal fp32 softmax either way."""
    from scail_amd import lib as L
    heads, Lq, Lk = 2, 1024, 1536
    D = heads * 128
    g = torch.Generator(device=DEV).manual_seed(9)
    q = torch.randn(1, Lq, D, device=DEV, generator=g)
    k = torch.randn(1, Lk, D, device=DEV, generator=g)
    v = torch.randn(1, Lk, D, device=DEV, generator=g).to(torch.bfloat16)
    vt = ops.transpose_v(v, heads)
    ctr = torch.zeros(1, device=DEV, dtype=torch.int32)

    def run(qq, kk):
        ctr.zero_()
        L.call("scail_flash_attn_count_restarts", ctr.data_ptr())
        try:
            o = ops.flash_attn((qq * ops.ATTN_LOG2_SCALE).to(torch.bfloat16), kk.to(torch.bfloat16), vt, q_prescaled=True)
        finally:
            L.call("scail_flash_attn_count_restarts", None)
        torch.cuda.synchronize()
        return o, int(ctr.item())

    o0, n0 = run(q, k)
    assert n0 == 0
    k2 = k.clone()
    k2[0, 64 * 5 + 3, :128] = q[0, 10, :128] * 12.0          # head 0: query row 10 (tile 0) against key 323: ~196 log2 units above tile 0's maximum
    k2[0, 64 * 9 + 1, :128] = q[0, 700, :128] * 12.0         # head 0: query row 700 (tile 2)
    o2, n2 = run(q, k2)
    assert n2 == 2, n2
    qb, kb = (q * ops.ATTN_LOG2_SCALE).to(torch.bfloat16).float(), k2.to(torch.bfloat16).float()
    for h in range(heads):
        sl = slice(h * 128, (h + 1) * 128)
        ref = torch.softmax(qb[0, :, sl] @ kb[0, :, sl].t() * math.log(2.0), dim=-1) @ v[0, :, sl].float()
        torch.testing.assert_close(o2[0, :, sl].float(), ref, rtol=2e-2, atol=2e-2)
    o3, n3 = run(q, k2)                                       # the count is per registration (the caller zeroes its counter)
    assert n3 == 2 and torch.equal(o3, o2)


def test_rope_tables_match_oracle():
    from scail_amd import rope
    cfg = O.DiTConfig(**O.TINY)
    for hs in (0, 2):
        cos, sin = rope.build_tables(128, 4, 4, 6, H_shift=hs)
        oc, os_ = O.rope_tables(cfg, 4, 4, 6, H_shift=hs)
        assert torch.equal(cos, oc[:, 0::2]) and torch.equal(sin, os_[:, 0::2])


def test_gemm_config2_size_sampled_rows(ops):
    """The 14B step's own GEMM shapes (M = 2 x 48 832 rows; the default kernel choice): 192 sampled output rows -- including
    the ragged last m-tile -- against an fp32 matmul of the same bf16 operands, for the bias, GELU and gate + residual epilogues."""
    from scail_amd import lib as L
    M = 97664
    gen = torch.Generator(device=DEV).manual_seed(3)
    rows = torch.cat([torch.randint(0, M, (184,), device=DEV, generator=gen), torch.arange(M - 8, M, device=DEV)])
    for N, K, epi in ((15360, 5120, L.EPI_BIAS), (13824, 5120, L.EPI_GELU_TANH), (5120, 13824, L.EPI_RESID)):
        x = torch.randn(M, K, device=DEV, generator=gen).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV, generator=gen) * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device=DEV, generator=gen)
        kw = {}
        if epi == L.EPI_RESID:
            resid = torch.randn(M, N, device=DEV, generator=gen).to(torch.bfloat16)
            gate = torch.randn(2, N, device=DEV, generator=gen)
            y = resid.clone()
            kw = dict(resid=y, gate=gate, rows_per_batch=M // 2)
        else:
            y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm(x, w, b, out=y, epilogue=epi, **kw)
        ref = x[rows].float() @ w.float().t() + b
        if epi == L.EPI_GELU_TANH:
            ref = torch.nn.functional.gelu(ref, approximate="tanh")
        elif epi == L.EPI_RESID:
            ref = resid[rows].float() + gate[(rows >= M // 2).long()] * ref
        torch.testing.assert_close(y[rows].float(), ref, rtol=2e-2, atol=2e-2)
        del x, w, y


@pytest.mark.parametrize("M,N", [(6104, 1536), (12208, 512), (24 * 256, 256), (36 * 256 + 17, 768), (70 * 256, 256), (2048, 2560)])
def test_gemm4_tile_table_covers_every_tile(ops, M, N):
    """the tile -> XCD order table of the generated GEMM (csrc/gemm.hip gemm4_table): whole rows of 8 m-groups dealt round-robin, the
    remaining groups split evenly -- EVERY output tile computed exactly once for group counts below 8 (6: one sequence-parallel rank of 8),
    between 8 and 16 (12, 9 + ragged), a multiple of 8 + remainder (17.5), and a single group; all rows against an fp32 GEMM on the GPU"""
    K = 128
    g = torch.Generator(device=DEV).manual_seed(M + N)
    x = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    y = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    from scail_amd import lib as L
    assert L.load().scail_gemm_kernel_for(K, N, 0, M, N, K, L.EPI_BIAS) == 4
    ops.gemm(x, w, None, out=y)
    ref = x.float() @ w.float().t()
    assert torch.isfinite(y.float()).all(), "a tile was never written"
    torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=1e-2)


def test_release_caches_then_relaunch(ops):
    """scail_release_caches frees the per-device tile-order tables; the next launch of a grid re-creates what it needs"""
    from scail_amd import lib as L
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(2304, 128, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(512, 128, device=DEV, generator=g) / math.sqrt(128)).to(torch.bfloat16)
    y0 = ops.gemm(x, w, None)
    torch.cuda.synchronize()
    L.call("scail_release_caches")
    y1 = ops.gemm(x, w, None)
    L.call("scail_release_caches")
    L.call("scail_release_caches")                       # idempotent
    y2 = ops.gemm(x, w, None)
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
    torch.testing.assert_close(y0.float(), x.float() @ w.float().t(), rtol=1e-2, atol=1e-2)


# ------------------------------------------------------------------------------------------------
# every schedule of the attention kernel that can be selected (default = software-pipelined 4/4) must give the
# same results: lock-step (2), lock-step + LDS-DMA staging (258), 4-wave x 2 workgroups (66), software-pipelined
# 5/5 (8 | 1<<12), software-pipelined 4/4 without the LDS-store placement (8 | 6<<12)
ATTN_VARIANTS = [8 | (2 << 12), 2, 258, 66, 8 | (1 << 12), 8 | (6 << 12)]


@pytest.fixture(params=ATTN_VARIANTS)
def attn_variant(request):
    from scail_amd import lib as L
    if request.param == ATTN_VARIANTS[0]:               # the product's 8-wave kernel
        yield request.param
        return
    if not L.ABLATIONS:
        pytest.skip("kernel variant of the measurement build (run with SCAIL_ABLATIONS=1)")
    L.tune_set("attn_variant", request.param)
    yield request.param
    L.tune_set("attn_variant", ATTN_VARIANTS[0])


def _attn_ref(q, k, v, heads):
    return O._merge(O.sdpa(O._heads(q, heads), O._heads(k, heads), O._heads(v, heads)))


@pytest.mark.parametrize("Lq,Lk", [(64, 64), (300, 300), (257, 512), (1000, 257), (96, 96), (40, 1)])
def test_flash_attn(ops, attn_variant, Lq, Lk):
    B, H = 2, 2
    D = H * 128
    q, k, v = rnd(B, Lq, D, seed=1), rnd(B, Lk, D, seed=2), rnd(B, Lk, D, seed=3)
    ref = _attn_ref(q, k, v, H)
    vt = ops.transpose_v(gpu_bf16(v), H)
    o = ops.flash_attn(gpu_bf16(q), gpu_bf16(k), vt)
    close(o, ref, rtol=2e-2, atol=1e-2, msg=f"flash_attn Lq={Lq} Lk={Lk}")


def test_flash_attn_strided_qkv_and_accumulate(ops, attn_variant):
    B, H, Lt = 2, 2, 200
    D = H * 128
    qkv = rnd(B, Lt, 3 * D, seed=1)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    ref = _attn_ref(q, k, v, H)
    g = gpu_bf16(qkv)
    vt = ops.transpose_v(g[..., 2 * D:], H)
    o = ops.flash_attn(g[..., :D], g[..., D:2 * D], vt)
    close(o, ref, atol=1e-2, msg="strided views into the qkv buffer")
    # second key set accumulated on top, batch-broadcast (text + CLIP cross attention)
    k2, v2 = rnd(1, 70, D, seed=5), rnd(1, 70, D, seed=6)
    ref2 = ref + _attn_ref(q, k2.expand(B, -1, -1), v2.expand(B, -1, -1), H)
    ops.flash_attn(g[..., :D], gpu_bf16(k2), ops.transpose_v(gpu_bf16(v2), H), out=o, accumulate=True)
    close(o, ref2, atol=2e-2, msg="accumulate + broadcast K/V")


def test_flash_attn_segments(ops, attn_variant):
    """n_seg > 1: keys of 3 equally sized segments (sequence-parallel all-gather layout)."""
    B, H, Lq, Ls, S = 1, 2, 130, 100, 3
    D = H * 128
    q = rnd(B, Lq, D, seed=1)
    ks, vs = rnd(S, B, Ls, D, seed=2), rnd(S, B, Ls, D, seed=3)
    ref = _attn_ref(q, torch.cat(list(ks), 1), torch.cat(list(vs), 1), H)
    kg = gpu_bf16(ks)
    vtg = torch.stack([ops.transpose_v(gpu_bf16(vs[s]), H) for s in range(S)])
    o = ops.flash_attn(gpu_bf16(q), kg[0], vtg[0], n_seg=S, k_seg_stride=kg.stride(0), vt_seg_stride=vtg.stride(0))
    close(o, ref, atol=1e-2, msg="segmented keys")


def test_flash_attn_rescale_branch(ops, attn_variant):
    """A key that dominates late in the sequence forces the online-softmax running max to jump
    (guide rule 26): the result must still match the fp32 oracle."""
    B, H, Lq, Lk = 1, 1, 64, 320
    q, k, v = rnd(B, Lq, 128, seed=1), rnd(B, Lk, 128, seed=2), rnd(B, Lk, 128, seed=3)
    k[0, 250] = bfr(q[0, 7] * 4.0)
    k[0, 100] = bfr(q[0, 9] * 2.0)
    ref = _attn_ref(q, k, v, H)
    o = ops.flash_attn(gpu_bf16(q), gpu_bf16(k), ops.transpose_v(gpu_bf16(v), H))
    close(o, ref, atol=1e-2, msg="spiked keys")


def test_flash_attn_properties_long(ops):
    """Size-independent properties at a long sequence (one head of the config-2 length):
    rows of softmax sum to one (V = 1 -> O = 1), linearity in V, key-permutation invariance."""
    B, H, Lt = 1, 1, 48832
    g = torch.Generator(device=DEV).manual_seed(0)
    q = torch.randn(B, 2048, 128, device=DEV, generator=g).to(torch.bfloat16)
    k = torch.randn(B, Lt, 128, device=DEV, generator=g).to(torch.bfloat16)
    v1 = torch.randn(B, Lt, 128, device=DEV, generator=g).to(torch.bfloat16)
    ones = torch.ones(B, Lt, 128, device=DEV, dtype=torch.bfloat16)
    o1 = ops.flash_attn(q, k, ops.transpose_v(ones, H)).float()
    assert (o1 - 1).abs().max() < 1e-2
    oa = ops.flash_attn(q, k, ops.transpose_v(v1, H)).float()
    ob = ops.flash_attn(q, k, ops.transpose_v((2 * v1.float()).to(torch.bfloat16), H)).float()
    assert (ob - 2 * oa).abs().max() < 2e-2 * max(1.0, float(oa.abs().max()))
    perm = torch.randperm(Lt, device=DEV, generator=g)
    oc = ops.flash_attn(q, k[:, perm].contiguous(), ops.transpose_v(v1[:, perm].contiguous(), H)).float()
    assert (oc - oa).abs().max() < 1e-2


# ------------------------------------------------------------------------------------------------
# attn4: the hand-scheduled 4-wave kernel (csrc/attn4.s) that scail_flash_attn_bf16 selects for Lk % 64 == 0, Lk >= 512
def _which(q, k, o, accumulate=False, prescaled=False):
    from scail_amd import lib as L
    return L.load().scail_flash_attn_kernel_for(q.stride(1), k.stride(1), o.stride(1), q.shape[1], k.shape[1], 1 if accumulate else 0,
                                                1 if prescaled else 0)


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 1, 256, 512), (2, 2, 300, 576), (1, 2, 700, 1024), (1, 3, 130, 832), (1, 2, 520, 1088)])
def test_attn4_vs_oracle(ops, B, H, Lq, Lk):
    """every remainder path of the unrolled tile loop (8, 9, 16, 13, 17 tiles), ragged query blocks, several heads / batch
    elements (XCD-aware and plain workgroup-id decode), against the fp32 oracle; the 8-wave kernel on the same inputs."""
    from scail_amd import lib as L
    D = H * 128
    q, k, v = rnd(B, Lq, D, seed=1), rnd(B, Lk, D, seed=2), rnd(B, Lk, D, seed=3)
    ref = _attn_ref(q, k, v, H)
    qg, kg = gpu_bf16(q), gpu_bf16(k)
    vt = ops.transpose_v(gpu_bf16(v), H)
    o = torch.empty(B, Lq, D, device=DEV, dtype=torch.bfloat16)
    assert _which(qg, kg, o) == 4
    ops.flash_attn(qg, kg, vt, out=o)
    close(o, ref, rtol=2e-2, atol=1e-2, msg=f"attn4 Lq={Lq} Lk={Lk}")
    L.set_option("attn4", 0)
    try:
        assert _which(qg, kg, o) == 8
        o8 = ops.flash_attn(qg, kg, vt)
    finally:
        L.set_option("attn4", 1)
    close(o8, ref, rtol=2e-2, atol=1e-2, msg="8-wave kernel on the same inputs")
    close(o, o8.float(), rtol=2e-2, atol=1e-2, msg="attn4 vs 8-wave")


def test_attn4_strided_views_segments_and_lazy_rescale(ops):
    from scail_amd import lib as L
    B, H, Lt = 2, 2, 640
    D = H * 128
    qkv = rnd(B, Lt, 3 * D, seed=1)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    k = k.clone()
    k[0, 600, :128] = bfr(q[0, 7, :128] * 3.0)          # late dominant keys: the lazy running max must jump (rescale subroutine)
    k[1, 90, 128:] = bfr(q[1, 300, 128:] * 2.0)
    qkv = torch.cat([q, k, v], -1)
    ref = _attn_ref(q, k, v, H)
    g = gpu_bf16(qkv)
    vt = ops.transpose_v(g[..., 2 * D:], H)
    o = torch.empty(B, Lt, D, device=DEV, dtype=torch.bfloat16)
    assert _which(g[..., :D], g[..., D:2 * D], o) == 4
    for thr in (8, 0, 2):                               # thr 0: rescale whenever any row maximum moves
        L.set_option("attn4_thr", thr)
        ops.flash_attn(g[..., :D], g[..., D:2 * D], vt, out=o)
        close(o, ref, atol=1e-2, msg=f"attn4 on views of the qkv buffer, thr {thr}")
    L.set_option("attn4_thr", 8)
    # key segments (sequence-parallel all-gather layout): 3 x 512 keys
    S, Ls, Lq = 3, 512, 200
    q = rnd(1, Lq, D, seed=4)
    ks, vs = rnd(S, 1, Ls, D, seed=5), rnd(S, 1, Ls, D, seed=6)
    ref = _attn_ref(q, torch.cat(list(ks), 1), torch.cat(list(vs), 1), H)
    kg = gpu_bf16(ks)
    vtg = torch.stack([ops.transpose_v(gpu_bf16(vs[s]), H) for s in range(S)])
    o = ops.flash_attn(gpu_bf16(q), kg[0], vtg[0], n_seg=S, k_seg_stride=kg.stride(0), vt_seg_stride=vtg.stride(0))
    close(o, ref, atol=1e-2, msg="attn4 segmented keys")
    # shapes outside its limits go to the 8-wave kernels
    assert _which(g[..., :D], g[:, :500, D:2 * D], o) == 8 and _which(g[..., :D], g[..., D:2 * D], o, accumulate=True) == 8
    assert _which(g[..., :D], g[:, :448, D:2 * D], o) == 8


# ------------------------------------------------------------------------------------------------
def test_small_ops(ops):
    from scail_amd import lib as L
    t = torch.tensor([0.0, 1000.0, 731.0, 995.9])
    e = ops.timestep_embedding(t.to(DEV), 256)
    close(e, O.timestep_embedding(t, 256), rtol=0, atol=2e-6, msg="timestep_embedding")
    x, w, b = torch.randn(2, 256), rnd(384, 256, seed=1, scale=0.05), rnd(384, seed=2)
    y = ops.small_linear(x.to(DEV), gpu_bf16(w), b.to(DEV), act_in=L.ACT_SILU, act_out=L.ACT_SILU)
    ref = torch.nn.functional.silu(torch.nn.functional.silu(x) @ w.t() + b)
    close(y, ref, rtol=1e-4, atol=1e-4, msg="small_linear")
    emb, tab = torch.randn(2, 96), torch.randn(5, 96)
    close(ops.adaln_table(emb.to(DEV), tab.to(DEV)), emb[None] + tab[:, None], rtol=0, atol=0, msg="adaln_table")
    xx, v = torch.randn(1, 3, 16, 8, 8), torch.randn(2, 3, 16, 8, 8)
    xg = xx.clone().to(DEV)
    ops.cfg_euler_(xg, v.to(DEV), 4.0, -0.25)
    close(xg, xx + (-0.25) * O.cfg_combine(v[:1], v[1:], 4.0), rtol=1e-6, atol=1e-6, msg="cfg_euler")
    z = torch.randn(1000)
    close(ops.to_bf16(z.to(DEV)), z.to(torch.bfloat16), rtol=0, atol=0, msg="to_bf16 RNE")


def test_patchify_embed_and_unpatchify(ops):
    cfg = O.DiTConfig(**O.TINY)
    sd = O.make_state_dict(cfg)
    B, T, H, W = 2, 3, 8, 12
    x, ref, pose = rnd(B, T, 16, H, W, seed=1), rnd(1, 1, 16, H, W, seed=2), rnd(1, T, 16, H // 2, W // 2, seed=3)
    x20 = torch.cat([x, torch.zeros(B, T, 4, H, W)], 2)
    r20 = torch.cat([ref.expand(B, -1, -1, -1, -1), torch.ones(B, 1, 4, H, W)], 2)
    p20 = torch.cat([pose.expand(B, -1, -1, -1, -1), torch.ones(B, T, 4, H // 2, W // 2)], 2)
    want = O.patch_embed(cfg, sd, x20, r20, p20)
    tok = ops.patchify(x.to(DEV), gpu_bf16(ref), gpu_bf16(pose))
    D = cfg.hidden_size
    Lrn = (1 + T) * (H // 2) * (W // 2)
    h = torch.empty(B, tok.shape[1], D, device=DEV, dtype=torch.bfloat16)
    for name, sl in (("proj", slice(0, Lrn)), ("proj_pose", slice(Lrn, None))):
        wp = torch.zeros(D, 128)
        wp[:, :80] = sd[f"mixins.patch_embed.{name}.weight"].reshape(D, 80)
        for b in range(B):
            ops.gemm(tok[b, sl], gpu_bf16(wp), sd[f"mixins.patch_embed.{name}.bias"].to(DEV), out=h[b, sl])
    close(h, want, msg="patchify + patch-embed GEMM")
    # unpatchify against the oracle's rearrangement (final_layer with identity weights is overkill:
    # test the index map directly)
    tokout = rnd(B, T * (H // 2) * (W // 2), 64, seed=9)
    wantu = tokout.reshape(B, T, H // 2, W // 2, 1, 2, 2, 16).permute(0, 1, 4, 7, 2, 5, 3, 6).reshape(B, T, 16, H, W)
    close(ops.unpatchify(gpu_bf16(tokout), T, H, W), wantu, rtol=0, atol=0, msg="unpatchify")


# ------------------------------------------------------------------------------------------------
# fused two-key-set cross attention (text + CLIP, dit_video_crossattn_sc_xc.py:1107-1203)
@pytest.mark.parametrize("cross4", [1, 0, 2])
@pytest.mark.parametrize("B,H,Lq,Lk1,Lk2,shared2", [(2, 2, 300, 512, 257, True), (1, 3, 128, 77, 1, False), (2, 1, 515, 64, 320, False),
                                                     (1, 2, 40, 130, 257, True), (2, 2, 1000, 512, 257, False), (2, 5, 1500, 512, 257, True),
                                                     (1, 1, 256 * 300 + 17, 64, 65, True), (1, 2, 700, 1100, 300, True)])
def test_cross_attn2_vs_oracle(ops, B, H, Lq, Lk1, Lk2, shared2, cross4):
    from scail_amd import lib as L_
    L_.set_option("cross4", cross4)       # 1: scail_attn4_x2 (generated, persistent workgroups) where eligible; 0: cross_attn2_kernel for every shape
    try:
        which = L_.load().scail_cross_attn2_kernel_for(3 * H * 128, H * 128, H * 128, H * 128, Lq, Lk1, Lk2, B, H)
        tiles = (Lk1 + 63) // 64 + (Lk2 + 63) // 64
        assert which == (4 if (Lk1 >= 64 and Lk2 >= 64 and (cross4 == 1 or (cross4 == 2 and tiles >= 21))) else 2)
        _cross_attn2_case(ops, B, H, Lq, Lk1, Lk2, shared2, generated=which == 4)
    finally:
        L_.set_option("cross4", 2)


@pytest.mark.parametrize("which_set", [0, 1])
def test_cross_attn2_generated_restart_in_every_wave(ops, which_set):
    """scail_attn4_x2 carries the same four-flag restart decision as scail_attn4_m16f, once per key set (see
    test_restart_decision_reaches_every_wave): an overflowing query row in each wave, in either set, must give fp32 softmax on all rows."""
    from scail_amd import lib as L_
    H, Lq, Lk1, Lk2 = 1, 512, 64 * 8, 64 * 5 + 1
    g = torch.Generator(device=DEV).manual_seed(11)
    q = torch.randn(1, Lq, 128, device=DEV, generator=g)
    ks = [torch.randn(1, n, 128, device=DEV, generator=g).to(torch.bfloat16) for n in (Lk1, Lk2)]
    vs = [torch.randn(1, n, 128, device=DEV, generator=g).to(torch.bfloat16) for n in (Lk1, Lk2)]
    vts = [ops.transpose_v(v, H) for v in vs]
    qb = (q * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
    L_.set_option("cross4", 1)
    try:
        assert L_.load().scail_cross_attn2_kernel_for(128, 128, 128, 128, Lq, Lk1, Lk2, 1, H) == 4
        for wave in range(4):
            for r in (256 + wave * 64, 256 + wave * 64 + 37, wave * 64 + 63):
                kk = [k.clone() for k in ks]
                kk[which_set][0, 64 * 2 + 9] = (q[0, r] * 14.0).to(torch.bfloat16)
                ref = torch.zeros(Lq, 128, device=DEV)
                for i, (k, v) in enumerate(zip(kk, vs)):
                    part = torch.softmax(qb[0].float() @ k[0].float().t() * math.log(2.0), dim=-1) @ v[0].float()
                    ref = part.to(torch.bfloat16).float() if i == 0 else ref + part      # set 0 is stored as bf16 and added to set 1
                for rep in range(3):
                    o = ops.cross_attn2(qb, kk[0], vts[0], kk[1], vts[1], q_prescaled=True)
                    torch.cuda.synchronize()
                    assert torch.isfinite(o.float()).all(), (wave, r, rep)
                    torch.testing.assert_close(o[0].float(), ref, rtol=2e-2, atol=2e-2, msg=lambda m: f"set {which_set} wave {wave} row {r} rep {rep}: {m}")
    finally:
        L_.set_option("cross4", 2)


def _cross_attn2_case(ops, B, H, Lq, Lk1, Lk2, shared2, generated=False):
    """one launch over two key sets = the oracle's two attentions, each rounded to bf16, added (the reference adds the bf16
    outputs of two attention_fn calls); ragged last tiles in both sets, one-key set, query rows not a multiple of 128, the
    second set shared by the batch (CLIP of an unbatched reference image) or per batch element, strided q view."""
    D = H * 128
    B2 = 1 if shared2 else B
    qkv = rnd(B, Lq, 3 * D, seed=1)
    q = qkv[..., :D]
    k1, v1 = rnd(B, Lk1, D, seed=2), rnd(B, Lk1, D, seed=3)
    k2, v2 = rnd(B2, Lk2, D, seed=4), rnd(B2, Lk2, D, seed=5, scale=2.0)
    k2[0, Lk2 - 1, :128] = bfr(q[0, 3, :128] * 2.0)            # a dominant key in the masked last tile of set 2
    ref = bfr(_attn_ref(q, k1, v1, H)) + _attn_ref(q, k2.expand(B, -1, -1), v2.expand(B, -1, -1), H)
    g = gpu_bf16(qkv)
    vt1 = ops.transpose_v(gpu_bf16(v1), H)
    vt2 = ops.transpose_v(gpu_bf16(v2), H)
    o = torch.full((B, Lq + 1, D), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.cross_attn2(g[..., :D], gpu_bf16(k1), vt1, gpu_bf16(k2), vt2, out=o[:, :Lq])
    # (the generated kernel works on queries in log2 units, rounded once more to bf16 in its prologue for a raw-scale caller: the contract's
    # rtol / atol 2e-2; the hipcc kernel holds the tighter bound)
    close(o[:, :Lq], ref, rtol=2e-2, atol=2e-2 if generated else 1e-2, msg=f"cross_attn2 {B, H, Lq, Lk1, Lk2}")
    assert (o[:, Lq] == 7.0).all(), "rows past Lq must not be written"
    # and against the two-launch path it replaces (same rounding points: bit-close, not just tolerance-close)
    o2 = ops.flash_attn(g[..., :D], gpu_bf16(k1), vt1)
    ops.flash_attn(g[..., :D], gpu_bf16(k2), vt2, out=o2, accumulate=True)
    close(o[:, :Lq], o2.float(), rtol=2e-2 if generated else 1e-2, atol=2e-2 if generated else 4e-3, msg="cross_attn2 vs flash_attn + accumulate")
    # queries handed over in log2 units (what the DiT executor does: scail_rmsnorm_rope_scaled + SCAIL_ATTN_Q_PRESCALED)
    qs = gpu_bf16(q * ops.ATTN_LOG2_SCALE)
    o3 = ops.cross_attn2(qs, gpu_bf16(k1), vt1, gpu_bf16(k2), vt2, q_prescaled=True)
    close(o3, ref, rtol=2e-2, atol=1.5e-2, msg="cross_attn2, prescaled q")


@pytest.mark.parametrize("B,H,Lq,Lk,which", [(1, 1, 256, 512, 4), (2, 2, 300, 576, 4), (1, 2, 700, 1024, 4), (1, 3, 130, 832, 4), (1, 2, 520, 1088, 4),
                                             (1, 2, 300, 513, 4), (2, 1, 130, 64 * 9 + 63, 4), (1, 1, 700, 64 * 13 + 17, 4), (1, 2, 64, 1000, 4),   # ragged key counts
                                             (2, 1, 200, 300, 8), (1, 2, 64, 448, 8)])
def test_flash_attn_prescaled_queries(ops, B, H, Lq, Lk, which):
    """scale == SCAIL_ATTN_Q_PRESCALED: q arrives multiplied by scale * log2(e) (one rounding, as scail_rmsnorm_rope_scaled leaves
    it).  Shapes the 4-wave kernel takes run scail_attn4_m16f (16x16x32 MFMAs, maximum folded into the accumulator init: every
    remainder path of its tile loop, ragged query blocks, XCD-aware and plain id decode); the others the 8-wave kernel with a unit
    scale.  Reference: fp32 attention of the queries the kernel effectively sees (q' / (scale log2 e))."""
    from scail_amd import lib as L
    D = H * 128
    c = ops.ATTN_LOG2_SCALE
    q, k, v = rnd(B, Lq, D, seed=1), rnd(B, Lk, D, seed=2), rnd(B, Lk, D, seed=3)
    k[0, Lk - 3, :128] = bfr(q[0, 7, :128] * 3.0)                   # late dominant key: the running maximum must jump
    qp = bfr(q * c)
    ref = _attn_ref(qp / c, k, v, H)
    qg, kg = gpu_bf16(qp), gpu_bf16(k)
    vt = ops.transpose_v(gpu_bf16(v), H)
    o = torch.empty(B, Lq, D, device=DEV, dtype=torch.bfloat16)
    assert _which(qg, kg, o, prescaled=True) == which
    assert _which(qg, kg, o) == which                            # a raw scale no longer changes the kernel (round 3: Q scaled in the prologue)
    for thr in ((8, 0) if which == 4 else (8,)):
        L.set_option("attn4_thr", thr)
        try:
            ops.flash_attn(qg, kg, vt, out=o, q_prescaled=True)
        finally:
            L.set_option("attn4_thr", 8)
        close(o, ref, rtol=2e-2, atol=1e-2, msg=f"prescaled q, kernel {which}, thr {thr}")


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 2, 300, 513), (2, 1, 130, 1000), (1, 1, 700, 64 * 13 + 17), (2, 2, 260, 64 * 11)])
def test_flash_attn_raw_scale_any_key_count_runs_the_4wave_kernel(ops, B, H, Lq, Lk):
    """seam B3 (sat/transformer_defaults.py:47-79: q arrives unscaled, any key count): scail_attn4_m16f multiplies its Q fragments
    by scale * log2(e) in the prologue; round 2 sent raw-scale callers to the 32x32 kernel and ragged key counts to the 8-wave one"""
    D = H * 128
    q, k, v = rnd(B, Lq, D, seed=1), rnd(B, Lk, D, seed=2), rnd(B, Lk, D, seed=3)
    k[0, Lk - 3, :128] = bfr(q[0, 7, :128] * 3.0)
    qg, kg = gpu_bf16(q), gpu_bf16(k)
    vt = ops.transpose_v(gpu_bf16(v), H)
    o = torch.empty(B, Lq, D, device=DEV, dtype=torch.bfloat16)
    assert _which(qg, kg, o) == 4
    ops.flash_attn(qg, kg, vt, out=o)
    close(o, _attn_ref(q, k, v, H), rtol=2e-2, atol=1e-2, msg=f"raw scale, Lk={Lk}")
    ops.flash_attn(qg, kg, vt, out=o, scale=0.05)                       # another scale than 1 / sqrt(128)
    ref = O._merge(O.sdpa(O._heads(q, H) * (0.05 * math.sqrt(128.0)), O._heads(k, H), O._heads(v, H)))
    close(o, ref, rtol=2e-2, atol=1e-2, msg="scale 0.05")
    from scail_amd import lib as L
    for bad in (0.0, -0.5, float("nan"), float("inf")):
        with pytest.raises(L.ScailHipError, match="positive finite"):
            ops.flash_attn(qg, kg, vt, out=o, scale=bad)


@pytest.mark.parametrize("factor,prescaled", [(5.0, True), (12.0, True), (12.0, False), (30.0, True)])
def test_flash_attn_optimistic_loop_overflow_restart(ops, factor, prescaled):
    """scail_attn4_m16f's hot loop fixes the reference point of exp2 after the first tile (+ 40 log2 units of headroom) and tracks no
    maximum; a key whose score exceeds that by more than ~100 (factor 12: ~196 log2 units above the first tile's maximum, factor
    30: ~490) overflows, the row-sum check in the epilogue sees it and the workgroup runs again with the lazy-maximum loop.
    factor 5 (~80 units) stays inside the headroom.  Spikes in several query blocks / heads / batch elements, in the hot loop
    (tiles 1..) and in the remainder tiles, against the fp32 oracle."""
    B, H, Lq, Lk = 2, 2, 700, 64 * 23 + 9
    D = H * 128
    c = ops.ATTN_LOG2_SCALE
    q, k, v = rnd(B, Lq, D, seed=1), rnd(B, Lk, D, seed=2), rnd(B, Lk, D, seed=3)
    for b, row, key, h in ((0, 7, 64 * 2 + 5, 0), (0, 300, 64 * 9, 1), (1, 699, Lk - 2, 0), (1, 513, 64 * 21 + 3, 1), (1, 100, 3, 1)):
        k[b, key, h * 128:(h + 1) * 128] = bfr(q[b, row, h * 128:(h + 1) * 128] * factor)
    if prescaled:
        qp = bfr(q * c)
        ref = _attn_ref(qp / c, k, v, H)
        o = ops.flash_attn(gpu_bf16(qp), gpu_bf16(k), ops.transpose_v(gpu_bf16(v), H), q_prescaled=True)
    else:       # raw scale: the prologue rounds q * scale * log2(e) to bf16 -- reference = the queries the loop sees, as in the prescaled case
        ref = _attn_ref(bfr(q * c) / c, k, v, H)
        o = ops.flash_attn(gpu_bf16(q), gpu_bf16(k), ops.transpose_v(gpu_bf16(v), H))
    assert torch.isfinite(o.float()).all()
    close(o, ref, rtol=2e-2, atol=1e-2, msg=f"spike factor {factor}")


def test_flash_attn_prescaled_extreme_first_tile(ops):
    """scail_attn4_m16f starts its running maximum at 0 and lets the first tile set it: rows whose scores all lie ~300 log2 units below
    (above) zero must neither underflow to 0 / 0 nor overflow"""
    H, Lq, Lk = 1, 256, 512
    c = ops.ATTN_LOG2_SCALE
    k = bfr(rnd(1, Lk, 128, seed=2) * 0.05 + 1.0)
    v = rnd(1, Lk, 128, seed=3)
    vt = ops.transpose_v(gpu_bf16(v), H)
    for sign in (-1.0, 1.0):
        qp = bfr((sign * 18.0 + rnd(1, Lq, 128, seed=1) * 0.05) * c)
        o = ops.flash_attn(gpu_bf16(qp), gpu_bf16(k), vt, q_prescaled=True)
        assert torch.isfinite(o.float()).all()
        close(o, _attn_ref(qp / c, k, v, H), rtol=2e-2, atol=1e-2, msg=f"sign {sign}")
