#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): denoising-step latent tokens/sec, SCAIL-14B, 512p x 81 f.

One "step" = one full sampler step of the reference (sampling.py:960-963): a batch-2 (uncond, cond)
DiT forward over the L = 48 832 token sequence [ref | noise | pose] + CFG combine + Euler update on
the fp32 state.  value = 37 632 noise tokens / t_step, whole job.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, or -- when WORLD_SIZE is not
                                                        set -- this script re-executes itself under it with N local ranks)

Inputs are synthetic and already resident in HBM when the timed region starts; weights
are random-init bf16 of the 14B architecture (no checkpoint offline).  Nothing is skipped inside the
timed region: the step-invariant text/CLIP K,V cache of the engine is DISABLED here, so every step
recomputes text_embedding / clip_proj / 40x K,V projections exactly like the reference does.

N > 1 shards the token axis (sequence parallel, K/V all-gather over xGMI) -> "scaling": "strong".
At N = 1 the timed step is the PRODUCT DEFAULT path: one call of the C executor (scail_dit_step, include/scail_dit.h) per network
evaluation; the per-kernel times come from the executor's own HIP-event pairs on the launch stream (scail_dit_profile).
The JSON line also carries
  roofline     MFMA roofline of the dominant kernel (self-attention flash kernel): algorithmic
               4*Lq*Lk*128*heads*B FLOP per launch / mean launch time measured with HIP events on the
               launch stream inside the timed region; peak 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md);
  roofline_gemm  the same for the six per-token GEMMs of a block taken together (36 % of the step's FLOPs);
  cpu_baseline the CPU oracle (fp32 port of the reference block) timed on this box's host cores after a warm-up: the per-token
               part at three sequence lengths, the self-attention (the L^2 term) separately on 4 of the 80 (batch, head)
               pairs up to L = 36 624 (0.75 x the bench length); fitted (with a standard error) and
               evaluated at the bench length (rank 0, N = 1; "kind": "port, extrapolated");
  config.vae   BASELINE config 4 (Wan2.1 VAE encode + decode at 81 x 512 x 896) run once after the timed region,
               with both roofline fractions (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# the host driver supports only dmabuf IPC: RCCL between the ranks of one node fails without this (set before HIP starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# RCCL channel count: deliberately NOT capped (no NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS here).  Round 6 measured on one GPU what a
# collective's CU occupancy costs a rank of 8 (a K-workgroup resident kernel on a side stream moving the real exchange volume, 48 GB/s
# per link and direction; profiles/r06_sp_comm_standin.log): K = 8 / 16 / 32 / 64 channels -> 71 / 81 / 88 / 90 % of T1 / 8 against 94.6 %
# with free communication -- the exchange's DURATION (fewer channels = a longer exposed way back) costs more than the CUs the channels
# hold, so a cap would hurt; telling the attention launch plan about the held CUs (option "attn4_cus") moved the four points by
# +2.1 / +3.6 / -3.4 / +0.5 and stays off.  An operator's NCCL_* settings are reported in the N > 1 line (config.rccl_env).

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (network params, latent T,H,W, text tokens, clip tokens)
    "14b": (dict(hidden_size=5120, num_layers=40, num_attention_heads=40, inner_hidden_size=13824, text_dim=4096,
                 time_freq_dim=256, time_embed_dim=5120), (21, 64, 112), 512, 257),
    "1.3b": (dict(hidden_size=1536, num_layers=30, num_attention_heads=12, inner_hidden_size=8960, text_dim=4096,
                  time_freq_dim=256, time_embed_dim=1536), (21, 64, 112), 512, 257),
    # BASELINE config 5: 2 reference frames + 2 pose streams (an EXTENSION: the reference has exactly one of each, dit...:1559;
    # parity is against the oracle extended the same way = unpinned by construction); L = 60 032 tokens
    "14b-2char": (dict(hidden_size=5120, num_layers=40, num_attention_heads=40, inner_hidden_size=13824, text_dim=4096,
                       time_freq_dim=256, time_embed_dim=5120), (21, 64, 112), 512, 257),
    "tiny": (dict(hidden_size=256, num_layers=2, num_attention_heads=2, inner_hidden_size=512, text_dim=64,
                  time_freq_dim=256, time_embed_dim=256), (4, 8, 8), 12, 5),
}
PEAK_BF16_TFLOPS = 2500.0


def step_flops(p, L, Lt, Lc, B=2):
    """Algorithmic FLOPs of one sampler step."""
    D, FF, nl = p["hidden_size"], p["inner_hidden_size"], p["num_layers"]
    per = 8 * L * D * D + 4 * L * D * D + 4 * (Lt + Lc) * D * D + 4 * L * D * FF + 4 * L * L * D + 4 * L * (Lt + Lc) * D
    return per * nl * B


class KernelTimer:
    """HIP-event bracket around tagged launches on the current (launch) stream."""

    def __init__(self):
        self.events = {}
        self.enabled = False

    def run(self, tag, fn, *a, **k):
        if not self.enabled:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        self.events.setdefault(tag, []).append((e0, e1))
        return r

    def mean_ms(self, tag):
        ev = self.events.get(tag, [])
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None


def _nnls(cols, ts):
    """Non-negative least squares of ts = sum_i coef_i * cols[i]: every subset of active terms is solved by ordinary least
    squares and the feasible (all coefficients >= 0) solution with the smallest residual wins.  Returns (coef, rms residual)."""
    import itertools
    import numpy as np
    ts = np.asarray(ts, dtype=np.float64)
    cols = [np.asarray(c, dtype=np.float64) for c in cols]
    best = None
    for k in range(1, min(len(cols), len(ts)) + 1):
        for act in itertools.combinations(range(len(cols)), k):
            A = np.stack([cols[i] for i in act], 1)
            sol, *_ = np.linalg.lstsq(A, ts, rcond=None)
            if (sol < 0).any():
                continue
            res = float(((A @ sol - ts) ** 2).sum())
            if best is None or res < best[0] - 1e-12 or (abs(res - best[0]) <= 1e-12 and k > best[2]):
                coef = [0.0] * len(cols)
                for i, v in zip(act, sol):
                    coef[i] = float(v)
                best = (res, coef, k)
    return best[1], (best[0] / len(ts)) ** 0.5


def _physical_cores():
    """Physical cores of the host (the figure north_star asks for), not hardware threads."""
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(p, L_bench):
    """Time the CPU oracle (fp32 restatement of the reference block, oracle/scail_oracle.py; reference
    dit_video_crossattn_sc_xc.py:1009-1051, sat/transformer_defaults.py:47-79) on a BOUNDED sample at the real width
    (D = 5120, 40 heads, B = 2) and extrapolate to one block at the bench length:
      1. thread count: the block's largest projection (2016 x 5120 x 15360) is timed at 32 / 64 / 128 / all hardware threads
         and the fastest is used (all SMT threads of the GPU box's host are NOT the fastest for torch CPU);
      2. per-token part (projections, MLP, norms, RoPE, the short-key cross attention): the whole block at L = 1008 / 2128 /
         3248 with the time of its self-attention call taken out -> fit  c + a L  (exactly linear in L by construction);
      3. self-attention (62 % of the step's FLOPs at the bench length, the L^2 term): O.sdpa alone on 4 of the 80 (batch,
         head) pairs (two calls of the block's own head-view call shape) at L = 6104 / 12 208 / 24 416 / 36 624 (as many as fit ~30 s),
         scaled to 80 pairs (the pairs are independent and each saturates the threads) -> fit  a2 L + b L^2; the bench length is
         1.33x the longest timed one (round 3: 2x, round 2: 9x); the fits' standard errors at the bench length go into the line;
    Returns a dict with the fits, their rms residuals, the samples and t_block(L_bench)."""
    from oracle import scail_oracle as O
    import torch.nn.functional as F
    hw = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    D, nh = p["hidden_size"], p["num_attention_heads"]
    xw, ww = torch.randn(2016, D, generator=g), torch.randn(3 * D, D, generator=g) * 0.02
    probe = {}
    for nt in sorted({min(hw, 32), min(hw, 64), min(hw, 128), hw}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            F.linear(xw, ww)
            t0 = time.perf_counter()
            F.linear(xw, ww)
            probe[nt] = time.perf_counter() - t0
    n_threads = min(probe, key=probe.get)
    torch.set_num_threads(n_threads)
    del xw, ww
    cfg = O.DiTConfig(hidden_size=D, num_layers=1, num_attention_heads=nh,
                      inner_hidden_size=p["inner_hidden_size"], text_dim=64, time_embed_dim=D)
    sd = {k: torch.randn(s, generator=g) * 0.02 for k, s in O.state_dict_spec(cfg).items()
          if ".layers.0." in k or "adaln_layer" in k}
    Lt, Lc = 512, 257
    adaln, text, clip = torch.randn(2, 6 * D, generator=g), torch.randn(2, Lt, D, generator=g), torch.randn(2, Lc, D, generator=g)
    self_t = [0.0]
    plain = O.sdpa

    def timed_sdpa(q, k, v):
        if q.shape[2] != k.shape[2]:                      # text / CLIP cross attention: part of the per-token term
            return plain(q, k, v)
        t0 = time.perf_counter()
        r = plain(q, k, v)
        self_t[0] += time.perf_counter() - t0
        return r

    def run_block(T):
        cos, sin = O.rope_tables(cfg, T, 16, 28)
        h = torch.randn(2, cos.shape[0], D, generator=g)
        self_t[0] = 0.0
        O.sdpa = timed_sdpa
        try:
            with torch.no_grad():
                t0 = time.perf_counter()
                O.block(cfg, sd, 0, h, adaln, text, clip, cos, sin)
                dt = time.perf_counter() - t0
        finally:
            O.sdpa = plain
        return cos.shape[0], dt, self_t[0]

    run_block(1)                              # warm-up: thread pool, allocator, BLAS kernel selection
    blocks = [run_block(T) for T in (1, 3, 5)]            # L = 1008, 2128, 3248
    (c0, a1), res_tok = _nnls([[1.0] * len(blocks), [q[0] for q in blocks]], [q[1] - q[2] for q in blocks])
    pairs_all, pairs, call_pairs = 2 * nh, 4, 2
    att = []                                  # (L, pairs, seconds for these pairs)
    spent = 0.0
    # lengths up to 0.75 x the bench length: the L^2 term is extrapolated 1.33x (round 3: 2x, round 2: 9x); 4 of the 80 (batch, head) pairs per
    # length, as two calls of the block's own call shape
    for La in ((6104, 12208, 24416, 36624) if D >= 1024 else (1024, 2048)):      # toy widths (tests): two short samples
        # bounded sample: stop before a length whose predicted cost ((L / L_prev)^2 x the previous one) would push the attention samples past ~30 s
        if len(att) >= 2 and spent + (La / att[-1][0]) ** 2 * att[-1][2] > 30.0:
            break
        dt = 0.0
        for _ in range(pairs // call_pairs):
            # the call shape the block uses: (1, pairs, L, 128) head views of token-major (1, L, pairs * 128) tensors
            q, k, v = (O._heads(torch.randn(1, La, call_pairs * 128, generator=g), call_pairs) for _ in range(3))
            with torch.no_grad():
                t0 = time.perf_counter()
                plain(q, k, v)
                dt += time.perf_counter() - t0
            del q, k, v
        att.append((La, pairs, dt))
        spent += dt
    pts = [(La, dt * pairs_all / pr) for La, pr, dt in att]
    (a2, b2), res_att = _nnls([[q[0] for q in pts], [q[0] ** 2 for q in pts]], [q[1] for q in pts])
    # standard error of the two fitted predictions at the bench length (ordinary least-squares formula on the active terms:
    # s^2 x0^T (X^T X)^-1 x0 with s^2 = RSS / max(n - p, 1))
    import numpy as np

    def pred_sigma(cols, coef, rms, n, x0):
        act = [i for i, cf in enumerate(coef) if cf > 0.0] or [0]
        X = np.stack([np.asarray(cols[i], dtype=np.float64) for i in act], 1)
        x = np.asarray([x0[i] for i in act], dtype=np.float64)
        s2 = rms * rms * n / max(n - len(act), 1)
        return float(np.sqrt(max(s2 * x @ np.linalg.pinv(X.T @ X) @ x, 0.0)))

    sig_tok = pred_sigma([[1.0] * len(blocks), [q[0] for q in blocks]], (c0, a1), res_tok, len(blocks), (1.0, L_bench))
    sig_att = pred_sigma([[q[0] for q in pts], [q[0] ** 2 for q in pts]], (a2, b2), res_att, len(pts), (L_bench, L_bench * L_bench))
    t_tok = c0 + a1 * L_bench
    t_att = a2 * L_bench + b2 * L_bench * L_bench
    return dict(threads=n_threads, probe=probe, blocks=blocks, att=att, c=c0, a=a1, a2=a2, b=b2, res_tok=res_tok, res_att=res_att,
                t_tok=t_tok, t_att=t_att, t_block=t_tok + t_att, max_timed_L=max(q[0] for q in pts), pairs_timed=pairs,
                sigma_block=(sig_tok ** 2 + sig_att ** 2) ** 0.5)


def _git_blob_sha1(path):
    """git's blob id of a file (sha1 of 'blob <size>\\0' + content): ties a committed PMC measurement to the kernel source."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


VAE_ALG = {"encode": (188.3e12, 148.7e9), "decode": (316.5e12, 229.4e9)}      # SURVEY.md 8d, 81 x 512 x 896


def vae_leg(dev, pmc=True):
    """BASELINE config 4 (Wan2.1 VAE encode + decode only, 512p x 81 f) after the timed DiT region: one warm-up and one
    timed call per direction, HIP events on the launch stream; algorithmic 188.3 / 316.5 TFLOP and 148.7 / 229.4 GB
    (every conv reads its input once and writes its output once, norm + SiLU fused)."""
    from scail_amd.wan_vae import WanVAE_
    m = WanVAE_(dim=96, z_dim=16, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    video = torch.rand(1, 3, 81, 512, 896, device=dev, generator=g) * 2 - 1
    z = torch.randn(1, 16, 21, 64, 112, device=dev, generator=g)
    res = {"workload": "Wan2.1 VAE encode + decode, 81x512x896, random-init weights, synthetic video", "dtype": "bf16"}
    for name, fn, arg in (("encode", m.encode, video), ("decode", m.decode, z)):
        fn(arg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(arg); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        fl, by = VAE_ALG[name]
        res[name + "_ms"] = ms
        res[name + "_achieved_mfma"] = {"TFLOP/s": fl / ms / 1e9, "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS}
        res[name + "_achieved_hbm"] = {"GB/s": by / ms / 1e6, "frac": by / ms / 1e6 / 8000.0}
        res[name + "_finite"] = bool(torch.isfinite(out).all().item())
        del out
    # fabric traffic of the dominant convolution launches (PMC, measured offline like roofline.traffic; an entry is dropped when the generated source its kernel comes from changed)
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        blobs = {f: _git_blob_sha1(os.path.join(ROOT, "scail_amd", "csrc", f)) for f in ("conv4.s", "conv4u.s")}      # an entry names the file its kernel comes from
        conv = {k: {"traffic_bytes": v["traffic_bytes"], "algorithmic_bytes": v["algorithmic_bytes"], "shape": v["shape"], "kernel": v["kernel"].split(" ")[0]}
                for k, v in tr.items() if k.startswith("conv4_c") and v.get("source_blob") == blobs.get(v.get("source"))}
        res["conv_traffic"] = conv or None
    except Exception:
        res["conv_traffic"] = None
    res["conv_traffic_source"] = "profiles/traffic.json (stamped with the git blob of the kernel source)"
    if pmc:
        # ... and from counters read in THIS run: the three dominant 3x3x3 shapes (tools/conv_pmc_probe.py), FETCH_SIZE / WRITE_SIZE one pass each
        try:
            t0 = time.time()
            live = {}
            for C_, (H_, W_) in ((96, (512, 896)), (192, (256, 448)), (384, (128, 224))):
                v = {c: max(_pmc_pass([os.path.join(ROOT, "tools", "conv_pmc_probe.py"), str(C_), "2"], c, "scail_conv4").items(), key=lambda kv: kv[1]) for c in ("FETCH_SIZE", "WRITE_SIZE")}
                live[f"conv4_c{C_}"] = {"traffic_bytes": _kib_to_bytes(v["FETCH_SIZE"][1], v["WRITE_SIZE"][1]), "algorithmic_bytes": 2.0 * 21 * H_ * W_ * C_ * 2 + 27 * C_ * C_ * 2,
                                        "shape": {"T": 21, "H": H_, "W": W_, "C": C_}, "kernel": v["FETCH_SIZE"][0]}
            res["conv_traffic_stamped"] = res["conv_traffic"]
            res["conv_traffic"] = live
            res["conv_traffic_source"] = f"PMC counters of this run (rocprofv3 --pmc around tools/conv_pmc_probe.py, {time.time() - t0:.0f} s)"
        except Exception as e:          # noqa: BLE001
            res["conv_traffic_in_run_error"] = f"{type(e).__name__}: {e}"[:300]
    return res


_PMC_BROKEN = []


def _pmc_pass(probe_args, counter, like):
    """ONE rocprofv3 --pmc pass (one counter set, kernel trace only -- the guide's recipe) around a child process that launches kernels on the
    product library; returns {kernel name: mean of the counter per dispatch (summed over its XCD / channel instances)} for names containing ``like``."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    if _PMC_BROKEN:                      # one failed / timed-out pass ends the in-run counter collection: the bench must not wait for a profiler
        raise RuntimeError("an earlier rocprofv3 pass of this run failed: " + _PMC_BROKEN[0])
    d = tempfile.mkdtemp(prefix="scail_pmc_", dir="/tmp")
    try:
        try:
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable] + probe_args,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=90)
        except Exception as e:          # noqa: BLE001  (timeout, exec failure)
            _PMC_BROKEN.append(f"{type(e).__name__}: {e}"[:200])
            raise
        dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
        if r.returncode != 0 or not dbs:
            _PMC_BROKEN.append(f"rc {r.returncode}, {len(dbs)} database(s)")
            raise RuntimeError(f"rocprofv3 --pmc {counter}: rc {r.returncode}, {len(dbs)} database(s): " + r.stdout.decode(errors="replace")[-300:])
        c = sqlite3.connect(dbs[0])
        cols = [row[1] for row in c.execute("pragma table_info(counters_collection)")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        cnt_col = "counter_name" if "counter_name" in cols else "counter"
        val_col = "value" if "value" in cols else "counter_value"
        q = (f"select {name_col}, avg(v) from (select {name_col}, dispatch_id, sum({val_col}) as v from counters_collection "
             f"where {name_col} like ? and {cnt_col} = ? group by {name_col}, dispatch_id) group by {name_col}")
        res = {n.split("(")[0].strip(): float(v) for n, v in c.execute(q, (f"%{like}%", counter))}
        c.close()
        if not res:
            raise RuntimeError(f"no dispatch of *{like}* with {counter} in the database")
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _kib_to_bytes(fetch_kib, write_kib):
    return (2.0 * fetch_kib + write_kib) * 1024.0          # gfx950: 128-byte read requests are tallied at 64 (the guide's correction)


def pmc_traffic_leg(expected_attn, expected_gemm):
    """roofline.traffic / roofline_gemm.traffic from counters read IN THIS RUN: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: one pass each)
    around child processes that launch (a) the config-2 self-attention (tools/attn_pmc_probe.py: B = 2, 40 heads, L = 48 832 -- the timed
    region's launch) and (b) the six per-token GEMMs of a block with the executor's shapes and epilogues (tools/gemm_layer_pmc_probe.py).
    Bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB.  A failure leaves the stamped values of profiles/traffic.json in place."""
    t0 = time.time()
    out = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one pass each) around tools/attn_pmc_probe.py and tools/gemm_layer_pmc_probe.py, in this run"}
    att = {c: _pmc_pass([os.path.join(ROOT, "tools", "attn_pmc_probe.py"), "prescaled", "2"], c, "scail_attn4_m16f")["scail_attn4_m16f"] for c in ("FETCH_SIZE", "WRITE_SIZE")}
    out["attention"] = {"traffic_bytes": _kib_to_bytes(att["FETCH_SIZE"], att["WRITE_SIZE"]), "fetch_kib": att["FETCH_SIZE"], "write_kib": att["WRITE_SIZE"],
                        "stamped_value_of_profiles_traffic_json": expected_attn}
    try:
        gm = {c: _pmc_pass([os.path.join(ROOT, "tools", "gemm_layer_pmc_probe.py"), "1"], c, "scail_gemm4_e") for c in ("FETCH_SIZE", "WRITE_SIZE")}
        n = {"e0": 2, "e1": 1, "e3": 2, "e4": 1}            # launches per layer: qkv + cross q, MLP up, attention out + MLP down, cross out
        tf = sum(gm["FETCH_SIZE"]["scail_gemm4_" + k] * cnt for k, cnt in n.items())
        tw = sum(gm["WRITE_SIZE"]["scail_gemm4_" + k] * cnt for k, cnt in n.items())
        out["gemm"] = {"traffic_bytes_per_launch_mean": _kib_to_bytes(tf, tw) / 6.0, "traffic_bytes_per_layer": _kib_to_bytes(tf, tw),
                       "stamped_value_of_profiles_traffic_json": expected_gemm}
    except Exception as e:              # noqa: BLE001
        out["gemm"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    out["seconds"] = round(time.time() - t0, 1)
    return out


def sp_compute_side_leg(net, p, thw, ctx, clip, dev, t1, args):
    """The launch sequence of rank 0 of a 2 / 4 / 8-rank sequence-parallel step through scail_dit_step_sp ON THIS ONE GPU, the collectives
    served by local copies (scail_amd.parallel.LocalCopyBackend; results meaningless, every kernel shape real): the compute side of the
    strong-scaling efficiency, (T_1 / N) / T_N against the step time just measured.  The exchange over xGMI comes on top and is NOT
    in these numbers (no multi-GPU node in this run).  One warm-up + one timed step per N."""
    from scail_amd import lib, ops
    from scail_amd.parallel import LocalCopyBackend, SequenceParallel
    T, H, W = thw
    nh = p["num_attention_heads"]
    g = torch.Generator().manual_seed(99)
    res = {"what": "rank 0's launches of an N-rank step on one GPU, collectives = local copies (compute side only; exchange time NOT included)",
           "s_per_step_1_rank": t1, "ranks": {}}
    old_sp = net.sp
    try:
        for N in (2, 4, 8):
            sp = SequenceParallel(LocalCopyBackend(N))
            net.sp = sp
            h = H // N
            x = torch.randn(1, T, 16, h, W, generator=g).to(dev)
            ref = torch.randn(1, 1, 16, h, W, generator=g).to(dev).to(torch.bfloat16)
            pose = torch.randn(1, T, 16, h // 2, W // 2, generator=g).to(dev).to(torch.bfloat16)
            tt = torch.tensor([700.0, 700.0], device=dev)

            def one():
                v = net.forward_f32(torch.cat([x, x], 0), tt, ctx, None, concat_images=torch.zeros(1, device=dev), image_clip_features=clip,
                                    ref_concat=ref, concat_smpl_render=pose, chunk_dim=3, cfg_pair=not args.no_cfg_pair)
                ops.cfg_euler_(x, v, args.cfg_scale, -0.01)

            one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            one()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            mode = sp.resolve_mode(nh)
            L = (1 + T) * (H // 2) * (W // 2) + T * (H // 4) * (W // 4)
            rows = lib.load().scail_flash_attn_rows_for(1, nh // N if mode == "ulysses" else nh, L if mode == "ulysses" else L // N)
            res["ranks"][str(N)] = {"mode": mode, "s_per_step_one_rank": dt, "compute_only_efficiency": (t1 / N) / dt, "attn_query_tile_rows": int(rows)}
    finally:
        net.sp = old_sp
        net._ws = {}
    return res


def multichar_leg(net, p, thw, ctx, clip, dev, Lt, Lc, args):
    """BASELINE config 5 on one GPU: 2 reference frames + 2 pose streams in one token sequence (L = 60 032), the same 40-layer network, one
    warm-up + two timed steps.  An EXTENSION of the reference (which has one of each, dit...:1559): parity is against the oracle extended
    the same way (tests/test_dit_gpu.py), unpinned by construction.  8-GPU sequence parallel of it: unmeasured."""
    from scail_amd import ops
    T, H, W = thw
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, T, 16, H, W, generator=g).to(dev)
    ref = torch.randn(1, 2, 16, H, W, generator=g).to(dev).to(torch.bfloat16)
    pose = torch.randn(1, 2 * T, 16, H // 2, W // 2, generator=g).to(dev).to(torch.bfloat16)
    tt = torch.tensor([700.0, 700.0], device=dev)

    def one():
        v = net.forward_f32(torch.cat([x, x], 0), tt, ctx, None, concat_images=torch.zeros(1, device=dev), image_clip_features=clip,
                            ref_concat=ref, concat_smpl_render=pose)
        ops.cfg_euler_(x, v, args.cfg_scale, -0.01)

    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one(); one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    hp, wp = H // 2, W // 2
    Lnoise = T * hp * wp
    L = 2 * hp * wp + Lnoise + 2 * T * (H // 4) * (W // 4)
    fl = step_flops(p, L, Lt, Lc)
    net._ws = {}
    return {"EXTENSION_not_in_reference": "2 reference frames + 2 pose streams (BASELINE config 5); parity unpinned by construction",
            "L_tokens": L, "layers": p["num_layers"], "steps_timed": 2, "ms_per_step": dt * 1e3, "latent_tokens_per_s": Lnoise / dt,
            "step_tflop": fl / 1e12, "step_mfma_frac": fl / dt / (PEAK_BF16_TFLOPS * 1e12), "finite": bool(torch.isfinite(x).all().item()),
            "path": "scail_dit_block per layer (C executor) + token assembly in the host", "n_gpus": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=os.environ.get("SCAIL_BENCH_CONFIG", "14b"), choices=list(CONFIGS),
                    help="14b = the headline workload (BASELINE config 2); 14b-2char = BASELINE config 5 (2 ref + 2 pose streams, an extension, flagged)")
    ap.add_argument("--layers", type=int, default=None, help="DEBUG ONLY: fewer layers (result flagged invalid)")
    ap.add_argument("--latent-hw", type=int, nargs=2, default=None, metavar=("H", "W"),
                    help="OTHER RESOLUTION (result flagged invalid for the headline metric): latent height / width, e.g. 60 104 = 480x832")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the config-4 VAE leg after the timed region")
    ap.add_argument("--cfg-scale", type=float, default=4.0)
    ap.add_argument("--no-cfg-pair", action="store_true", help="A/B: evaluate both CFG elements in layer 0 (no SCAIL_DIT_CFG_PAIR)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip config.sp_compute_side and config.multichar after the timed region")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes behind roofline.traffic (the stamped profiles/traffic.json value is reported then)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher -- the same command line the driver uses, one rank per GPU
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # SCAIL_DIST_BACKEND=gloo: TEST vehicle only -- several ranks share GPU 0 and the exchanges are staged through the
    # host (RCCL refuses duplicate devices); lets the N > 1 code path of this script run on a 1-GPU box
    backend = os.environ.get("SCAIL_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from scail_amd import lib, ops, parallel
    from scail_amd.dit import DiffusionTransformer
    from scail_amd.sampler import make_flow_timesteps
    lib.load()
    sp = parallel.init_from_env(backend) if world > 1 else None
    import torch.distributed as dist
    # N > 1: verify every collective of the layer exchange on rank-stamped data before anything is timed; the result
    # (ranks, backend, RCCL version) goes into the JSON line so a scaling run can be validated from its output alone
    sp_check = sp.self_check(dev) if sp is not None else None

    p, (T, H, W), Lt, Lc = CONFIGS[args.config]
    if args.latent_hw is not None:
        H, W = args.latent_hw
    p = dict(p)
    if args.layers is not None:
        p["num_layers"] = args.layers
    net = DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=81, latent_width=300,
                               latent_height=300, share_adaln=True, use_i2v_clip=True, device=dev, init_seed=1234, **p)
    net.cache_conditioning = False          # recompute text/CLIP K,V every step like the reference
    net.sp = sp
    n_char = 2 if args.config == "14b-2char" else 1
    # The product default at every N: a network evaluation is ONE call of the C executor -- scail_dit_step on one rank, scail_dit_step_sp on
    # a sequence-parallel rank (same kernels; only the collectives of the per-layer exchange come back to the host through the exchange
    # callback) -- and the kernel times come from the executor's own event pairs.  The multi-character extension assembles its tokens in the
    # host and runs every block as one executor call (scail_dit_block / scail_dit_block_sp).  SCAIL_C_STEP=0: the per-op host path
    # (cross-check), timed by bracketing the tagged launches.
    use_c = net.use_c_step
    timer = KernelTimer()
    net.kernel_timer = None if use_c else timer

    # ---- synthetic inputs, identical on every rank, already on the GPU ----
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(1, T, 16, H, W, generator=g).to(dev)
    ref = torch.randn(1, n_char, 16, H, W, generator=g).to(dev).to(torch.bfloat16)
    pose = torch.randn(1, n_char * T, 16, H // 2, W // 2, generator=g).to(dev).to(torch.bfloat16)
    c_ctx = torch.randn(1, Lt, p["text_dim"], generator=g)
    c_ctx[:, 64:] = 0                        # zeroed padding rows (umt5.py:516-522)
    uc_ctx = torch.zeros(1, Lt, p["text_dim"])
    uc_ctx[:, :1] = torch.randn(1, 1, p["text_dim"], generator=g)
    ctx = torch.cat([uc_ctx, c_ctx], 0).to(dev).to(torch.bfloat16)
    clip = torch.randn(1, Lc, 1280, generator=g).to(dev).to(torch.bfloat16)
    chunk_dim = None
    if sp is not None:
        chunk_dim = 3 if H < W else 4
        sp.check_latent(H, W, chunk_dim)
        x, ref, pose = sp.chunk(x, chunk_dim), sp.chunk(ref, chunk_dim), sp.chunk(pose, chunk_dim)
    sig = make_flow_timesteps(0, 50, shift_scale=5, mode="normal")
    dummy = torch.zeros(1, device=dev)

    def step(i):
        xin = torch.cat([x, x], 0)
        t = (sig[i] * 1000.0).repeat(2).to(dev)
        # cfg_pair: what the sampler's VanillaCFG passes with its [x; x] batch (scail_amd/sampler.py): the executor evaluates layer 0 up to its
        # first cross attention once (SCAIL_DIT_CFG_PAIR, result-preserving; accounted for in step_tflop_executed below)
        v = net.forward_f32(xin, t, ctx, None, concat_images=dummy, ref_concat=ref, concat_smpl_render=pose,
                            image_clip_features=clip, chunk_dim=chunk_dim, cfg_pair=not args.no_cfg_pair)
        ops.cfg_euler_(x, v, args.cfg_scale, float(sig[i + 1] - sig[i]))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if use_c:
        if net._cstep is None:               # --warmup 0: create the executor handle outside the timed region
            from scail_amd.cstep import CStep
            net._cstep = CStep(net, net.prepare())
        net._cstep.profile(True)
    timer.enabled = True
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    elapsed = torch.tensor([t1 - t0], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    finite = bool(torch.isfinite(x).all().item())
    x_full = sp.gather_to_rank0(x, chunk_dim) if sp is not None else x       # after the timed region: result fingerprint
    x_abs_mean = float(x_full.abs().mean().item())

    hp, wp = H // 2, W // 2
    Lnoise = T * hp * wp
    L = n_char * hp * wp + Lnoise + n_char * T * (H // 4) * (W // 4)
    t_step = elapsed / args.steps
    nh = p["num_attention_heads"]
    gemm_ms = gemm_n = None
    if use_c:
        cs = net._cstep
        ms, n_att = cs.profile_read(cs.PROF_SELF_ATTN)
        attn_ms = ms / n_att if n_att else None
        gemm_ms, gemm_n = cs.profile_read(cs.PROF_GEMM)
        # workgroups of the timed self-attention launches that restarted after an overflow of the optimistic pass (0 on this synthetic data;
        # on trained weights it says whether the measured rate holds: profiles/r06_attn_restart_probe.log)
        attn_restarts = cs.profile_read(cs.PROF_ATTN_RESTARTS)[1]
        xch = None
        if sp is not None:
            # the EXPOSED part of the layer exchange on this rank: event pairs around nothing but the launch stream's waits for the collectives
            fw, n_fw = cs.profile_read(cs.PROF_XCH_FWD_WAIT)
            bw, n_bw = cs.profile_read(cs.PROF_XCH_BACK_WAIT)
            xch = {"fwd_wait_ms_per_step": fw / args.steps, "back_wait_ms_per_step": bw / args.steps, "waits_per_step": (n_fw + n_bw) / args.steps,
                   "rank": rank, "timed_by": "scail_dit_profile categories 3 / 4 (event pairs around the stream waits inside the C executor)"}
        cs.profile(False)
    else:
        attn_ms = timer.mean_ms("self_attn")
        n_att = len(timer.events.get("self_attn", []))
        attn_restarts = None
    # one launch = local queries x all keys; in ulysses mode a launch covers heads/world heads of ONE source rank's queries
    sp_mode = sp.resolve_mode(nh) if sp is not None else "none"
    attn_heads = nh // world if sp_mode == "ulysses" else nh
    attn_B = 2 if sp is None else 1          # the sequence-parallel path launches per CFG batch element (comm / compute overlap)
    attn_Lq = L if sp_mode == "ulysses" else L // world      # ulysses: all ranks' query rows of this rank's heads
    # scail_dit_step runs the LAST layer's queries / out-projection / cross attention / MLP on the noise tokens only (its other rows never
    # reach the final layer; csrc/dit_step.hip): that launch has Lnoise / L of the FLOPs, and the means below are over all launches
    # Two result-preserving prunings of the executor (n_char == 1), both accounted for here:
    #   last layer: its output is read at the noise rows only -> out-projection / cross attention / MLP on those rows; the queries too, except
    #               in the ulysses exchange (rank-major full sequence); the all-gather rank prunes its Q projection as well;
    #   cfg pair:   layer 0 up to the first cross attention runs for ONE of the two CFG elements (SCAIL_DIT_CFG_PAIR).
    pruned = use_c and n_char == 1 and p["num_layers"] > 0
    pair = pruned and not args.no_cfg_pair and p["num_layers"] > 1
    nl_ = p["num_layers"]
    D0, FF0 = p["hidden_size"], p["inner_hidden_size"]
    frac_noise = Lnoise / L
    q_pruned = pruned and sp_mode != "ulysses"
    # self-attention FLOPs of this rank per step, in units of one (element, layer) launch of 4 Lq Lk 128 heads
    unit = 4.0 * attn_Lq * L * 128 * attn_heads
    att_units = 2.0 * nl_ - (1.0 if pair else 0.0) - (2.0 * (1.0 - frac_noise) if q_pruned else 0.0)
    att_launches = nl_ if attn_B == 2 else 2 * nl_ - (1 if pair else 0)      # one rank: a launch covers both elements (layer 0 of a cfg pair: one)
    attn_flops = unit * att_units / att_launches                  # mean over the launches of a step
    ach = attn_flops / (attn_ms * 1e-3) / 1e12 if attn_ms else None
    fl = step_flops(p, L, Lt, Lc)
    # FLOPs the step actually executes (whole job): the reference's algorithmic count minus what the two prunings skip
    fl_exec = fl
    if pruned:
        fl_exec -= 2 * (L - Lnoise) * ((4 * L * D0 if q_pruned else 0.0) + 4 * (Lt + Lc) * D0 + 2 * D0 * (3 * D0 + 2 * FF0)
                                       + (2 * D0 * D0 if sp_mode == "allgather" else 0.0))
    if pair:
        fl_exec -= 4.0 * L * L * D0 + 2.0 * L * D0 * 4 * D0
    # HBM/fabric bytes per launch of the dominant kernel: measured offline with rocprofv3 --pmc (separate passes, guide
    # corrections; profiles/), committed with the git blob id of the kernel source it was measured on -- dropped (null)
    # when the kernel source (the generated csrc/attn4.s, or attn.hip for shapes the 8-wave kernel serves) has changed since
    # or the launch shape differs, so the field cannot go stale silently
    traffic = None
    Dm = p["hidden_size"]
    if sp_mode == "ulysses":                 # column thirds of the received (L, 3 Dn) q | k | v matrix of this rank's head group (parallel.py)
        strides = (3 * Dm // world, 3 * Dm // world, Dm // world)
    elif sp_mode == "allgather":             # local q rows of the fused qkv buffer, gathered k | v rows
        strides = (3 * Dm, 2 * Dm, Dm)
    else:
        strides = (3 * Dm, 3 * Dm, Dm)
    which = lib.load().scail_flash_attn_kernel_for(*strides, attn_Lq, L, 0, 1)
    ksrc = "attn4.s" if which == 4 else "attn.hip"
    kname = ("scail_attn4_m16f" + ("_q3" if lib.load().scail_flash_attn_rows_for(attn_B, attn_heads, attn_Lq) == 192 else "")
             + " (hand-scheduled 4-wave flash attention, 16x16x32 MFMAs, queries in log2 units; csrc/attn4.s)"
             if which == 4 else "flash_attn_swp_kernel<4, 4, 0, 1>")
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["flash_attn_self"]
        blob = _git_blob_sha1(os.path.join(ROOT, "scail_amd", "csrc", ksrc))
        if tr["shape"] == {"B": attn_B, "heads": attn_heads, "Lq": attn_Lq, "Lk": L} and tr.get("source") == ksrc and tr.get("source_blob") == blob:
            traffic = tr["traffic_bytes"]
    except Exception:
        pass
    out = {
        "metric": "denoising-step latent tokens/sec", "value": Lnoise / t_step, "unit": "latent tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_step * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"SCAIL-{args.config} DiT sampler step (batch-2 CFG forward + Euler), {8 * H}x{8 * W}x81f latent "
                               f"({T},16,{H},{W}), L={L} tokens (ref+noise+pose), text {Lt}, clip {Lc}, "
                               f"{p['num_layers']} layers, random-init bf16 weights",
                   "parallelism": f"sp{world}" + (f"-{sp_mode}" if sp is not None else ""), "cond_cache": False, "noise_tokens": Lnoise, "all_tokens_x_batch": 2 * L,
                   "step_tflop": fl / 1e12, "step_tflop_executed": fl_exec / 1e12, "step_mfma_frac": fl_exec / t_step / (world * PEAK_BF16_TFLOPS * 1e12),
                   "finite": finite, "x_abs_mean": x_abs_mean, "sp_check": sp_check,
                   "result_preserving_prunings": {"last_layer_noise_rows_only": bool(pruned), "cfg_pair_layer0_once": bool(pair)},
                   "attn_query_tile_rows": int(lib.load().scail_flash_attn_rows_for(attn_B, attn_heads, attn_Lq))},
        "roofline": {"bound": "mfma", "kernel": kname + " (self-attention)", "achieved": ach,
                     "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": (ach / PEAK_BF16_TFLOPS) if ach else None,
                     "traffic": traffic, "flop_per_launch": attn_flops, "ms_per_launch": attn_ms,
                     "launches_timed": n_att,
                     "attn_restarts_per_launch": (attn_restarts / n_att) if (attn_restarts is not None and n_att) else None,
                     "timed_by": "scail_dit_profile (event pairs inside the C executor)" if use_c else "HIP events around the tagged host launches"},
    }
    if use_c and sp is not None and xch is not None:
        # what the exchange adds to this rank's step (the rest of it runs under the other CFG element's kernels); with the per-rank compute
        # known from config.sp_compute_side of the 1-GPU line, a scaling run explains itself: t_step(N) ~ compute side + these waits
        out["config"]["exchange_exposed"] = xch
        out["config"]["exchange_collectives_per_layer"] = 4 if sp_mode == "ulysses" else 2
        out["config"]["rccl_env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))}
    if not use_c:
        out["config"]["path"] = "per-op host path (scail_amd.dit._run)"
    elif n_char > 1:
        out["config"]["path"] = ("scail_dit_block_sp" if sp is not None else "scail_dit_block") + " per layer (C executor) + token assembly in the host"
    else:
        out["config"]["path"] = ("scail_dit_step_sp (C executor; collectives through the exchange callback)" if sp is not None
                                 else "scail_dit_step (C executor, product default)")
    if gemm_ms:
        # the six per-token GEMMs of a block (qkv, attention out, cross q, cross out, MLP up, MLP down), all launches of the timed region
        D_, FF_ = p["hidden_size"], p["inner_hidden_size"]
        gflop = 2.0 * (2 * L // world) * D_ * (3 * D_ + 3 * D_ + 2 * FF_) * p["num_layers"] * args.steps      # this rank's token rows
        if pruned:
            gflop -= 2.0 * (2 * (L - Lnoise) // world) * D_ * (3 * D_ + 2 * FF_ + (D_ if sp_mode == "allgather" else 0)) * args.steps
        if pair:
            gflop -= 2.0 * (L // world) * D_ * 4 * D_ * args.steps
        g_ach = gflop / (gemm_ms * 1e-3) / 1e12
        g_traffic = None
        try:
            trg = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["gemm4_step"]
            if trg.get("source_blob") == _git_blob_sha1(os.path.join(ROOT, "scail_amd", "csrc", "gemm4.s")) and trg["shape"] == {"M": 2 * L // world, "D": D_, "FF": FF_}:
                g_traffic = trg["traffic_bytes_per_launch_mean"]
        except Exception:
            pass
        out["roofline_gemm"] = {"bound": "mfma", "kernel": "scail_gemm4_e0 / e1 / e3 / e4 (generated 4-wave GEMM, 256x256x64 tiles; csrc/gemm4.s): the six per-token GEMMs of a block",
                                "achieved": g_ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": g_ach / PEAK_BF16_TFLOPS,
                                "traffic": g_traffic, "flop_per_launch": gflop / gemm_n, "ms_per_launch": gemm_ms / gemm_n, "launches_timed": gemm_n,
                                "timed_by": "scail_dit_profile (event pairs inside the C executor)"}
    if n_char > 1:
        out["config"]["EXTENSION_not_in_reference"] = (f"{n_char} reference frames + {n_char} pose streams in one token sequence (BASELINE config 5); the reference "
                                                       "has one of each (dit...:1559), so parity is against the oracle extended the same way: unpinned by construction")
    if args.layers is not None:
        out["config"]["INVALID_debug_layers"] = args.layers
    if args.latent_hw is not None:
        out["config"]["INVALID_other_resolution"] = list(args.latent_hw)
    def leg(fn, *a):
        """a leg after the timed region must never cost the headline line: a failure is reported in its place"""
        try:
            return fn(*a)
        except Exception as e:          # noqa: BLE001
            try:
                torch.cuda.synchronize()
            except Exception:           # noqa: BLE001
                pass
            return {"error": f"{type(e).__name__}: {e}"[:500]}

    if rank == 0 and world == 1 and use_c and args.config == "14b" and args.latent_hw is None and not args.no_extra_legs:
        out["config"]["sp_compute_side"] = leg(sp_compute_side_leg, net, p, (T, H, W), ctx, clip, dev, t_step, args)
        out["config"]["multichar"] = leg(multichar_leg, net, p, (T, H, W), ctx, clip, dev, Lt, Lc, args)
    if rank == 0 and world == 1 and use_c and args.config == "14b" and args.latent_hw is None and args.layers is None and n_char == 1 and not args.no_pmc:
        # fabric traffic of the two big kernels from PMC counters collected in THIS run (the stamped file stays the fallback and the cross-check)
        pm = leg(pmc_traffic_leg, traffic, out.get("roofline_gemm", {}).get("traffic"))
        out["roofline"]["traffic_in_run"] = pm
        stamped = "profiles/traffic.json (stamped with the git blob of the kernel source)"
        if isinstance(pm, dict) and pm.get("attention", {}).get("traffic_bytes"):
            out["roofline"]["traffic"] = pm["attention"]["traffic_bytes"]
            out["roofline"]["traffic_source"] = "PMC counters of this run (roofline.traffic_in_run)"
        else:
            out["roofline"]["traffic_source"] = stamped
        if "roofline_gemm" in out:
            if isinstance(pm, dict) and pm.get("gemm", {}).get("traffic_bytes_per_launch_mean"):
                out["roofline_gemm"]["traffic"] = pm["gemm"]["traffic_bytes_per_launch_mean"]
                out["roofline_gemm"]["traffic_source"] = "PMC counters of this run (roofline.traffic_in_run.gemm)"
            else:
                out["roofline_gemm"]["traffic_source"] = stamped
    if rank == 0 and world == 1 and not args.no_vae:
        out["config"]["vae"] = leg(vae_leg, dev, (not args.no_pmc) and args.config == "14b" and args.layers is None)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(p, L)
        t_cpu = p["num_layers"] * cb["t_block"]
        out["cpu_baseline"] = {
            "value": Lnoise / t_cpu, "unit": "latent tokens/s", "cores": _physical_cores(), "threads": cb["threads"],
            "hardware_threads": os.cpu_count(), "kind": "port, extrapolated",
            "fit": {"per_token_s": {"c": cb["c"], "a_per_token": cb["a"], "rms_residual_s": cb["res_tok"]},
                    "self_attention_s": {"a_per_token": cb["a2"], "b_per_token2": cb["b"], "rms_residual_s": cb["res_att"]},
                    "longest_timed_L": cb["max_timed_L"], "extrapolation_in_L": L / cb["max_timed_L"], "pairs_timed_of_80": cb["pairs_timed"],
                    "attention_share_at_bench_L": cb["t_att"] / cb["t_block"],
                    # one standard error of the fitted block time at the bench length, carried to the value (the fit's own uncertainty;
                    # box-to-box spread of the host is larger: quote the baseline as "about 2-2.5 tokens/s")
                    "block_s_at_bench_L": cb["t_block"], "block_s_sigma": cb["sigma_block"],
                    "fit_standard_error": Lnoise / t_cpu * cb["sigma_block"] / cb["t_block"]},
            # spread of this figure over the GPU boxes of the pool (same code, rounds 3-5: host load and NUMA placement differ from box to box);
            # much wider than the fit's standard error above -- quote the baseline as "about 2-2.5 tokens/s"
            "box_to_box_range": [1.9, 2.5],
            # Does the port cost what the reference costs?  Both timed side by side in the BUILD container (the reference cannot travel):
            # the reference's own AdaLNMixin.layer_forward inside DiffusionTransformer.forward vs O.block inside O.dit_forward, same
            # weights / inputs / 8 threads, D = 5120, B = 2: 2.35 vs 2.83 s at L = 1008, 6.52 vs 9.04 s at L = 2128 (outputs equal to 3e-6).
            # The port is the SLOWER of the two: the reference's block on this host would give about value x this ratio.
            # tools/cpu_port_vs_reference.py, profiles/r06_cpu_port_vs_reference.log; a constant, not measured in this run.
            "port_vs_reference_time_ratio": 1.34,
            "sample": f"oracle block (fp32, torch CPU, {cb['threads']} threads = fastest of "
                      + ", ".join(f"{k}: {v * 1e3:.0f} ms" for k, v in sorted(cb["probe"].items())) + " on the 2016x5120x15360 projection) at full "
                      f"width D={p['hidden_size']}, B=2, after one warm-up call: whole block at L = "
                      + ", ".join(f"{Ls} ({dt:.2f} s, of which self-attention {ta:.2f} s)" for Ls, dt, ta in cb["blocks"])
                      + "; self-attention alone (O.sdpa) on " + ", ".join(f"{pr} of 80 (batch, head) pairs at L = {La} ({dt:.2f} s)" for La, pr, dt in cb["att"])
                      + f"; fits: per-token part {cb['c']:.2f} + {cb['a']:.3e} L s, self-attention {cb['a2']:.3e} L + {cb['b']:.3e} L^2 s "
                      f"-> {cb['t_tok']:.1f} + {cb['t_att']:.1f} s per block at L = {L}, x {p['num_layers']} layers = {t_cpu:.0f} s per step "
                      f"(embeddings / final layer < 0.1 % not included)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
