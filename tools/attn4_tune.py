#!/usr/bin/env python
"""GPU check + A/B timing of the generated attn4 schedules (scail_amd/asmgen/attn4.py) against the 8-wave kernel.
Needs the measurement build for the variants:  SCAIL_ABLATIONS=1 python -m scail_amd.build ; SCAIL_ABLATIONS=1 python tools/attn4_tune.py
One JSON line per case (stdout)."""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib, ops  # noqa: E402

DEV = "cuda"


def ref_rows(q, k, v, rows, heads_sel):
    out = {}
    for b in range(q.shape[0]):
        for h in heads_sel:
            sl = slice(h * 128, (h + 1) * 128)
            s = q[b, rows, sl].float() @ k[b, :, sl].float().t() / math.sqrt(128.0)
            out[(b, h)] = torch.softmax(s, dim=-1) @ v[b, :, sl].float()
    return out


def check(name, o, ref, rows):
    worst = 0.0
    for (b, h), r in ref.items():
        got = o[b, rows, h * 128:(h + 1) * 128].float()
        worst = max(worst, float((got - r).abs().max()))
    return worst


def timeit(fn, iters):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=48832)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--variants", default=",dmamid,sm48")
    ap.add_argument("--ablations", default="abl_fma,abl_fma_c4,abl_fma_sm44")
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--prescaled", action="store_true", help="hand q over in log2 units (scale == 0 path: the fold variants need it)")
    ap.add_argument("--full", action="store_true", help="also time the 40-head launch of the best variant")
    ap.add_argument("--full-also", default="", help="more variants for the 40-head launch (comma separated; '' = the shipped kernel)")
    a = ap.parse_args()
    lib.load()
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
    variants = a.variants.split(",")
    setk = lambda v: lib.tune_set("attn4_kernel" + (":" + v if v and v != "default" else ""), 0)      # "" / "default" = the shipped kernel

    if not a.skip_check:
        # ---- correctness: ragged Lq, several tile counts (all remainder paths), spiked keys (rescale), 2 segments ----
        for (B, H, Lq, Lk, nseg) in ((1, 2, 700, 1024, 1), (2, 2, 300, 576, 1), (1, 1, 256, 512, 1), (1, 2, 520, 1088, 1), (1, 2, 333, 640, 2)):
            D = H * 128
            q = rn(B, Lq, D)
            ks, vs = rn(nseg, B, Lk, D), rn(nseg, B, Lk, D)
            ks[-1, 0, Lk - 5, :128] = (q[0, 7, :128].float() * 3).to(torch.bfloat16)
            ks[0, 0, 70, :128] = (q[0, 9, :128].float() * 2).to(torch.bfloat16)
            vts = torch.stack([ops.transpose_v(vs[s], H) for s in range(nseg)])
            rows = torch.arange(Lq, device=DEV)
            kcat, vcat = torch.cat(list(ks), 1), torch.cat(list(vs), 1)
            kw = dict(n_seg=nseg, k_seg_stride=ks.stride(0), vt_seg_stride=vts.stride(0))
            if a.prescaled:
                q = (q.float() * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
                ref = ref_rows(q.float() / ops.ATTN_LOG2_SCALE, kcat, vcat, rows, range(H))
                kw["q_prescaled"] = True
            else:
                ref = ref_rows(q, kcat, vcat, rows, range(H))
            lib.tune_set("attn4", 0)
            o_old = ops.flash_attn(q, ks[0], vts[0], **kw)
            e_old = check("old", o_old, ref, rows)
            lib.tune_set("attn4", 1)
            for var in variants:
                setk(var)
                for thr in (8, 0):
                    lib.tune_set("attn4_thr", thr)
                    o = ops.flash_attn(q, ks[0], vts[0], **kw)
                    torch.cuda.synchronize()
                    e = check("new", o, ref, rows)
                    print(json.dumps({"check": [B, H, Lq, Lk, nseg], "variant": var, "thr": thr, "max_err_attn4": e, "max_err_8wave": e_old,
                                      "ok": bool(e < 2e-2 and math.isfinite(e))}), flush=True)
            lib.tune_set("attn4_thr", 8)

    # ---- timing on the config-2 slice (B = 2, `heads` heads of the 40, L keys) ----
    D = a.heads * 128
    qkv = rn(2, a.L, 3 * D)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    vt = ops.transpose_v(v, a.heads)
    out = torch.empty(2, a.L, D, device=DEV, dtype=torch.bfloat16)
    fl = 4.0 * a.L * a.L * 128 * a.heads * 2
    rows = torch.cat([torch.arange(0, 64), torch.arange(a.L - 64, a.L), torch.randint(0, a.L, (64,))]).to(DEV)
    akw = {}
    if a.prescaled:
        qs = (q.float() * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
        ref = ref_rows((qs.float() / ops.ATTN_LOG2_SCALE), k, v, rows, (0, a.heads - 1))
        q = qs
        akw = dict(q_prescaled=True)
    else:
        ref = ref_rows(q, k, v, rows, (0, a.heads - 1))
    lib.tune_set("attn4", 0)
    med, best = timeit(lambda: ops.flash_attn(q, k, vt, out=out, **akw), a.iters)
    print(json.dumps({"kernel": "8-wave swp (round 1)", "ms": med, "TFLOPs": fl / med / 1e9, "best_TFLOPs": fl / best / 1e9,
                      "max_err": check("old", out, ref, rows)}), flush=True)
    lib.tune_set("attn4", 1)
    results = []
    for var in variants + [x for x in a.ablations.split(",") if x]:
        setk(var)
        out.zero_()
        med, best = timeit(lambda: ops.flash_attn(q, k, vt, out=out, **akw), a.iters)
        err = check("new", out, ref, rows)
        results.append((med, var))
        print(json.dumps({"kernel": f"attn4 variant {var}", "ms": med, "TFLOPs": fl / med / 1e9, "best_TFLOPs": fl / best / 1e9, "max_err": err}), flush=True)
    if a.full:
        bestvar = min(r for r in results if "abl" not in r[1])[1]
        setk(bestvar)
        D = 40 * 128
        qkv = rn(2, a.L, 3 * D)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        vt = ops.transpose_v(v, 40)
        out = torch.empty(2, a.L, D, device=DEV, dtype=torch.bfloat16)
        fl = 4.0 * a.L * a.L * 128 * 40 * 2
        if a.prescaled:
            q = (q.float() * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
        for var in [bestvar] + [x for x in a.full_also.split(",") if x]:
            setk(var)
            med, best = timeit(lambda: ops.flash_attn(q, k, vt, out=out, **akw), 3)
            print(json.dumps({"kernel": f"attn4 variant {var}, 40 heads", "ms": med, "TFLOPs": fl / med / 1e9, "best_TFLOPs": fl / best / 1e9}), flush=True)
        setk(bestvar)
        for xcd in (1, 0):
            lib.tune_set("attn4_xcd", xcd)
            med, best = timeit(lambda: ops.flash_attn(q, k, vt, out=out, **akw), 3)
            print(json.dumps({"kernel": f"attn4 variant {bestvar}, 40 heads, xcd-aware ids {xcd}", "ms": med, "TFLOPs": fl / med / 1e9, "best_TFLOPs": fl / best / 1e9}), flush=True)
        lib.tune_set("attn4_xcd", 1)
        lib.tune_set("attn4", 0)
        med, best = timeit(lambda: ops.flash_attn(q, k, vt, out=out, **akw), 3)
        print(json.dumps({"kernel": "8-wave swp, 40 heads", "ms": med, "TFLOPs": fl / med / 1e9, "best_TFLOPs": fl / best / 1e9}), flush=True)


if __name__ == "__main__":
    main()
