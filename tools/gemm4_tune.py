#!/usr/bin/env python
"""GPU check + A/B timing of the generated gemm4 kernels (scail_amd/asmgen/gemm4.py) against the q8 kernel of csrc/gemm.hip and the
vendor library (torch F.linear) on the config-2 GEMM shapes.  The variants need the measurement build (SCAIL_ABLATIONS=1).
One JSON line per case."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib, ops  # noqa: E402
from scail_amd import lib as L  # noqa: E402

DEV = "cuda"


def timeit(fn, iters):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def ref_rows(x, w, b, rows, epi, resid=None, gate=None, rpb=0):
    y = x[rows].float() @ w.float().t() + (b if b is not None else 0)
    if epi == L.EPI_GELU_TANH:
        y = torch.nn.functional.gelu(y, approximate="tanh")
    if epi == L.EPI_RESID:
        if gate is not None:
            y = y * gate[(rows // rpb)]
        y = resid[rows].float() + y
    return y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--variants", default=",spread,lds_dma,dma2,dma2_c2,dma2_rd1,dma2_d05")
    ap.add_argument("--ablations", default="dma2_abl_dma,dma2_abl_lds,dma2_abl_dma_lds")
    ap.add_argument("--variants8", default="-", help="gemm8 variants (\"-\" = skip gemm8)")
    ap.add_argument("--vendor", action="store_true")
    ap.add_argument("--skip-check", action="store_true")
    a = ap.parse_args()
    lib.load()
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    setk = lambda v: lib.tune_set("gemm4_kernel" + (":" + v if v else ""), 0)

    # ---- correctness of the four epilogues on ragged-M shapes (sampled rows vs fp32) ----
    for (M, N, K) in ((2048 + 136, 512, 192), (4096, 768, 320), (2600 + 8, 256, 128)):
        x = rn(M, K).to(torch.bfloat16)
        w = (rn(N, K) * K ** -0.5).to(torch.bfloat16)
        b = rn(N)
        resid = rn(M, N).to(torch.bfloat16)
        gate = rn(3, N)
        rpb = (M + 2) // 3
        rows = torch.cat([torch.arange(0, 64), torch.arange(M - 140, M), torch.randint(0, M, (128,))]).to(DEV)
        for name, epi, kw in (("bias", L.EPI_BIAS, {}), ("nobias", L.EPI_BIAS, dict(nobias=True)), ("gelu", L.EPI_GELU_TANH, {}),
                              ("resid+gate", L.EPI_RESID, dict(resid=True, gate=True)), ("resid", L.EPI_RESID, dict(resid=True))):
            res = {}
            for mode in (8, 4, 0):
                if a.skip_check:
                    res[mode] = 0.0
                    continue
                lib.tune_set("gemm4", mode)
                y = resid.clone() if kw.get("resid") else torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
                ops.gemm(x, w, None if kw.get("nobias") else b, out=y, epilogue=epi, resid=y if kw.get("resid") else None,
                         gate=gate if kw.get("gate") else None, rows_per_batch=rpb if kw.get("gate") else 0)
                want = ref_rows(x, w, None if kw.get("nobias") else b, rows, epi, resid if kw.get("resid") else None,
                                gate if kw.get("gate") else None, rpb)
                res[mode] = float((y[rows].float() - want).abs().max())
                which = lib.load().scail_gemm_kernel_for(K, N, N if kw.get("resid") else 0, M, N, K, epi)
            lib.tune_set("gemm4", 0)
            print(json.dumps({"check": [M, N, K], "epi": name, "max_err_gemm8": res[8], "max_err_gemm4": res[4], "max_err_q8": res[0],
                              "ok": res[4] < 6e-2 and res[8] < 6e-2}), flush=True)

    # ---- timing on the step's shapes ----
    M = 97664
    for (N, K, epi, tag) in ((15360, 5120, L.EPI_BIAS, "qkv"), (5120, 5120, L.EPI_RESID, "out-proj + gate/resid"),
                             (13824, 5120, L.EPI_GELU_TANH, "mlp up + gelu"), (5120, 13824, L.EPI_RESID, "mlp down + gate/resid")):
        x = rn(M, K).to(torch.bfloat16)
        w = (rn(N, K) * 0.02).to(torch.bfloat16)
        b = rn(N)
        y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        gate = rn(2, N)
        kw = dict(resid=y, gate=gate, rows_per_batch=M // 2) if epi == L.EPI_RESID else {}
        fl = 2.0 * M * N * K
        out = {"shape": [M, N, K], "what": tag}
        lib.tune_set("gemm4", 0)
        ms = timeit(lambda: ops.gemm(x, w, b, out=y, epilogue=epi, **kw), a.iters)
        out["q8_TFLOPs"] = fl / ms / 1e9
        rows = torch.randint(0, M, (256,), device=DEV)
        y_q8 = y[rows].float().clone() if epi == L.EPI_BIAS else None
        lib.tune_set("gemm4", 8)
        for v in ([t for t in a.variants8.split(",") if t or not a.variants8 == "-"] if epi == L.EPI_BIAS and a.variants8 != "-" else ([""] if a.variants8 != "-" else [])):
            setk(v)
            ms = timeit(lambda: ops.gemm(x, w, b, out=y, epilogue=epi, **kw), a.iters)
            out["gemm8" + ("_" + v if v else "") + "_TFLOPs"] = fl / ms / 1e9
        setk("")
        lib.tune_set("gemm4", 4)
        names = a.variants.split(",") + ([v for v in a.ablations.split(",") if v] if epi == L.EPI_BIAS else [])
        for v in (names if epi == L.EPI_BIAS else [""]):
            setk(v)
            ms = timeit(lambda: ops.gemm(x, w, b, out=y, epilogue=epi, **kw), a.iters)
            out["gemm4" + ("_" + v if v else "") + "_TFLOPs"] = fl / ms / 1e9
            if y_q8 is not None and "abl" not in v:
                out["gemm4" + ("_" + v if v else "") + "_maxdiff_vs_q8"] = float((y[rows].float() - y_q8).abs().max())
        setk("")
        lib.tune_set("gemm4", 0)
        if a.vendor:
            ms = timeit(lambda: torch.nn.functional.linear(x, w), a.iters)
            out["vendor_TFLOPs"] = fl / ms / 1e9
        print(json.dumps(out), flush=True)
        del x, w, y


if __name__ == "__main__":
    main()
