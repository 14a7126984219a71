"""Build libscail_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/scail_hip.h)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
SOURCES = ["errors.hip", "rowops.hip", "gemm.hip", "attn.hip", "conv.hip", "dit_step.hip", "vae_exec.hip"]
ARCH = "gfx950"
# per-file extra flags hook (e.g. {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]} keeps MFMA results of 256-thread
# kernels in arch VGPRs: used while measuring the removed 4-wave attention variants; no shipped kernel needs one)
EXTRA_FLAGS = {}
# SCAIL_ABLATIONS=1: also compile the timing-ablation kernel variants (wrong results on purpose) that tools/microbench.py
# selects through scail_tune_set; the shipped library is built without them and rejects their codes.
ABLATIONS = os.environ.get("SCAIL_ABLATIONS", "0") not in ("", "0")
LIB = os.path.join(PKG, "libscail_hip_abl.so" if ABLATIONS else "libscail_hip.so")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(PKG), "include", "scail_hip.h"),
                        os.path.join(os.path.dirname(PKG), "include", "scail_dit.h"),
                        os.path.join(os.path.dirname(PKG), "include", "scail_vae.h"),
                        os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(os.path.dirname(PKG), "build_abl" if ABLATIONS else "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + (["-DSCAIL_ABLATIONS"] if ABLATIONS else []) + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
