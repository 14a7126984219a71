#!/usr/bin/env python
"""profiles/traffic.json from a PMC summary (tools/run_r03_*.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, one counter set per
pass, tools/rocpd_counters.py per-kernel means).  Bytes = 2 * FETCH_SIZE KiB * 1024 (the guide's gfx950 correction: 128-byte
requests tallied at 64 bytes) + WRITE_SIZE KiB * 1024.  Each entry is stamped with the git blob id of the kernel source in the
working tree: run it on the SAME tree the PMC passes ran on (bench.py drops an entry whose blob differs from the tree's).
usage: python tools/update_traffic.py <pmc_summary.txt> <note>"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def blob(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def main():
    src, note = sys.argv[1], sys.argv[2]
    d = {}
    for line in open(src):
        m = re.match(r"(\S+)\s+(\S+)\s+mean/launch\s+(\S+)\s+launches\s+(\d+)", line)
        if m:
            d[(m.group(1), m.group(2))] = float(m.group(3))
    path = os.path.join(ROOT, "profiles", "traffic.json")
    tr = json.load(open(path))
    if ("scail_attn4_m16f", "FETCH_SIZE") in d:      # (a summary may hold only some of the kernels: the other entries stay)
      f, w = d[("scail_attn4_m16f", "FETCH_SIZE")], d[("scail_attn4_m16f", "WRITE_SIZE")]
      tr["flash_attn_self"] = {
        "shape": {"B": 2, "heads": 40, "Lq": 48832, "Lk": 48832}, "fetch_kib": f, "write_kib": w,
        "traffic_bytes": (2 * f + w) * 1024,
        "kernel": "scail_attn4_m16f (16x16x32 MFMAs, queries in log2 units, optimistic hot loop, XCD-aware workgroup ids)",
        "source": "attn4.s", "source_blob": blob(os.path.join(ROOT, "scail_amd", "csrc", "attn4.s")),
        "measured": note + "; algorithmic bytes per launch = Q + K + V^T + O of 80 (batch, head) slices = 4.0 GB"}
    # the six per-token GEMMs of a block: launches per layer e0 x 2 (qkv, cross q), e3 x 2 (attention out, MLP down), e1 (MLP up), e4 (cross out)
    n = {"e0": 2, "e1": 1, "e3": 2, "e4": 1}
    if all(("scail_gemm4_" + k, "FETCH_SIZE") in d for k in n):
      tf = sum(d[("scail_gemm4_" + k, "FETCH_SIZE")] * c for k, c in n.items())
      tw = sum(d[("scail_gemm4_" + k, "WRITE_SIZE")] * c for k, c in n.items())
      tr["gemm4_step"] = {
        "shape": {"M": 97664, "D": 5120, "FF": 13824}, "fetch_kib_per_layer": tf, "write_kib_per_layer": tw,
        "traffic_bytes_per_layer": (2 * tf + tw) * 1024, "traffic_bytes_per_launch_mean": (2 * tf + tw) * 1024 / 6,
        "algorithmic_bytes_per_layer": 21.0e9,
        "kernel": "scail_gemm4_e0 / e1 / e3 / e4, the six per-token GEMMs of one block (tools/gemm_layer_pmc_probe.py)",
        "source": "gemm4.s", "source_blob": blob(os.path.join(ROOT, "scail_amd", "csrc", "gemm4.s")), "measured": note}
    # the dominant VAE convolution launches (tools/conv_pmc_probe.py, one shape per PMC run: lines "conv C=<C> <counter> <value>")
    conv = {}
    for line in open(src):
        m = re.match(r"conv C=(\d+)\s+(\S+)\s+(\S+)", line)
        if m:
            conv.setdefault(int(m.group(1)), {})[m.group(2)] = float(m.group(3))
    for C, v in conv.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            H, W = {96: (512, 896), 192: (256, 448), 384: (128, 224)}[C]
            alg = 2.0 * 21 * H * W * C * 2 + 27 * C * C * 2
            tr[f"conv_halo_c{C}"] = {
                "shape": {"T": 21, "H": H, "W": W, "C": C}, "fetch_kib": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"],
                "traffic_bytes": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024, "algorithmic_bytes": alg,
                "kernel": "conv_halo_kernel<0, 32, 1, false, 96, 2, 3, false> (3x3x3 causal convolution, two output frames per workgroup)",
                "source": "conv.hip", "source_blob": blob(os.path.join(ROOT, "scail_amd", "csrc", "conv.hip")), "measured": note}
    # the generated convolution kernel on the same shapes (lines "conv4 C=<C> <counter> <value>")
    conv4 = {}
    for line in open(src):
        m = re.match(r"conv4 C=(\d+)\s+(\S+)\s+(\S+)", line)
        if m:
            conv4.setdefault(int(m.group(1)), {})[m.group(2)] = float(m.group(3))
    for C, v in conv4.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            H, W = {96: (512, 896), 192: (256, 448), 384: (128, 224)}[C]
            alg = 2.0 * 21 * H * W * C * 2 + 27 * C * C * 2
            # one n tile (C = 96): the tile-continuation variant scail_conv4c_e0 of csrc/conv4u.s (option conv4_cont, default on); else scail_conv4_e0
            kern, srcf = (("scail_conv4c_e0", "conv4u.s") if C == 96 else ("scail_conv4_e0", "conv4.s"))
            tr[f"conv4_c{C}"] = {
                "shape": {"T": 21, "H": H, "W": W, "C": C}, "fetch_kib": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"],
                "traffic_bytes": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024, "algorithmic_bytes": alg,
                "kernel": kern + " (generated 3x3x3 causal convolution: persistent workgroups, 2 frames x 16 x 16 voxels x 96 channels per tile)",
                "source": srcf, "source_blob": blob(os.path.join(ROOT, "scail_amd", "csrc", srcf)), "measured": note}
    json.dump(tr, open(path, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in tr[k].items() if kk in ("traffic_bytes", "traffic_bytes_per_layer", "algorithmic_bytes", "source_blob")}
                      for k in tr if isinstance(tr[k], dict)}, indent=1))


if __name__ == "__main__":
    main()
