"""ctypes binding of libscail_hip.so (include/scail_hip.h, include/scail_dit.h, include/scail_vae.h).  Fails loudly when the library is
missing -- there is deliberately no fallback path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCAIL_ABLATIONS=1 selects the measurement build that also holds the timing-ablation kernels (scail_amd/build.py)
LIB_PATH = os.path.join(_HERE, "libscail_hip_abl.so" if os.environ.get("SCAIL_ABLATIONS", "0") not in ("", "0") else "libscail_hip.so")

_p = C.c_void_p
_i64 = C.c_int64
_i = C.c_int
_f = C.c_float

# name -> argtypes; must list EVERY symbol declared in include/scail_hip.h (tests/test_abi.py
# cross-checks this table against the header).
SIGNATURES = {
    "scail_gemm_bf16": [_p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _p, _i64, _p, _i64, _i64, _p],
    "scail_ln_modulate": [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _f, _p],
    "scail_layernorm_affine": [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _f, _p],
    "scail_rmsnorm_rope": [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _f, _p],
    "scail_rmsnorm_rope_scaled": [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _f, _f, _p],
    "scail_rmsnorm_rope_slabs": [_p, _i64, _p, _i64, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _f, _f, _p],
    "scail_slabs_to_rows": [_p, _i64, _i64, _p, _i64, _i64, _i64, _p],
    "scail_transpose_v": [_p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _p],
    "scail_flash_attn_bf16": [_p, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64,
                              _i64, _i64, _i64, _i64, _i64, _f, _i, _p],
    "scail_flash_attn_kernel_for": [_i64, _i64, _i64, _i64, _i64, _i, _i],
    "scail_flash_attn_rows_for": [_i64, _i64, _i64],
    "scail_flash_attn_count_restarts": [_p],
    "scail_comm_standin": [_p, _p, _i64, C.c_int32, _i64, _p],
    "scail_cross_attn2_kernel_for": [_i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64],
    "scail_gemm_kernel_for": [_i64, _i64, _i64, _i64, _i64, _i64, _i],
    "scail_conv3d_kernel_for": [_p, _i64, _i64, _i],
    "scail_cross_attn2_bf16": [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64,
                               _i64, _i64, _i64, _f, _p],
    "scail_timestep_embedding": [_p, _p, _i64, _i64, _p],
    "scail_small_linear": [_p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _p],
    "scail_adaln_table": [_p, _p, _p, _i64, _i64, _i64, _p],
    "scail_patchify": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p],
    "scail_unpatchify": [_p, _p, _i64, _i64, _i64, _i64, _p],
    "scail_cfg_euler": [_p, _p, _i64, _f, _f, _p],
    "scail_conv3d_cl": [_p, _p, _p, _p, _i64, _p, _i64, _p, _p],
    "scail_conv3d_cl_norm": [_p, _p, _p, _p, _i64, _p, _p, _p],
    "scail_conv3d_cl_resid_norm": [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p],
    "scail_rms_silu": [_p, _p, _p, _i64, _i64, _i, _p],
    "scail_softmax_rows": [_p, _i64, _i64, _i64, _f, _p],
    "scail_transpose2d": [_p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "scail_to_channels_last": [_p, _p, _p, _p, _i64, _i64, _i64, _p],
    "scail_from_channels_last": [_p, _i64, _p, _p, _p, _i64, _i64, _f, _f, _p],
    "scail_attn_small": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _p, _p, _p, _i64, _p],
    "scail_mul_bf16": [_p, _p, _p, _i64, _p],
    "scail_row_affine": [_p, _p, _p, _p, _i64, _i64, _i64, _p],
    "scail_set_option": [C.c_char_p, _i],
    "scail_release_caches": [],
    "scail_f32_to_bf16": [_p, _p, _i64, _p],
    "scail_bf16_to_f32": [_p, _p, _i64, _p],
    # include/scail_dit.h (structs are passed by pointer; scail_amd/cstep.py builds them)
    "scail_dit_create": [_p, _p, _p],
    "scail_dit_destroy": [_p],
    "scail_dit_workspace_bytes": [_p, _i64, _i64, _i64, _i64],
    "scail_dit_step": [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, C.c_uint32, _p, _i64, _p],
    "scail_dit_profile": [_p, _i],
    "scail_dit_profile_read": [_p, _i, _p, _p],
    "scail_dit_block_workspace_bytes": [_p, _i64, _i64],
    "scail_dit_block": [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _p, _i64, _p],
    "scail_dit_sp_workspace_bytes": [_p, C.c_int32, C.c_int32, _i64, _i64, _i64, _i64],
    "scail_dit_step_sp": [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _p, C.c_uint32, _p, _i64, _p],
    "scail_dit_block_sp_workspace_bytes": [_p, C.c_int32, C.c_int32, _i64, _i64],
    "scail_dit_block_sp": [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _i64, _p],
    "scail_dit_sample_workspace_bytes": [_p, _i64, _i64, _i64],
    "scail_dit_sample": [_p, _p, _p, _p, _i64, _f, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _i64, _p],
    # include/scail_vae.h (scail_amd/cvae.py builds the structs)
    "scail_vae_create": [_p, _p],
    "scail_vae_destroy": [_p],
    "scail_vae_set_trace": [_p, _p, _p],
    "scail_vae_workspace_bytes": [_p, _i64, _i64, _i64],
    "scail_vae_encode": [_p, _p, _p, _i64, _i64, _i64, _p, _i64, _p],
    "scail_vae_decode": [_p, _p, _p, _i64, _i64, _i64, _p, _i64, _p],
}
# return types other than the int status
RESTYPES = {"scail_dit_destroy": None, "scail_vae_destroy": None, "scail_vae_workspace_bytes": _i64, "scail_dit_workspace_bytes": _i64, "scail_dit_sample_workspace_bytes": _i64,
            "scail_dit_block_workspace_bytes": _i64, "scail_dit_sp_workspace_bytes": _i64, "scail_dit_block_sp_workspace_bytes": _i64}

# include/scail_hip_ablation.h: only libscail_hip_abl.so (SCAIL_ABLATIONS=1) exports these
ABLATION_SIGNATURES = {
    "scail_tune_set": [C.c_char_p, _i],
    "scail_debug_cycles": [C.c_void_p, _i],
}
ABLATIONS = LIB_PATH.endswith("_abl.so")

EPI_BIAS, EPI_GELU_TANH, EPI_GELU_ERF, EPI_RESID = 0, 1, 2, 3
ACT_NONE, ACT_SILU, ACT_GELU_TANH = 0, 1, 2
ABI_VERSION = 5          # 5 = scail_rmsnorm_rope_slabs takes a slab row stride; the sequence-parallel exchange is ONE collective per direction
                         # (send / recv layouts of scail_dit.h), exchange-wait / restart categories of scail_dit_profile_read;
                         # 4 = `flags` argument of scail_dit_step / scail_dit_step_sp (SCAIL_DIT_CFG_PAIR), options "attn4_rows" / "attn4_xcd";
                         # include/scail_hip.h scail_abi_version: 2 = negative SCAIL_ATTN_Q_PRESCALED sentinel + the SP executor entry points;
                         # 3 = scail_vae_set_trace, options "row_wave" / "conv_direct", scail_conv3d_kernel_for = 4 for the kt = 1 / narrow / fused-norm shapes

_lib = None


class ScailHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built (``python -m scail_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScailHipError(
            f"{LIB_PATH} not found: build it with `python -m scail_amd.build` (hipcc, gfx950). "
            "scail_amd has no CPU/torch fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.scail_last_error.restype = C.c_char_p
    lib.scail_last_error.argtypes = []
    lib.scail_abi_version.restype = C.c_int
    lib.scail_abi_version.argtypes = []
    if lib.scail_abi_version() != ABI_VERSION:
        raise ScailHipError(f"ABI mismatch: library {lib.scail_abi_version()} vs binding {ABI_VERSION}")
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is missing: loud by design
        fn.argtypes = args
        fn.restype = RESTYPES.get(name, C.c_int)
    if ABLATIONS:
        for name, args in ABLATION_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        for env, knob in (("SCAIL_ATTN_VARIANT", b"attn_variant"), ("SCAIL_GEMM_TILE", b"gemm_tile")):   # A/B overrides for measurement runs
            if os.environ.get(env):
                lib.scail_tune_set(knob, int(os.environ[env]))
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise ScailHipError(f"{name} failed ({rc}): {lib.scail_last_error().decode()}")


def set_option(name: str, value: int) -> None:
    """Runtime option of the product library (include/scail_hip.h scail_set_option: "attn4", "attn4_thr")."""
    call("scail_set_option", name.encode(), int(value))


def tune_set(knob: str, value: int) -> None:
    """Schedule A/B knobs and timing ablations: measurement build only (SCAIL_ABLATIONS=1, include/scail_hip_ablation.h)."""
    if not ABLATIONS:
        raise ScailHipError(f"tune_set({knob!r}): kernel variants exist only in the measurement build; run with SCAIL_ABLATIONS=1 "
                            "after `SCAIL_ABLATIONS=1 python -m scail_amd.build`")
    call("scail_tune_set", knob.encode(), int(value))
