"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads without a GPU and exports
every symbol include/*.h declares; the ctypes table covers the same set; the product path
refuses to run without a GPU instead of silently falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ["scail_hip.h", "scail_dit.h", "scail_vae.h"]


def _header_source():
    src = "\n".join(open(os.path.join(ROOT, "include", h)).read() for h in HEADERS)
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _header_symbols():
    return sorted(set(re.findall(r"\b(scail_[a-z0-9_]+)\s*\(", _header_source())))


@pytest.fixture(scope="module")
def libpath():
    from scail_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    syms = _header_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/scail_hip.h but not exported"


def test_ctypes_table_matches_header(libpath):
    from scail_amd import lib as L
    declared = set(_header_symbols()) - {"scail_last_error", "scail_abi_version"}
    assert set(L.SIGNATURES) == declared
    src = _header_source()
    for name, args in L.SIGNATURES.items():
        m = re.search(r"(?:int|void|int64_t)\s+" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]) == len(args), name
    lib = L.load()
    assert lib.scail_abi_version() == L.ABI_VERSION


def test_product_library_has_no_measurement_aids(libpath):
    """Schedule A/B knobs, timing ablations, cycle probes and the generated GEMM experiments live in the measurement build only
    (include/scail_hip_ablation.h -> libscail_hip_abl.so); the shipped library neither exports nor embeds them."""
    from scail_amd import lib as L
    lib = ctypes.CDLL(libpath)
    abl = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "scail_hip_ablation.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(scail_[a-z0-9_]+)\s*\(", abl)))
    assert set(names) == set(L.ABLATION_SIGNATURES)
    for n in names:
        assert not hasattr(lib, n), f"{n} must not be exported by the product library"
    blob = open(libpath, "rb").read()
    for kernel in (b"scail_gemm4_e0_reg", b"scail_gemm8_e0", b"scail_attn4_rd3"):
        assert kernel not in blob, kernel
    assert b"scail_attn4" in blob and b"scail_gemm4_e3" in blob       # the shipped generated kernels are embedded
    assert not L.ABLATIONS
    with pytest.raises(L.ScailHipError, match="measurement build"):
        L.tune_set("attn_variant", 2)


def test_argument_validation_without_gpu(libpath):
    """Pure host-side checks of the ABI fire before any launch, so they are testable on CPU."""
    from scail_amd import lib as L
    L.load()
    with pytest.raises(L.ScailHipError, match="multiple of 64"):
        L.call("scail_gemm_bf16", None, 72, None, None, None, 16, 8, 16, 72, 0, None, 0, None, 0, 0, None)
    with pytest.raises(L.ScailHipError, match="M must be <= 8"):
        L.call("scail_small_linear", None, None, None, None, 9, 8, 8, 0, 0, None)
    with pytest.raises(L.ScailHipError, match="null argument"):
        L.call("scail_dit_create", None, None, None)
    assert L.load().scail_dit_workspace_bytes(None, 2, 4, 8, 8) == -1
    with pytest.raises(L.ScailHipError, match="unknown epilogue"):
        L.call("scail_gemm_bf16", None, 64, None, None, None, 16, 8, 16, 64, 99, None, 0, None, 0, 0, None)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_silent_cpu_fallback():
    from scail_amd import lib as L, ops
    x = torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises(L.ScailHipError, match="GPU"):
        ops.gemm(x, x)
    from scail_amd.dit import DiffusionTransformer
    net = DiffusionTransformer(transformer_args=dict(model_parallel_size=1), hidden_size=128, num_layers=1,
                               num_attention_heads=1, text_dim=64, time_embed_dim=128, time_freq_dim=256,
                               inner_hidden_size=256, share_adaln=True, use_i2v_clip=True, device="cpu")
    with pytest.raises(L.ScailHipError, match="GPU"):
        net(torch.zeros(2, 1, 16, 4, 4), timesteps=torch.zeros(2), context=torch.zeros(2, 4, 64),
            concat_images=torch.zeros(1), ref_concat=torch.zeros(1, 1, 16, 4, 4),
            concat_smpl_render=torch.zeros(1, 1, 16, 2, 2), image_clip_features=torch.zeros(1, 3, 1280))


def test_state_dict_keys_match_reference_layout():
    """Key names/shapes equal the reference's (oracle.state_dict_spec is pinned to the real reference by
    a strict load in oracle/gen_golden.py)."""
    from oracle import scail_oracle as O
    from scail_amd.dit import DiffusionTransformer
    cfg = O.DiTConfig(**O.TINY)
    net = DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=13, latent_width=32,
                               latent_height=32, hidden_size=256, text_dim=64, num_layers=2, num_attention_heads=2,
                               time_freq_dim=256, time_embed_dim=256, share_adaln=True, inner_hidden_size=512,
                               use_i2v_clip=True, device="cpu")
    spec = O.state_dict_spec(cfg)
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == {k: tuple(v) for k, v in spec.items()}
    with pytest.raises(NotImplementedError, match="SwiGLU"):
        DiffusionTransformer(use_SwiGLU=True, share_adaln=True, use_i2v_clip=True, device="cpu", num_layers=1)


def test_reference_yaml_targets_resolve():
    from scail_amd import config, sampler
    assert config.get_obj_from_str("sgm.modules.diffusionmodules.sampling.RFSampler") is sampler.RFSampler
    s = config.instantiate_from_config({"target": "sgm.modules.diffusionmodules.sampling.RFSampler", "params": dict(
        hunyuan_schedule=True, shift_scale=5, num_steps=50,
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})})
    sig = s.sigmas()
    assert sig.shape == (51,) and sig[0] == 1 and sig[-1] == 0 and s.guider.scale == 4


def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    """<load>/latest + <iter>/mp_rank_00_model_states.pt with 'module' keys model.diffusion_model.* (model_io.py)."""
    from scail_amd import checkpoint
    from scail_amd.cli import TINY
    from scail_amd.engine import SATVideoDiffusionEngine
    mc = dict(TINY["model"]); mc["build_first_stage"] = False
    a = SATVideoDiffusionEngine(mc, device="cpu")
    name = checkpoint.save_checkpoint(a, str(tmp_path), 1000)
    assert name.endswith("1000/mp_rank_00_model_states.pt")
    keys = list(torch.load(name)["module"].keys())
    assert all(k.startswith("model.diffusion_model.") for k in keys) and len(keys) == 73
    mc2 = dict(mc); mc2["network_config"] = dict(mc["network_config"]); mc2["network_config"]["params"] = dict(mc["network_config"]["params"], init_seed=99)
    b = SATVideoDiffusionEngine(mc2, device="cpu")
    k0 = "model.diffusion_model.transformer.layers.0.attention.dense.weight"
    assert not torch.equal(a.state_dict()[k0], b.state_dict()[k0])
    it, missing, unexpected = checkpoint.load_checkpoint(b, str(tmp_path))
    assert it == 1000 and not missing and not unexpected
    assert all(torch.equal(a.state_dict()[k], b.state_dict()[k]) for k in keys)
    with pytest.raises(ValueError, match="metadata"):
        checkpoint.load_checkpoint(b, str(tmp_path / "nope"))


def test_more_argument_validation_without_gpu(libpath):
    """Every entry point validates shapes / strides / alignment on the host before touching the device."""
    import ctypes as C
    from scail_amd import lib as L
    L.load()
    A = 0x1000       # a fake, suitably aligned device address: validation fails before it is ever dereferenced
    cases = [
        ("keep 16-byte", "scail_flash_attn_bf16", (A, 0, 132, A, 0, 0, 128, A, 0, 0, A, 0, 128, 1, 1, 8, 8, 1, 0.1, 0, None)),
        ("at least one key", "scail_flash_attn_bf16", (A, 0, 128, A, 0, 0, 128, A, 0, 0, A, 0, 128, 1, 1, 8, 0, 1, 0.1, 0, None)),
        ("vt batch stride", "scail_flash_attn_bf16", (A, 0, 128, A, 0, 0, 128, A, 0, 8, A, 0, 128, 2, 1, 8, 8, 1, 0.1, 0, None)),
        ("multiple of 8 and <= 6144", "scail_ln_modulate", (A, 8200, A, 8200, A, A, 8200, 1, 1, 1, 0, 8200, 1e-6, None)),
        ("16-byte aligned", "scail_layernorm_affine", (A + 2, 64, A, 64, A, A, 1, 64, 1e-6, None)),
        ("cos and sin tables go together", "scail_rmsnorm_rope", (A, 128, A, 128, A, A, None, 1, 1, 128, 128, 1e-6, None)),
        ("head_dim must be a multiple of 8 dividing D", "scail_rmsnorm_rope", (A, 128, A, 128, A, None, None, 1, 1, 128, 48, 1e-6, None)),
        ("16-byte alignment", "scail_transpose_v", (A, 12, 0, A, 1, 1, 128, 8, None)),
        ("K must be a multiple of 8", "scail_small_linear", (A, A, None, A, 2, 8, 12, 0, 0, None)),
        ("multiples of 4", "scail_patchify", (A, A, A, A, 2, 1, 1, 1, 6, 8, 128, None)),
        ("kpad must be", "scail_patchify", (A, A, A, A, 2, 1, 1, 1, 8, 8, 72, None)),
        ("cond batch must be 1 or n_batch", "scail_patchify", (A, A, A, A, 4, 2, 1, 1, 8, 8, 128, None)),
        ("C must be a multiple of 8", "scail_rms_silu", (A, A, A, 4, 20, 1, None)),
        ("need 1 <= n <= 32768", "scail_softmax_rows", (A, 40000, 1, 40000, 1.0, None)),
        ("need 1 <= n <= 32768", "scail_softmax_rows", (A, 8, 1, 15, 1.0, None)),          # ld does not cover ceil8(n)
        ("RESID epilogue needs resid", "scail_gemm_bf16", (A, 64, A, None, A, 16, 8, 16, 64, 3, None, 0, None, 0, 0, None)),
        ("gate needs rows_per_batch", "scail_gemm_bf16", (A, 64, A, None, A, 16, 8, 16, 64, 3, A, 16, A, 16, 0, None)),
        ("pointer alignment", "scail_gemm_bf16", (A + 4, 64, A, None, A, 16, 8, 16, 64, 0, None, 0, None, 0, 0, None)),
        ("each key set needs at least one key", "scail_cross_attn2_bf16", (A, 0, 128, A, 0, 128, A, 0, 0, A, 0, 128, A, 0, 8, A, 0, 128, 1, 1, 8, 0.1, None)),
        ("cross_attn2: strides", "scail_cross_attn2_bf16", (A, 0, 132, A, 0, 128, A, 0, 8, A, 0, 128, A, 0, 8, A, 0, 128, 1, 1, 8, 0.1, None)),
        ("unknown option", "scail_set_option", (b"no_such_option", 1)),
        ("attn4_thr must be in", "scail_set_option", (b"attn4_thr", 99)),
    ]
    for needle, fn, args in cases:
        with pytest.raises(L.ScailHipError, match=needle):
            L.call(fn, *args)
    geom = (C.c_int32 * 21)(4, 8, 8, 12, 4, 8, 8, 3, 3, 3, 1, 1, 1, 2, 1, 1, 0, 1, 0, 16, 384)
    with pytest.raises(L.ScailHipError, match="Cin must be a multiple of 8"):
        L.call("scail_conv3d_cl", A, A, None, A, 16, None, 0, C.cast(geom, C.c_void_p), None)
    # empty problems are accepted and do nothing (no launch, so no device needed)
    L.call("scail_gemm_bf16", A, 64, A, None, A, 16, 0, 16, 64, 0, None, 0, None, 0, 0, None)
    L.call("scail_flash_attn_bf16", A, 0, 128, A, 0, 0, 128, A, 0, 0, A, 0, 128, 1, 1, 0, 8, 1, 0.1, 0, None)
    L.call("scail_rms_silu", A, A, A, 0, 32, 1, None)


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under scail_amd/ (nor bench.py outside its cpu_baseline leg) may import it."""
    import ast
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1]
    for f in sorted((root / "scail_amd").glob("*.py")):
        tree = ast.parse(f.read_text())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n.split(".")[0] == "oracle" for n in names), f"{f.name} imports the oracle"
    for f in sorted((root / "scail_amd" / "csrc").iterdir()):
        bad = [ln for ln in f.read_text(errors="ignore").splitlines()
               if "oracle" in ln and ("#include" in ln or "dlopen" in ln)]
        assert not bad, f"{f.name} links the oracle: {bad}"
    src = (root / "bench.py").read_text()
    tree = ast.parse(src)
    for node in tree.body:                      # module level: no oracle import; only inside the cpu_baseline function
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            mod = node.module if isinstance(node, ast.ImportFrom) else ",".join(a.name for a in node.names)
            assert "oracle" not in (mod or "")
    users = [n.name for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)
             and any(isinstance(m, (ast.Import, ast.ImportFrom)) and "oracle" in ((getattr(m, "module", None) or "") + ",".join(a.name for a in m.names))
                     for m in ast.walk(n))]
    assert all("cpu" in u for u in users), users
