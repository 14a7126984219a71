import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the CPU emulator of the generated kernels checks the hardware's minimum issue distances along every executed path (tools/asm_emu.py
# Emu.check_hazards; round 6: a rule the in-order emulator cannot see in the numbers cost a restart flag on the GPU)
os.environ.setdefault("SCAIL_EMU_HAZARDS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "variant: parametrisation that selects a schedule A/B variant of the MEASUREMENT build "
                                       "(SCAIL_ABLATIONS=1); deselected on the product build")


# Fixtures whose parameters select kernel variants that exist only in the measurement build (libscail_hip_abl.so): fixture name ->
# the parameter value that is the product dispatch.  Every other value of these fixtures is a variant.
VARIANT_FIXTURES = {"gemm_tile": 0, "attn_variant": 8 | (2 << 12), "conv_halo": 4}


def pytest_collection_modifyitems(config, items):
    """On the product build, variant-only parametrisations are DESELECTED (reported as such) rather than collected and skipped one by
    one: `pytest -m gpu` then shows what the shipped library can run; `SCAIL_ABLATIONS=1 pytest -m gpu` runs the variants too."""
    ablations = os.environ.get("SCAIL_ABLATIONS", "0") not in ("", "0")
    keep, drop = [], []
    for it in items:
        params = getattr(getattr(it, "callspec", None), "params", {})
        is_variant = any(name in params and params[name] != default for name, default in VARIANT_FIXTURES.items())
        if is_variant:
            it.add_marker(pytest.mark.variant)
        (drop if (is_variant and not ablations) else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def attention_plan_rows(cus: int, pairs: int, Lq: int) -> int:
    """Independent statement of the launch plan of csrc/attn.hip attn4_plan (include/scail_hip.h scail_flash_attn_rows_for) for a device
    with ``cus`` compute units: 256 / 192 = one launch of that tile height, 448 = whole rounds of 256-row workgroups + 192-row workgroups
    for the remaining rows.  Tests derive their expectation from the device's CU count with this instead of a constant that only holds
    at 256 CUs."""
    import math
    n4, n3 = -(-Lq // 256), -(-Lq // 192)
    W4, W3 = pairs * n4, pairs * n3
    rounds = lambda w: float(-(-w // cus))
    best, rows = rounds(W4), 256
    if rounds(W3) * 0.79 < best * 0.985:
        best, rows = rounds(W3) * 0.79, 192
    kk = W4 // cus
    for k in range(kk, max(0, kk - 5), -1):
        if k < 1:
            break
        amax = k * cus
        p, c = amax // n4, (amax % n4) // 3
        if p >= pairs:
            continue
        a, i3 = p * n4 + 3 * c, p * n3 + 4 * c
        b = W3 - i3
        if a <= 0 or b <= 0:
            continue
        cost = rounds(a) + rounds(b) * 0.79 + 0.02
        if cost < best * 0.985:
            best, rows = cost, 448
    return rows
