import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


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
