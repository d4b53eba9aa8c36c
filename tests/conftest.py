import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # run-time compiled kernels: a cache of this session only, so that a green run never validates code objects left
    # behind by an earlier build (the key covers the kernel headers since round 3; this is belt and braces)
    if "PCG_JIT_CACHE" not in os.environ:
        import tempfile

        os.environ["PCG_JIT_CACHE"] = tempfile.mkdtemp(prefix="pcg_jit_test_")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a box without a GPU: they are skipped
    # only when deselected by -m "not gpu"; if selected without a GPU they fail loudly.
    if _has_gpu():
        return
    def _no_gpu(*a, **k):
        pytest.fail("this test is marked gpu and was selected, but no GPU is visible (torch.cuda.is_available() is "
                    "False): deselect with -m 'not gpu' on a CPU box", pytrace=False)

    for item in items:
        if "gpu" in item.keywords:
            item.obj = _no_gpu  # fail loudly: a GPU test must never look green (or xfail-quiet) without a GPU
