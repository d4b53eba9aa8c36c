import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # kernel-instantiation coverage: the library notes every kernel it launches (include/pcgym_hip.h: pcg_coverage_names);
    # must be in the environment before libpcgym_hip.so is loaded
    os.environ.setdefault("PCG_COVERAGE", "1")
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


# ---- kernel-instantiation coverage -----------------------------------------------------------------------------------------
# After every GPU test the library is asked which kernel instantiations the test launched; the test counts as a check
# "against the oracle" if it reached for oracle/libpcg_oracle.so while it ran, "against a golden fixture" if it loaded one
# of tests/golden/*.npz, otherwise it is a self-consistency / property test ("other").  The record is written to
# gpurun_out/kernel_coverage.json at the end of the session (tools/kernel_coverage.py turns it into
# profiles/rN/kernel_coverage.txt) and gated by tests/test_zz_kernel_coverage.py, which runs last.
from _coverage_state import COVERAGE, COVERAGE_STATE  # noqa: E402  (tests/ is on sys.path: shared with the gate test)


def _coverage_drain():
    """names launched since the last call (empty when the library is not loaded yet or was built without the hook)"""
    from pcgym_amd import _lib

    lib = _lib._lib if hasattr(_lib, "_lib") else None
    if lib is None:
        return []
    import ctypes as C

    n = lib.pcg_coverage_names(None, 0, 0)
    if n <= 1:
        return []
    buf = C.create_string_buffer(int(n))
    lib.pcg_coverage_names(buf, n, 1)
    return [s for s in buf.value.decode().split("\n") if s]


def pytest_deselected(items):
    COVERAGE_STATE["gpu_deselected"] += sum(1 for it in items if "gpu" in it.keywords)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    if "gpu" not in item.keywords or not _has_gpu():
        yield
        return
    from oracle import oracle as O
    import helpers

    _coverage_drain()  # launches of fixtures / collection do not belong to this test
    c0, g0 = O.CALLS[0], helpers.GOLD_LOADS[0]
    outcome = yield
    kind = "oracle" if O.CALLS[0] > c0 else "golden" if helpers.GOLD_LOADS[0] > g0 else "other"
    names = _coverage_drain()
    COVERAGE_STATE["gpu_ran"] += 1
    if outcome.excinfo is not None:
        if outcome.excinfo[0] is pytest.skip.Exception:
            COVERAGE_STATE["gpu_skipped_in_call"] = COVERAGE_STATE.get("gpu_skipped_in_call", 0) + 1
        else:
            COVERAGE_STATE["gpu_failed"] += 1
        return  # a failing (or skipping) test checks nothing
    for n in names:
        COVERAGE.setdefault(n, {"oracle": [], "golden": [], "other": []})[kind].append(item.nodeid)


def pytest_sessionfinish(session, exitstatus):
    if not COVERAGE:
        return
    import json

    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    rec = {k: {kind: {"n": len(v), "tests": sorted(set(v))[:4]} for kind, v in d.items()} for k, d in COVERAGE.items()}
    with open(os.path.join(out, "kernel_coverage.json"), "w") as f:
        json.dump({"state": COVERAGE_STATE, "kernels": rec}, f, indent=0, sort_keys=True)
