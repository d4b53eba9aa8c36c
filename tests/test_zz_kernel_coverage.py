"""The gate on kernel-instantiation coverage (runs LAST in the GPU suite: the file name sorts after every other test file).

libpcgym_hip.so carries several hundred kernel instantiations (model x integrator x counter mode x launch shape);
round 5 found one of them silently wrong after two green rounds, because nothing said which instantiations the suite ever
launched against the oracle.  Now the library notes every kernel it launches (pcg_coverage_names), conftest.py attributes
the launches to tests, and this test compares the record with the kernels the library actually carries
(tools/kernel_inventory.py reads them out of the code objects):

  * every shipped kernel must have been launched by a PASSING test that checks against the oracle or a committed
    reference fixture -- or be listed, with the reason, in tests/kernel_coverage_allow.txt;
  * the allow list must not go stale: every pattern in it has to match a shipped kernel that the suite did NOT check.

Protects integrator.py:90-107 for every registry model of pcgym.py:128-148.
"""
import fnmatch
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
ALLOW = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_coverage_allow.txt")


def allow_list():
    """[(glob pattern over demangled names, reason)]"""
    out = []
    with open(ALLOW) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line.strip() or line.lstrip().startswith("#"):
                continue
            pat, _, why = line.partition("  # ")
            assert why.strip(), f"allow-list entry without a reason: {line!r}"
            out.append((pat.strip(), why.strip()))
    return out


def test_inventory_reads_the_library():
    """CPU: the code objects of the built library parse, every kernel family is there, names demangle"""
    import kernel_inventory as KI

    ks = KI.inventory()
    fams = {k["family"] for k in ks}
    assert {"pcg::step_kernel", "pcg::step_kernel_queue", "pcg::step_kernel_pipe", "pcg::rollout_kernel",
            "pcg::integrate_kernel", "pcg::rhs_kernel", "pcg::reset_kernel"} <= fams, fams
    assert len(ks) > 300
    assert all(k["vgpr"] > 0 and k["max_wg"] > 0 for k in ks)
    # every allow-list pattern names something the library carries
    for pat, _ in allow_list():
        assert any(fnmatch.fnmatchcase(k["demangled"], pat) for k in ks), f"stale allow-list pattern: {pat}"


@pytest.mark.gpu
def test_every_shipped_kernel_ran_against_the_oracle():
    import kernel_inventory as KI
    from _coverage_state import COVERAGE, COVERAGE_STATE

    if COVERAGE_STATE["gpu_deselected"] or COVERAGE_STATE["gpu_failed"]:
        pytest.skip(f"not a full green run of the GPU suite in this process ({COVERAGE_STATE}): nothing to gate")
    assert COVERAGE, "the library recorded no launches: PCG_COVERAGE was not in the environment when it was loaded"
    ks = KI.inventory()
    checked = {n for n, d in COVERAGE.items() if d["oracle"] or d["golden"]}
    launched = set(COVERAGE)
    shipped = {k["name"] for k in ks}
    assert launched - shipped <= {n for n in launched if n.startswith("jit:")}, \
        f"launched kernels that the inventory does not list: {sorted(launched - shipped)[:5]}"
    allow = allow_list()
    missing, used = [], set()
    for k in ks:
        if k["name"] in checked:
            continue
        hit = [pat for pat, _ in allow if fnmatch.fnmatchcase(k["demangled"], pat)]
        if hit:
            used.update(hit)
        else:
            missing.append(("launched, never against the oracle: " if k["name"] in launched else "never launched: ") + k["demangled"])
    assert not missing, f"{len(missing)} of {len(ks)} shipped kernels unchecked:\n" + "\n".join(missing[:60])
    stale = [pat for pat, _ in allow if pat not in used]
    assert not stale, f"allow-list entries that no unchecked kernel needs any more: {stale}"
