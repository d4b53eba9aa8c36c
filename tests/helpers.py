"""Shared helpers for the parity tests."""
import copy
import os

import numpy as np

import scenarios as SC  # tests/golden/scenarios.py
from pcgym_amd.config import EnvSpec

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def rel_err(a, b, floor=1e-300):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)) if a.size else 0.0


def close(a, b, rtol, atol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def scenario_spec(name, **overrides):
    sc = SC.scenarios()[name]
    p = copy.deepcopy(sc["env_params"])
    p.update(overrides)
    return EnvSpec(p), sc


# integrator settings that make the time-discretisation error negligible (<1e-10)
# against the LSODA(1e-12) recordings, so the epilogue parity is tested tightly
TIGHT = {
    "cstr": dict(integrator="rk4", substeps=64),
    "four_tank": dict(integrator="rk4", substeps=128),
    "multistage_extraction": dict(integrator="dopri5", rtol=1e-12, atol=1e-14),
    "multistage_extraction_reactive": dict(integrator="dopri5", rtol=1e-12, atol=1e-14),
    "crystallization": dict(integrator="rk4", substeps=512),
    None: dict(integrator="rk4", substeps=64),
    "complex_cstr": dict(integrator="dopri5", rtol=1e-12, atol=1e-14),
    "photobioreactor": dict(integrator="dopri5", rtol=1e-12, atol=1e-14),
    "distillation_column": dict(integrator="dopri5", rtol=1e-12, atol=1e-14),
    "first_order_system": dict(integrator="rk4", substeps=64),
    "biofilm_reactor": dict(integrator="dopri5", rtol=1e-12, atol=1e-14),
    "heat_exchanger": dict(integrator="dopri5", rtol=1e-12, atol=1e-14),
}


def tight_for(env_params):
    return TIGHT[env_params.get("model")]


# ---- adaptive (DOPRI5) parity -------------------------------------------------------------------------------------
# The step-size controller is quantised (DESIGN.md "Adaptive stepping"), so the GPU and the oracle take IDENTICAL step
# sequences -- every env, checked as equality of the accepted / rejected counts, states to round-off -- whenever the
# step size is set by ACCURACY.  Measured on the GPU (tools/parity_probe.py, profiles/r2/parity_probe.txt): 100 % of
# 4096 envs for cstr, the 20-state reactive extraction model and every other registry model.
# The 10-state extraction model over its full action box is the exception: at |lambda| dt ~ 240 and rtol = 1e-8 the
# explicit pair runs at its STABILITY limit, where the embedded error estimate is round-off amplified by the marginally
# stable high-frequency modes (~1e8 x): a last-bit difference of one RHS evaluation (FMA contraction on the GPU) changes
# E by O(1) a few steps later.  There the two sides take different -- equally valid -- sequences for a few % of the
# envs, and the comparison is: every env within the integrator's own tolerance class of the other side AND of a
# 1e-12 solve, most envs still on identical counts.
STABILITY_LIMITED = ("multistage_extraction",)


def adaptive_check(model_name, x_gpu, x_orc, ns_gpu, ns_orc, tag, tol=1e-11, x_truth=None):
    """x_* (nx, B); ns_* (2, B) accepted / rejected counts."""
    xs = np.maximum(np.abs(x_orc), 1e-6 * np.max(np.abs(x_orc), axis=1, keepdims=True))
    ex = np.max(np.abs(x_gpu - x_orc) / xs, axis=0)
    same = np.all(ns_gpu == ns_orc, axis=0)
    if model_name in STABILITY_LIMITED:
        assert same.mean() >= 0.85, (tag, "identical step counts", same.mean())
        assert ex.max() <= 2e-6, (tag, ex.max())
        assert np.mean(ex <= 1e-9) >= 0.5, (tag, np.mean(ex <= 1e-9))
        if x_truth is not None:
            et = np.max(np.abs(x_gpu - x_truth) / xs, axis=0)
            eo = np.max(np.abs(x_orc - x_truth) / xs, axis=0)
            assert et.max() <= 3e-6 and et.max() <= 3 * max(eo.max(), 1e-7), (tag, "vs 1e-12 solve", et.max(), eo.max())
    else:
        assert same.all(), (tag, "identical step counts", same.mean())
        assert ex.max() <= tol, (tag, ex.max())
    return ex
