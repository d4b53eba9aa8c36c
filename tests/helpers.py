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
