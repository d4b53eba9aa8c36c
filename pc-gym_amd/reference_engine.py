"""The reference-side binding: a third ``integration_method`` for pc-gym's own ``make_env``.

pc-gym builds an ``integration_engine(make_env, env_params)`` at every ``reset()`` (``pcgym.py:281``) and calls
``engine.casadi_step(state, uk)["xf"].full()`` or ``engine.jax_step(state, uk)`` once per ``step()``
(``pcgym.py:423-429``; the engine: ``integrator.py:19-107``).  ``hip_integration_engine`` keeps exactly that contract
and integrates through the C ABI (``pcg_integrate``, ``include/pcgym_hip.h``) -- one env, one launch.  It is the plug
a maintainer of the reference would add next to the casadi / jax engines (INTEGRATION.md section 2 lists the three-line
change in ``integrator.py`` / ``pcgym.py``); the batched façade (``make_vec_env``) is what makes the GPU worthwhile,
this class is what makes the engine a drop-in at the reference's own boundary.

There is no CPU path: without the built library or without a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .config import EnvSpec

_PLANS = {}  # plans are immutable once created: one per distinct numeric configuration, shared by every engine


class _DM:
    """what ``Fk["xf"]`` is to ``make_env.step``: an object whose ``.full()`` is the (nx, 1) array (pcgym.py:424-427)"""

    def __init__(self, xf):
        self._xf = xf

    def full(self):
        return self._xf.reshape(-1, 1)

    def __array__(self, dtype=None, copy=None):
        return self._xf if dtype is None else self._xf.astype(dtype)


class hip_integration_engine:
    """Same constructor and step methods as ``pcgym.integrator.integration_engine`` (integrator.py:19-107).

    ``make_env`` is accepted for signature parity (the reference uses it to build a second env and read its model,
    integrator.py:28); the numeric image of ``env_params`` comes from ``EnvSpec`` instead.  Integrator settings are the
    package's optional ``env_params`` keys (``integrator``, ``substeps``, ``rtol``, ``atol``, ``max_steps``)."""

    def __init__(self, make_env=None, env_params=None):
        import torch

        if env_params is None:
            raise ValueError("env_params is required")
        self._lib = _lib.load()  # raises when the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("hip_integration_engine needs a ROCm GPU; there is no CPU implementation of this path")
        self._torch = torch
        p = dict(env_params)
        if p.get("integration_method") in ("hip", None):
            p["integration_method"] = "hip"
        self.spec = spec = EnvSpec(p)
        cfg, keep = spec.to_cfg()
        # (plans are bound to the device they were created on: the device is part of the key)
        key = (torch.cuda.current_device(), spec.model.model_id, spec.integrator, spec.substeps, spec.rtol, spec.atol,
               spec.max_steps, spec.dt, spec.nu, tuple(np.asarray(spec.param_vector()).ravel().tolist()),
               spec.user_rhs_src, spec.ep_frac, spec.ep_kmax)
        ent = _PLANS.get(key)
        if ent is None:
            plan = C.c_void_p()
            _lib.check(self._lib.pcg_plan_create(C.byref(plan), C.byref(cfg)), "pcg_plan_create")
            ent = _PLANS[key] = (plan, keep)
        self._plan = ent[0]
        self.nx, self.nu = spec.nx, spec.nu
        self._x = torch.zeros((self.nx, 1), dtype=torch.float64, device="cuda")
        self._u = torch.zeros((self.nu, 1), dtype=torch.float64, device="cuda")
        self._hx = torch.zeros((self.nx + self.nu,), dtype=torch.float64).pin_memory()
        self.env = None  # the reference keeps its helper env here (integrator.py:28); nothing on this path reads it

    def _integrate(self, state, uk):
        torch = self._torch
        state = np.asarray(state, dtype=np.float64).reshape(-1)
        uk = np.asarray(uk, dtype=np.float64).reshape(-1)
        if state.shape[0] < self.nx:
            raise ValueError(f"state has {state.shape[0]} entries, the model has {self.nx} states")
        if uk.shape[0] != self.nu:
            raise ValueError(f"uk has {uk.shape[0]} entries, expected {self.nu} (inputs + model disturbance inputs)")
        h = self._hx.numpy()
        h[: self.nx] = state[: self.nx]  # make_env hands over its whole state vector [x | SP | d]: integrator.py:99
        h[self.nx:] = uk
        self._x[:, 0].copy_(self._hx[: self.nx], non_blocking=True)
        self._u[:, 0].copy_(self._hx[self.nx:], non_blocking=True)
        _lib.check(self._lib.pcg_integrate(self._plan, 1, self._x.data_ptr(), self._u.data_ptr(), None,
                                           torch.cuda.current_stream().cuda_stream), "pcg_integrate")
        xf = self._x[:, 0].cpu().numpy()
        if not np.isfinite(xf).all():  # CVODES raises on a failed integration; so does this engine
            raise RuntimeError("integration failed (step budget exhausted, step-size underflow or non-finite state)")
        return xf

    def casadi_step(self, state, uk):
        """integrator.py:90-107: integrate over [0, dt] with uk held constant; returns {"xf": DM-like}"""
        return {"xf": _DM(self._integrate(state, uk))}

    def jax_step(self, state, uk):
        """integrator.py:65-88: returns the new model state as an array of nx entries"""
        return self._integrate(state, uk)
