"""Batched rollout collector + reproducibility metrics.

Counterpart of the reference's ``policy_eval.rollout / get_rollouts``
(src/pcgym/policy_evaluation.py:71-197) and ``reproducibility_metric``
(src/pcgym/evaluation_metrics.py:182-327), for B environments at once and with
every array on the GPU in the reference's axis order:

    r (1, N, B)   r[0, 0] = r_init = 0, r[0, i+1] = reward of step i          (:86, :113-116)
    x (Nx, N, B)  x[:, 0] = reset observation, x[:, i+1] = observation after step i,
                  both de-normalised with observation_space_base                (:88-106)
    u (na, N, B)  physical (de-normalised) actions, column N-1 = the action the policy
                  proposes for the final observation                            (:101-104, 123-127)
    g (n_con, N, 1, B)  constraint rows (cons_info)                             (:118-121, 180-183)

``reps`` of the reference (independent repetitions of one env) is the env axis B here.
Closed loop: ``policy(obs (B, Nobs) tensor) -> (B, na)`` (or ``(na, B)``) tensor, one
kernel launch per step.  Open loop (``actions`` given, no constraint rows to record): the fused
``pcg_rollout_strided`` kernel writes straight into these layouts, state in registers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _torch():
    import torch

    return torch


def _denorm_obs(env, o_soa):
    """(o + 1) * (high - low) / 2 + low, the inverse map policy_eval applies (:88-90); identity when
    observations are not normalised."""
    torch = _torch()
    s = env.spec
    if not s.normalise_o:
        return o_soa
    shp = (-1,) + (1,) * (o_soa.dim() - 1)
    lo = torch.as_tensor(s.o_low, device=env.device).reshape(shp)
    hi = torch.as_tensor(s.o_high, device=env.device).reshape(shp)
    return (o_soa + 1) * (hi - lo) / 2 + lo


def _denorm_act(env, a_soa):
    torch = _torch()
    s = env.spec
    if not s.normalise_a:
        return a_soa
    shp = (-1,) + (1,) * (a_soa.dim() - 1)
    lo = torch.as_tensor(s.a_low, device=env.device).reshape(shp)
    hi = torch.as_tensor(s.a_high, device=env.device).reshape(shp)
    return (a_soa + 1) * (hi - lo) / 2 + lo


def collect_rollouts(env, policy=None, actions=None):
    """Roll all B envs of a VecEnv through one episode (N-1 steps) and return the reference-shaped dict.

    policy  : callable obs(B,Nobs) -> action (B,na)|(na,B) tensor (closed loop), or
    actions : (N, na, B) tensor of policy outputs (open loop; row N-1 is only recorded in ``u``).
    """
    torch = _torch()
    s = env.spec
    B, N, dev = env.B, s.N, env.device
    if (policy is None) == (actions is None):
        raise ValueError("give exactly one of policy / actions")
    if env.per_env_t:
        raise ValueError("collect_rollouts needs a lock-stepped VecEnv")
    f64 = torch.float64
    x = torch.zeros((s.nobs, N, B), dtype=f64, device=dev)
    u = torch.zeros((s.na, N, B), dtype=f64, device=dev)
    r = torch.zeros((1, N, B), dtype=f64, device=dev)
    g = torch.zeros((s.ncon, N, 1, B), dtype=f64, device=dev) if s.ncon else None
    obs, _ = env.reset()
    x[:, 0] = _denorm_obs(env, env.obs_soa)
    # the fused rollout records observations and rewards, not the constraint rows; per-env parameters need the
    # per-step kernel; the 20-state DOPRI5 rollout kernel is slower than stepping (tools/rollout_probe.py)
    fused_ok = (not s.ncon and not s.nunc and s.integrator != "rodas3" and s.user_rhs_src is None
                and (s.integrator == "rk4" or s.nx <= 10))
    if actions is not None:
        actions = actions.to(device=dev, dtype=f64)
        if actions.shape != (N, s.na, B):
            raise ValueError(f"actions must have shape ({N},{s.na},{B})")
        u[:] = _denorm_act(env, actions.permute(1, 0, 2))
        if fused_ok:
            # fused: the kernel writes normalised obs rows directly into x[:, 1:, :] / r[0, 1:, :]
            a = actions.contiguous()
            rc = env._lib.pcg_rollout_strided(
                env._plan, env._bufp, 0, N - 1, a.data_ptr(), s.na * B, B,
                x[:, 1:].data_ptr(), B, N * B, r[:, 1:].data_ptr(), B,
                env._episode_seed(), env._stream())
            _lib.check(rc, "pcg_rollout_strided")
            env.t += N - 1
            if s.normalise_o:
                x[:, 1:] = _denorm_obs(env, x[:, 1:])
            out = {"r": r, "x": x, "u": u}
            return out
        pol = None
    for i in range(N - 1):
        a = actions[i] if actions is not None else policy(obs)
        a = env._as_soa(a, s.na, "action")
        if actions is None:
            u[:, i] = _denorm_act(env, a)
        obs, rew, done, _, info = env.step(a)
        x[:, i + 1] = _denorm_obs(env, env.obs_soa)
        r[0, i + 1] = rew
        if g is not None:
            if i == 0:
                g[:, 0, 0] = env.g_pre
            g[:, i + 1, 0] = env.g
    if actions is None:
        u[:, N - 1] = _denorm_act(env, env._as_soa(policy(obs), s.na, "action"))
    out = {"r": r, "x": x, "u": u}
    if g is not None:
        out["g"] = g
    return out


class reproducibility_metric:
    """Same constructor / methods / dict shapes as the reference class
    (evaluation_metrics.py:182-327), computed with torch on whatever device the data lives on.
    The reductions run along the last axis (the reps / env axis)."""

    def __init__(self, dispersion: str, performance: str, scalarised_weight: float):
        if dispersion not in ("std", "mad"):
            raise ValueError("Invalid dispersion metric")
        if performance not in ("mean", "median"):
            raise ValueError("Invalid performance metric")
        self.dispersion, self.performance, self.scalarised_weight = dispersion, performance, scalarised_weight

    @staticmethod
    def _median(t):
        # np.median averages the two middle values for even counts; torch.median returns the lower one
        torch = _torch()
        return torch.quantile(t, 0.5, dim=-1)

    def _op(self, comp, t):
        return t.amax(dim=0) if comp == "g" else t  # greatest constraint row (evaluation_metrics.py:322-326)

    def _perf(self, t):
        return t.mean(dim=-1) if self.performance == "mean" else self._median(t)

    def _disp(self, t):
        torch = _torch()
        if self.dispersion == "std":
            return t.std(dim=-1, unbiased=False)  # np.std default (ddof = 0)
        med = self._median(t)
        return self._median((t - med.unsqueeze(-1)).abs())

    def _apply(self, fn, data, component):
        out = {k: {} for k in data}
        for pol, d in data.items():
            for comp in (d.keys() if component is None else [component]):
                out[pol][comp] = fn(self._op(comp, _torch().as_tensor(d[comp])))
        return out

    def policy_performance_metric(self, data, component=None):
        return self._apply(self._perf, data, component)

    def policy_dispersion_metric(self, data, component=None):
        return self._apply(self._disp, data, component)

    def scalarised_performance(self, data, component=None):
        p = self.policy_performance_metric(data, component)
        d = self.policy_dispersion_metric(data, component)
        return {k: {c: p[k][c] + self.scalarised_weight * d[k][c] for c in p[k]} for k in p}

    def evaluate(self, policy_evaluator, component=None):
        data = getattr(policy_evaluator, "data", None)
        if data is None:
            data = policy_evaluator.get_rollouts()
        return self.scalarised_performance(data, component)
