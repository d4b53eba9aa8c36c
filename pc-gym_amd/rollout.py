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


def _affine(env, low, high, active):
    """de-normalisation (v + 1) * (high - low) / 2 + low as v * half + mid: (half, mid) column tensors, or None"""
    if not active:
        return None
    torch = _torch()
    lo = torch.as_tensor(np.asarray(low, dtype=np.float64), device=env.device)
    hi = torch.as_tensor(np.asarray(high, dtype=np.float64), device=env.device)
    return (hi - lo) / 2, (hi + lo) / 2


def _denorm_(t, hm, dim):
    """in place, one pass over the data: t[..] = t[..] * half + mid with (half, mid) broadcast along `dim`"""
    if hm is None:
        return t
    torch = _torch()
    shp = [1] * t.dim()
    shp[dim] = -1
    half, mid = hm[0].reshape(shp), hm[1].reshape(shp)
    return torch.addcmul(mid, t, half, out=t)


def collect_rollouts(env, policy=None, actions=None):
    """Roll all B envs of a VecEnv through one episode (N-1 steps) and return the reference-shaped dict.

    policy  : callable obs(B,Nobs) -> action (B,na)|(na,B) tensor (closed loop), or
    actions : (N, na, B) tensor of policy outputs (open loop; row N-1 is only recorded in ``u``).

    Recording is zero-copy: each step's kernel writes its observation / reward rows straight into the trajectory
    storage (``VecEnv.bind_outputs``), and the de-normalisation to physical units (policy_evaluation.py:88-106) is one
    in-place pass over the finished arrays.  ``x`` / ``u`` come back in the reference's axis order ``(Nx, N, B)`` as
    views of step-major storage.  (Measured at B = 2^20, N = 60, profiles/r2/collector_probe.txt.)
    """
    torch = _torch()
    s = env.spec
    B, N, dev = env.B, s.N, env.device
    if (policy is None) == (actions is None):
        raise ValueError("give exactly one of policy / actions")
    if env.per_env_t:
        raise ValueError("collect_rollouts needs a lock-stepped VecEnv")
    f64 = torch.float64
    o_hm = _affine(env, s.o_low, s.o_high, s.normalise_o)
    a_hm = _affine(env, s.a_low, s.a_high, s.normalise_a)
    r = torch.empty((1, N, B), dtype=f64, device=dev)
    r[0, 0] = 0.0
    g = torch.zeros((s.ncon, N, 1, B), dtype=f64, device=dev) if s.ncon else None
    obs, _ = env.reset()
    # the fused rollout records observations and rewards, not the constraint rows; per-env parameters roll through the
    # general rollout kernel's UNC form (RK4 / explicit pair, round 4); the 20-state DOPRI5 rollout kernel is slower than stepping (tools/rollout_probe.py)
    # (user models: the run-time compiled module carries its own rollout kernel for the register-only integrators)
    fused_ok = (not s.ncon and (not s.nunc or s.integrator in ("rk4", "dopri5")) and (s.integrator not in ("rodas3", "rodas4", "rodas5", "tsit5") or (s.integrator in ("rodas4", "rodas5") and s.model.name == "multistage_extraction"))
                and (s.user_rhs_src is None or s.integrator in ("rk4", "cv8", "dopri5"))
                # a plan with user expressions runs from its run-time compiled module, which carries a rollout kernel only
                # for the register-only explicit schemes (pcg_abi.hip: jit_kernels): the Rosenbrock pairs step instead
                and not ((s.user_reward_src or s.user_cons_src) and s.integrator in ("rodas3", "rodas4", "rodas5"))
                and (s.integrator in ("rk4", "cv8") or s.nx <= 10))
    if actions is not None:
        actions = actions.to(device=dev, dtype=f64)
        if actions.shape != (N, s.na, B):
            raise ValueError(f"actions must have shape ({N},{s.na},{B})")
    if actions is not None and fused_ok:
        # fused: the kernel writes the observation rows directly into x[:, 1:, :] / r[0, 1:, :]
        x = torch.empty((s.nobs, N, B), dtype=f64, device=dev)
        x[:, 0] = env.obs_soa
        a = actions.contiguous()
        rc = env._lib.pcg_rollout_strided(
            env._plan, env._bufp, 0, N - 1, a.data_ptr(), s.na * B, B,
            x[:, 1:].data_ptr(), B, N * B, r[:, 1:].data_ptr(), B,
            env._episode_seed(), env._stream())
        _lib.check(rc, "pcg_rollout_strided")
        env.t += N - 1
        u = a if a_hm is None else _denorm_(a.clone(), a_hm, 1)  # never scale the caller's tensor in place
        return {"r": r, "x": _denorm_(x, o_hm, 0), "u": u.permute(1, 0, 2)}
    # per-step path: step-major storage, the env's kernels write into it
    xs = torch.empty((N, s.nobs, B), dtype=f64, device=dev)
    us = torch.empty((N, s.na, B), dtype=f64, device=dev)
    xs[0] = env.obs_soa
    rs = r[0]
    saved = (env.obs_soa, env.rew)
    try:
        for i in range(N - 1):
            a = actions[i] if actions is not None else policy(obs)
            a = env._as_soa(a, s.na, "action")
            us[i] = a
            env.bind_outputs(xs[i + 1], rs[i + 1])
            obs, rew, done, _, info = env.step(a)
            if g is not None:
                if i == 0:
                    g[:, 0, 0] = env.g_pre
                g[:, i + 1, 0] = env.g
        if actions is None:
            us[N - 1] = env._as_soa(policy(obs), s.na, "action")
        else:
            us[N - 1] = actions[N - 1]
    finally:
        last_o, last_r = env.obs_soa, env.rew
        env.bind_outputs(*saved)
        env.obs_soa.copy_(last_o)  # the env keeps its own storage; its latest outputs stay readable there
        env.rew.copy_(last_r)
    out = {"r": r, "x": _denorm_(xs, o_hm, 1).permute(1, 0, 2), "u": _denorm_(us, a_hm, 1).permute(1, 0, 2)}
    if g is not None:
        out["g"] = g
    return out


class reproducibility_metric:
    """Same constructor / methods / dict shapes as the reference class (evaluation_metrics.py:182-327), computed with
    torch on whatever device the data lives on.  The reductions run along the last axis (the reps / env axis).

    Pinned to the reference's own outputs (tests/golden/metrics_ref.npz, generated by importing evaluation_metrics.py).
    Quirk Q14 (`reference_compat=True`, the default): the reference's MAD subtracts the median WITHOUT keeping the reduced
    axis (evaluation_metrics.py:127-130, "currently only works for the reward component"), so `data - median` follows
    NumPy broadcasting: on (n, N, reps) data it raises unless N == reps (and then subtracts the median of another
    time index), on the constraint component (N, 1, reps) it returns an (N, N) table, 1-D data is reshaped to (n, 1)
    first.  All of that is reproduced, including the ValueError.  `reference_compat=False` gives the median absolute
    deviation about each row's own median, for every shape."""

    def __init__(self, dispersion: str, performance: str, scalarised_weight: float, reference_compat: bool = True):
        if dispersion not in ("std", "mad"):
            raise ValueError("Invalid dispersion metric")
        if performance not in ("mean", "median"):
            raise ValueError("Invalid performance metric")
        self.dispersion, self.performance, self.scalarised_weight = dispersion, performance, scalarised_weight
        self.reference_compat = bool(reference_compat)

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
        if self.dispersion == "std":
            return t.std(dim=-1, unbiased=False)  # np.std default (ddof = 0)
        if not self.reference_compat:
            return self._median((t - self._median(t).unsqueeze(-1)).abs())
        if t.dim() < 2:  # evaluation_metrics.py:112-114
            t = t.reshape(t.shape[0], 1)
        med = self._median(t)
        try:
            dev = t - med  # NumPy broadcasting of (.., reps) against (..): evaluation_metrics.py:127-130
        except RuntimeError as e:
            raise ValueError(f"operands could not be broadcast together with shapes {tuple(t.shape)} "
                             f"{tuple(med.shape)} (the reference's MAD, evaluation_metrics.py:127-130: quirk Q14; "
                             "reference_compat=False gives the deviation about each row's own median)") from e
        return self._median(dev.abs())

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
