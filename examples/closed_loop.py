#!/usr/bin/env python3
"""pc-gym's quick-start configuration (README.md:16-55 of the reference) on the HIP path, three ways:

  1. make_env       -- the reference's single-env surface (numpy in/out), a proportional controller in a Python loop
  2. make_vec_env   -- 65,536 envs with initial-state uncertainty, the same controller as a torch expression on the
                       device, one kernel launch per step
  3. collect_rollouts + reproducibility_metric -- the reference's `x (Nx, N, reps)` trajectory arrays and the
                       median / MAD summary of evaluation_metrics.py, computed on the device

Needs an MI355X (there is no CPU path):  python examples/closed_loop.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcgym_amd import collect_rollouts, make_env, make_vec_env, reproducibility_metric  # noqa: E402

N = 60
env_params = {
    "model": "cstr", "N": N, "tsim": 26,
    "SP": {"Ca": [0.85] * (N // 3) + [0.9] * (N // 3) + [0.87] * (N - 2 * (N // 3))},
    "o_space": {"low": np.array([0.7, 300.0, 0.8]), "high": np.array([1.0, 350.0, 0.9])},
    "a_space": {"low": np.array([295.0]), "high": np.array([302.0])},
    "x0": np.array([0.8, 330.0, 0.8]), "r_scale": {"Ca": 1e3}, "normalise_a": True, "normalise_o": True,
}
KP = 3.0  # normalised action = KP * (Ca - Ca_SP) in normalised observation units (more coolant heat -> less Ca)


def main():
    # 1. single env, reference surface
    env = make_env(env_params)
    obs, info = env.reset()
    ret = 0.0
    for _ in range(N - 1):
        a = np.clip(np.array([KP * (obs[0] - obs[2])]), -1, 1)
        obs, r, done, trunc, info = env.step(a)
        ret += r
    print(f"single env : return {ret:9.3f}   final Ca {env.state[0]:.4f} (set-point {env_params['SP']['Ca'][-1]})")

    # 2. batched, policy on the device
    B = 1 << 16
    p = dict(env_params, uncertainty_percentages={"x0": [0.03, 0.005]}, distribution="uniform")
    venv = make_vec_env(p, n_envs=B, seed=0)
    obs, _ = venv.reset()
    total = torch.zeros(B, dtype=torch.float64, device=venv.device)
    for _ in range(N - 1):
        a = torch.clamp(KP * (obs[:, 0] - obs[:, 2]), -1, 1).reshape(1, B)
        obs, r, done, trunc, info = venv.step(a)
        total += r
    print(f"{B} envs : return mean {total.mean().item():9.3f}  std {total.std().item():.3f}")

    # 3. the reference's trajectory arrays + reproducibility summary, on the device
    venv2 = make_vec_env(p, n_envs=4096, seed=1)
    data = collect_rollouts(venv2, policy=lambda o: torch.clamp(KP * (o[:, 0] - o[:, 2]), -1, 1).reshape(-1, 1))
    print("collect_rollouts:", {k: tuple(v.shape) for k, v in data.items() if hasattr(v, "shape")})
    metric = reproducibility_metric(dispersion="mad", performance="median", scalarised_weight=1.0)
    perf = metric.scalarised_performance({"P-controller": data}, component="r")["P-controller"]["r"]  # (1, N)
    print("median + MAD of the reward, summed over the episode:", float(perf.sum()))


if __name__ == "__main__":
    main()
