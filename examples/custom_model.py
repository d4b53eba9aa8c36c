#!/usr/bin/env python3
"""A user model on the HIP path: pc-gym's ``custom_model`` hook (pcgym.py:150-153) with a right-hand side that is not in
the registry -- a Monod chemostat with substrate inhibition and a disturbed feed concentration -- written as C
expressions and compiled into the step kernels when the env is created (hipRTC, cached on disk).

  * 65,536 envs, proportional control of the biomass set-point through the dilution rate, a non-affine constraint
    (also an expression) recorded per step, the feed concentration as a scheduled disturbance;
  * the same plant made numerically stiff (fast dilution dynamics): the explicit pair against the L-stable
    Rosenbrock integrator (`integrator: 'rodas3'`).

Needs an MI355X (there is no CPU path):  python examples/custom_model.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcgym_amd import make_vec_env  # noqa: E402

N = 60
# The model can also be handed over as a plain Python object in the reference's protocol (__call__(x, u) + info()): its
# arithmetic is then traced into the same kind of expressions automatically (pcgym_amd.config.trace_callable).
CHEMOSTAT = {
    "states": ["X", "S"], "inputs": ["D"], "disturbances": ["Sf"],
    "parameters": {"mumax": 0.53, "Ks": 0.12, "Ki": 22.0, "Y": 0.4, "Sf": 4.0},
    "aux": {"mu": "mumax*S/(Ks + S + S*S/Ki)"},          # named sub-expressions, evaluated first
    "rhs": ["(mu - D)*X", "D*(Sf - S) - mu*X/Y"],        # one expression per state
}
env_params = {
    "custom_model": CHEMOSTAT, "N": N, "tsim": 30.0,
    "x0": np.array([1.2, 0.6, 1.4]),                      # [X, S | SP slot]
    "SP": {"X": [1.4] * (N // 2) + [1.1] * (N - N // 2)}, "r_scale": {"X": 10.0},
    "a_space": {"low": np.array([0.0]), "high": np.array([0.45])},
    "o_space": {"low": np.array([0.0, 0.0, 0.0]), "high": np.array([3.0, 6.0, 3.0])},
    "disturbances": {"Sf": 4.0 + 0.8 * np.sin(np.arange(N) / 4.0)},
    "disturbance_bounds": {"low": np.array([2.0]), "high": np.array([6.0])},
    "constraints": {"expr": ["S*X - 0.9"]},               # g(x, u) <= 0, any expression over the same names
    "r_penalty": False, "done_on_cons_vio": False,
    "reference_compat": False,   # constraints see physical values (the reference hands them re-scaled ones, quirk Q3)
    "uncertainty_percentages": {"x0": [0.2, 0.3]}, "distribution": "uniform",   # per-env initial states
    "normalise_a": True, "normalise_o": True,
}


def episode(env, kp=4.0):
    obs, _ = env.reset()
    ret = torch.zeros(env.B, dtype=torch.float64, device=env.device)
    viol = torch.zeros(env.B, dtype=torch.float64, device=env.device)
    for _ in range(N - 1):
        a = torch.clamp(kp * (obs[:, 0] - obs[:, 2]), -1, 1).reshape(1, -1)   # more dilution when X is above its SP
        obs, r, done, _, info = env.step(a)
        ret += r
        viol += info["viol"]
    torch.cuda.synchronize()
    return ret, viol


def main():
    B = 65536
    t0 = time.perf_counter()
    env = make_vec_env(env_params, n_envs=B, seed=0)
    print("plan created in %.2f s (hipRTC the first time, a disk-cache hit afterwards)" % (time.perf_counter() - t0))
    ret, viol = episode(env)
    t0 = time.perf_counter()
    ret, viol = episode(env)
    dt = time.perf_counter() - t0
    print("%d envs x %d steps in %.1f ms (%.2e env-steps/s incl. the torch policy): return %.3f +- %.3f, "
          "steps with S*X > 0.9 per episode %.2f" % (B, N - 1, dt * 1e3, B * (N - 1) / dt, ret.mean(), ret.std(), viol.mean()))
    x_ref = env.x.clone()
    env.close()
    # the same plant with 1000x faster dilution dynamics (a stiff user model): explicit pair vs Rosenbrock
    for integ in ("dopri5", "rodas3"):
        p = dict(env_params, integrator=integ, rtol=1e-6, atol=1e-8, max_steps=200000)
        p["custom_model"] = dict(CHEMOSTAT, rhs=["(mu - D)*X", "1000.0*(D*(Sf - S) - mu*X/Y)"])
        e = make_vec_env(p, n_envs=4096, seed=0)
        episode(e)
        t0 = time.perf_counter()
        ret, _ = episode(e)
        print("stiff variant, %-6s: %.1f ms per episode of 4096 envs, %.0f integrator steps per env step, failed envs %d"
              % (integ, (time.perf_counter() - t0) * 1e3, e.nsteps.sum(dim=0).double().mean(), int((e.status != 0).sum())))
        e.close()
    del x_ref


if __name__ == "__main__":
    main()
