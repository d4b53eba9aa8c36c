#!/usr/bin/env python3
"""Fused pcg_rollout against T separate pcg_step launches for the compute-bound models (B = 2^17, T = 10)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import scenarios as SC
from pcgym_amd import VecEnv
for name in ("me_canonical", "me_reactive", "cryst_adelta", "four_tank_canonical"):
    B, T = 1 << 17, 10
    p = dict(SC.scenarios()[name]["env_params"])
    env = VecEnv(p, n_envs=B, seed=3)
    gen = torch.Generator(device=env.device).manual_seed(7)
    acts = 0.2 * (2 * torch.rand((T, env.spec.na, B), generator=gen, device=env.device, dtype=torch.float64) - 1) - 0.5
    for mode in ("step", "rollout"):
        for rep in range(2):
            env.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            if mode == "step":
                for i in range(T): env.step(acts[i])
            else:
                env.rollout(acts, collect_obs=True, collect_rew=True)
            torch.cuda.synchronize(); w = time.perf_counter() - t0
        print(f"{name:22s} {env.spec.integrator:7s} {mode:8s} {B*T/w:.3e} env-steps/s")
