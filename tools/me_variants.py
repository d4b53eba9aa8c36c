#!/usr/bin/env python3
"""ME / DOPRI5 (BASELINE configs[2]) under the launch variants: PCG_VARIANT=0 auto, 1 classic, 2 persistent streaming."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import scenarios as SC
from pcgym_amd import VecEnv

for name in ("me_canonical", "me_reactive"):
    for variant in (0, 1, 2):
        B = 262144
        p = dict(SC.scenarios()[name]["env_params"]); p.update(integrator="dopri5", rtol=1e-8, atol=1e-8)
        env = VecEnv(p, n_envs=B, seed=3, variant=variant); env.reset()
        gen = torch.Generator(device=env.device).manual_seed(7)
        acts = [2 * torch.rand((env.spec.na, B), generator=gen, device=env.device, dtype=torch.float64) - 1 for _ in range(8)]
        for i in range(3): env.step(acts[i])
        torch.cuda.synchronize(); t0 = time.perf_counter(); K = 20
        for i in range(K):
            env.step(acts[i % 8])
            if env.t == env.N - 1: env.reset()
        torch.cuda.synchronize(); w = time.perf_counter() - t0
        print(f"{name:14s} variant {variant}: {B*K/w:.3e} env-steps/s  {w/K*1e3:.3f} ms/step")
