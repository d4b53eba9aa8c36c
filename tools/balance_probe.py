#!/usr/bin/env python3
"""How much of the ME/DOPRI5 step time is lane divergence (different accepted-step counts inside a wave)?
Same envs and actions, once in random order and once with the ACTION COLUMNS sorted by a stiffness proxy
(so every wave holds envs of similar cost).  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import scenarios as SC
from pcgym_amd import VecEnv

def run(name, B, sort):
    p = dict(SC.scenarios()[name]["env_params"]); p.update(integrator="dopri5", rtol=1e-8, atol=1e-8)
    env = VecEnv(p, n_envs=B, seed=3); env.reset()
    gen = torch.Generator(device=env.device).manual_seed(7)
    acts = [2 * torch.rand((env.spec.na, B), generator=gen, device=env.device, dtype=torch.float64) - 1 for _ in range(8)]
    if sort:
        lo = torch.tensor(env.spec.a_low, device=env.device)[:, None]; hi = torch.tensor(env.spec.a_high, device=env.device)[:, None]
        out = []
        for a in acts:
            phys = (a + 1) * (hi - lo) / 2 + lo
            key = torch.maximum(phys[0], phys[1])   # L/Vl vs G/Vg with Vl = Vg
            out.append(a[:, torch.argsort(key)].contiguous())
        acts = out
    for i in range(3): env.step(acts[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for i in range(K):
        env.step(acts[i % 8])
        if env.t == env.N - 1: env.reset()
    torch.cuda.synchronize(); w = time.perf_counter() - t0
    ns = env.nsteps.double()
    print(f"{name:34s} sorted={sort!s:5s} {B*K/w:.3e} env-steps/s  {w/K*1e3:.3f} ms/step  acc {ns[0].mean():.1f} max {ns[0].max():.0f}")

for name in ("me_canonical", "me_reactive"):
    for sort in (False, True):
        run(name, 262144, sort)
