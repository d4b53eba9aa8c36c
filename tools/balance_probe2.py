#!/usr/bin/env python3
"""Wave efficiency of DOPRI5 on ME: mean(attempted steps) / mean(per-wave max) for several orderings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import scenarios as SC
from pcgym_amd import VecEnv

for name in ("me_canonical", "me_reactive"):
    B = 262144
    p = dict(SC.scenarios()[name]["env_params"]); p.update(integrator="dopri5", rtol=1e-8, atol=1e-8)
    env = VecEnv(p, n_envs=B, seed=3); env.reset()
    gen = torch.Generator(device=env.device).manual_seed(7)
    lo = torch.tensor(env.spec.a_low, device=env.device)[:, None]; hi = torch.tensor(env.spec.a_high, device=env.device)[:, None]
    for it in range(4):
        a = 2 * torch.rand((env.spec.na, B), generator=gen, device=env.device, dtype=torch.float64) - 1
        env.step(a)
    n = (env.nsteps[0] + env.nsteps[1]).double()
    phys = (a + 1) * (hi - lo) / 2 + lo
    L, G = phys[0], phys[1]
    def eff(order):
        m = n[order].reshape(-1, 64)
        return (m.mean() / m.max(dim=1).values.mean()).item()
    ident = torch.arange(B, device=env.device)
    print(name, "attempted mean %.1f max %.0f" % (n.mean().item(), n.max().item()))
    print("  random order      eff %.3f" % eff(ident))
    for label, key in (("max(L,G)", torch.maximum(L, G)), ("L+G", L + G), ("G", G), ("L", L), ("2L+G", 2 * L + G), ("L+2G", L + 2 * G), ("true nsteps", n)):
        c = torch.corrcoef(torch.stack([key, n]))[0, 1].item()
        print("  sorted by %-12s eff %.3f  corr %.3f" % (label, eff(torch.argsort(key)), c))
