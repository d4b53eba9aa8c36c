#!/usr/bin/env python3
"""Six DOPRI5 steps of a multistage-extraction config at B = 2^18: the workload rocprofv3 --pmc is pointed at to
count VALU instructions per launch ("fp64-VALU-bound kernels").  usage: me_pmc_probe.py <scenario>"""
import os, sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import scenarios as SC
from pcgym_amd import VecEnv
name = sys.argv[1]
B = 262144
p = dict(SC.scenarios()[name]["env_params"]); p.update(integrator="dopri5", rtol=1e-8, atol=1e-8)
env = VecEnv(p, n_envs=B, seed=3); env.reset()
gen = torch.Generator(device=env.device).manual_seed(7)
for i in range(6):
    a = 2 * torch.rand((env.spec.na, B), generator=gen, device=env.device, dtype=torch.float64) - 1
    env.step(a)
torch.cuda.synchronize()
ns = env.nsteps.double()
print("attempted mean", (ns[0] + ns[1]).mean().item())
