#!/usr/bin/env python3
"""four_tank (canonical, RK4 x4) at B = 2^20 on the lean pipelined kernel: us per step; pointed at by rocprofv3 --pmc
for SQ_INSTS_VALU (1.99e7 wave-instructions per launch = 1215 per env: 57 % of the fp64 issue rate at 56.6 us)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import scenarios as SC
from pcgym_amd import VecEnv
B = 1 << 20
p = dict(SC.scenarios()["four_tank_canonical"]["env_params"])
env = VecEnv(p, n_envs=B, seed=3); env.reset()
a = 2 * torch.rand((8, env.spec.na, B), device=env.device, dtype=torch.float64) - 1
K = int(os.environ.get("K", 40))
for i in range(10): env.step(a[i % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K): env.step(a[i % 8])
torch.cuda.synchronize(); print("us/step", (time.perf_counter() - t0) / K * 1e6, "substeps", env.spec.substeps)
