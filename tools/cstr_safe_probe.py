import sys, os, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden')
import bench as BN
from pcgym_amd import VecEnv
_, p, B, _, _ = BN.single_workload("cstr_safe")
env = VecEnv(p, n_envs=B, seed=1234, auto_reset=False)
env.reset()
gen = torch.Generator(device="cuda").manual_seed(1)
for i in range(40):
    a = 2*torch.rand((1,B),generator=gen,device="cuda",dtype=torch.float64)-1
    env.step(a)
    if i in (0,1,2,5,10,20,39):
        ns = env.nsteps.sum(0).double(); esc = ns>0
        print(i, "escalated %.3f"%esc.double().mean().item(), "attempts among escalated mean %.1f max %d"%(ns[esc].mean().item() if esc.any() else 0, int(ns.max())), "T max %.0f hot frac %.3f"%(env.x[1].max().item(), (env.x[1]>400).double().mean().item()))
