"""debug: lean kernel on an odd batch against the same envs of an even batch"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import scenarios as SC
from pcgym_amd import VecEnv
for integ in ("rk4", "cv8"):
    p = copy.deepcopy(SC.scenarios()["four_tank_canonical"]["env_params"])
    p["integrator"] = integ
    outs = []
    for B in (2048, 2047, 2046, 1023):
        env = VecEnv(p, n_envs=B, seed=3)
        env.reset()
        gen = torch.Generator(device="cuda").manual_seed(0)
        a = 2 * torch.rand((2, 2048), generator=gen, device="cuda", dtype=torch.float64) - 1
        x0 = env.x[:, :1023].clone()
        for i in range(5):
            env.step(a[:, :B].contiguous())
        outs.append((env.x[:, :1023].clone(), x0))
        env.close()
    for k in range(1, 4):
        d = (outs[0][0] - outs[k][0]).abs()
        print(integ, "vs B idx", k, "x0 equal", torch.equal(outs[0][1], outs[k][1]), "max diff", float(d.max()), "n differing envs", int((d.max(dim=0).values > 0).sum()), "first", (d.max(dim=0).values > 0).nonzero()[:5].flatten().tolist())
