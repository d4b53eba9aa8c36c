import sys, os, copy, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import numpy as np, torch
import scenarios as SC
from pcgym_amd import VecEnv
name, B, ms = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
p = copy.deepcopy(SC.scenarios()[name]["env_params"]); p["integrator"] = "dopri5"; p["max_steps"] = ms
e = VecEnv(p, n_envs=B, seed=3); e.reset()
a = torch.tensor(np.random.default_rng(1).uniform(-1, 1, (e.spec.na, B)), device="cuda")
t0 = time.time(); e.step(a); torch.cuda.synchronize()
print(name, B, "ok %.3fs" % (time.time() - t0), "status counts", np.bincount(e.status.cpu().numpy(), minlength=4), "attempts max", int(e.nsteps.sum(0).max()), flush=True)
