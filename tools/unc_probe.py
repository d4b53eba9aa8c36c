"""Timing of the per-env parameter (reset-time uncertainty) and auto-reset paths on the headline workload (needs a GPU)."""
import copy, os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, os.path.join(ROOT,"tests","golden"))
import torch
import bench as BN
from tools.user_model_probe import run
B=1<<20
base=BN.workload_params()
for tag,upd,kw in [("lean rk4x1", {}, {}),
               ("+ 2 uncertain parameters (per-env p_unc)", {"uncertainty_percentages": {"x0":[0.1,0.03], "q":0.05, "UA":0.1}, "uncertainty_bounds": {"low":[90.0,4e4],"high":[110.0,6e4]}}, {}),
               ("per-env t + auto-reset", {}, {"per_env_t":True,"auto_reset":True}),
               ("+ unc + per-env t + auto-reset", {"uncertainty_percentages": {"x0":[0.1,0.03], "q":0.05, "UA":0.1}, "uncertainty_bounds": {"low":[90.0,4e4],"high":[110.0,6e4]}}, {"per_env_t":True,"auto_reset":True})]:
    p=copy.deepcopy(base); p.update(upd)
    from pcgym_amd import VecEnv
    import time
    env=VecEnv(p,n_envs=B,seed=1,track_status=False,**kw)
    acts=[torch.rand((1,B),device=env.device,dtype=torch.float64)*2-1 for _ in range(8)]
    best=1e9
    for r in range(4):
        env.reset(); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(59): env.step(acts[i%8])
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/59*1e3)
    print("%-50s %.1f us per step" % (tag,best))
    env.close()
