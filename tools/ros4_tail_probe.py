"""attempted steps per env of the Rosenbrock pair on the bench workloads: mean / quantiles / maximum, and where in the
action box the heaviest envs sit"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN  # noqa: E402
from pcgym_amd import MixedVecEnv, VecEnv  # noqa: E402

gen = torch.Generator(device="cuda").manual_seed(1)
_, p, B, _, _ = BN.single_workload("me10")
p["integrator"] = "rodas4"
p.pop("rtol", None), p.pop("atol", None)
env = VecEnv(p, n_envs=B, seed=1234)
segs = BN.mixed_segments(1 << 20)
segs[2][0]["integrator"] = "rodas4"
segs[2][0].pop("rtol", None), segs[2][0].pop("atol", None)
mix = MixedVecEnv(segs, seed=1234)
for name, e, stepper in (("me10", env, None), ("mixed ME segment", mix.envs[2], mix)):
    (stepper or e).reset()
    for i in range(4):
        acts = [2 * torch.rand((q.spec.na, q.B), generator=gen, device="cuda", dtype=torch.float64) - 1
                for q in (mix.envs if stepper else [e])]
        if stepper:
            acts[1] = 0.75 * acts[1] + 0.25
            stepper.step(acts)
            a = acts[2]
        else:
            e.step(acts[0])
            a = acts[0]
        torch.cuda.synchronize()
        att = e.nsteps.sum(dim=0).double()
        q = torch.quantile(att[:1 << 20], torch.tensor([0.5, 0.9, 0.99, 0.999], device="cuda", dtype=torch.float64))
        j = int(att.argmax())
        lo, hi = torch.tensor(e.spec.a_low, device="cuda"), torch.tensor(e.spec.a_high, device="cuda")
        LG = (a[:, j] + 1) * (hi - lo) / 2 + lo
        print(f"{name} step {i}: mean {att.mean():.1f} p50/90/99/99.9 {[round(float(v)) for v in q]} max {int(att.max())} "
              f"at (L,G)=({float(LG[0]):.1f},{float(LG[1]):.1f}) rejected mean {e.nsteps[1].double().mean():.2f}", flush=True)
