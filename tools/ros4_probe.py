"""Per-attempt cost of the adaptive kernels on uniform batches (every env the same input and state, so every lane takes
the same number of attempts): lone-wave latency (one env per lane, one wave per SIMD) against shared-SIMD throughput.
  python tools/ros4_probe.py [integrator ...]"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import scenarios as SC  # noqa: E402
from pcgym_amd import VecEnv  # noqa: E402

os.environ["PCG_Q_FORCE"] = "1"
integs = sys.argv[1:] or ["rodas4", "dopri5"]
for integ in integs:
    for B, label in ((1 << 16, "1 env/lane, 256 WG: one wave per SIMD"), (1 << 17, "1 env/lane, 512 WG: two waves per SIMD"),
                     (1 << 18, "2 envs/lane, two waves per SIMD"), (1 << 20, "8 envs/lane")):
        for LG in ((5.0, 1000.0), (250.0, 500.0)):
            p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
            p.update(integrator=integ, N=400, tsim=400.0, SP={"X5": [0.3] * 400})
            env = VecEnv(p, n_envs=B)
            env.reset()
            lo, hi = env.spec.a_low, env.spec.a_high
            an = [2 * (LG[i] - lo[i]) / (hi[i] - lo[i]) - 1 for i in range(2)]
            acts = []
            for k in range(2):  # alternate between two inputs so that every step has a transient
                f = 1.0 if k == 0 else 0.6
                a = torch.tensor([[an[0] * f], [an[1] * f]], device="cuda", dtype=torch.float64).repeat(1, B)
                acts.append(a.contiguous())
            for i in range(6):
                env.step(acts[i % 2])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n, att = 20, 0.0
            e0.record()
            for i in range(n):
                env.step(acts[i % 2])
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            for i in range(2):
                env.step(acts[i % 2])
                att += float(env.nsteps.sum(dim=0).double().mean().item()) / 2
            print(f"{integ:7s} B=2^{int(np.log2(B))} ({label}) (L,G)={LG}: {ms*1e3:8.1f} us/step, {att:6.1f} attempts/env "
                  f"-> {ms*1e3/att/ max(1, B >> 17):6.2f} us per attempt-round", flush=True)
            env.close()
