"""the ME segment of configs[4] on its own: launch geometry of the Rodas4 work-queue kernel against the batch size"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN  # noqa: E402
from pcgym_amd import VecEnv  # noqa: E402

p = BN.mixed_segments(1 << 20)[2][0]
for B in (262144, 349524, 524288):
    for env_kw in ({}, {"PCG_Q_BPC": "2"}, {"PCG_Q_BPC": "1"}):
        for k in ("PCG_Q_BPC",):
            os.environ.pop(k, None)
        os.environ.update(env_kw)
        env = VecEnv(p, n_envs=B, seed=1234)
        env.reset()
        gen = torch.Generator(device="cuda").manual_seed(1)
        acts = [2 * torch.rand((2, B), generator=gen, device="cuda", dtype=torch.float64) - 1 for _ in range(4)]
        for i in range(6):
            env.step(acts[i % 4])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 30
        for i in range(n):
            env.step(acts[i % 4])
        e1.record()
        torch.cuda.synchronize()
        att = env.nsteps.sum(dim=0).double()
        print(f"B={B} {env_kw or 'default'}: {e0.elapsed_time(e1) / n * 1e3:.1f} us/step  attempts mean {att.mean():.1f} max {int(att.max())}",
              flush=True)
        env.close()
