"""Host cost of one VecEnv.step() call (Python + ctypes + launch): tiny batch, so the GPU is never the limit (needs a GPU)."""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench as BN  # noqa: E402
from pcgym_amd import VecEnv  # noqa: E402


def main():
    for kw in ({}, {"per_env_t": True, "auto_reset": True}):
        env = VecEnv(copy.deepcopy(BN.workload_params()), n_envs=1024, seed=1, **kw)
        a = torch.zeros((1, 1024), dtype=torch.float64, device=env.device)
        env.reset()
        for _ in range(200):
            env.step(a)
            if env.t >= env.N - 1 and not kw:
                env.reset()
        torch.cuda.synchronize()
        n = 0
        t0 = time.perf_counter()
        for ep in range(200):
            if not kw:
                env.reset()
            for _ in range(env.N - 1):
                env.step(a)
                n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-40s %.2f us per step() call (B = 1024, %d calls)" % (kw or "lock-stepped", dt / n * 1e6, n))
        env.close()


if __name__ == "__main__":
    main()
