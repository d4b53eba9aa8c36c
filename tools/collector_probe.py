"""Closed-loop collector (collect_rollouts with an on-device policy) and reset against raw stepping (needs a GPU)."""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import bench as BN  # noqa: E402
from pcgym_amd import VecEnv, collect_rollouts  # noqa: E402


def main():
    B = 1 << 20
    p = BN.workload_params()
    env = VecEnv(copy.deepcopy(p), n_envs=B, seed=1)
    N = env.spec.N
    acts = torch.rand((N, 1, B), device=env.device, dtype=torch.float64) * 2 - 1

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    def raw():
        env.reset()
        for i in range(N - 1):
            env.step(acts[i])

    t_raw = timed(raw)
    t_open = timed(lambda: collect_rollouts(env, actions=acts))
    pol = lambda obs: -0.5 * obs[:, :1]  # noqa: E731  (B,1) on device
    t_closed = timed(lambda: collect_rollouts(env, policy=pol))
    t_reset = timed(lambda: env.reset(), reps=5)
    steps = (N - 1) * B
    print("B = %d, N = %d" % (B, N))
    print("raw reset + %d steps          %.2f ms  %.2e env-steps/s" % (N - 1, t_raw * 1e3, steps / t_raw))
    print("collect_rollouts open loop    %.2f ms  %.2e env-steps/s (fused rollout kernel writes x / r in the reference's axis order)" % (t_open * 1e3, steps / t_open))
    print("collect_rollouts closed loop  %.2f ms  %.2e env-steps/s (policy + per-step recording of x, u, r)" % (t_closed * 1e3, steps / t_closed))
    print("reset()                       %.3f ms" % (t_reset * 1e3))
    env.close()


if __name__ == "__main__":
    main()
