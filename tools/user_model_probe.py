"""What a run-time compiled user model costs: the reference's cstr written out as C expressions (PCG_MODEL_USER, hipRTC)
against the built-in cstr kernels, same plan otherwise.  B = 2^20, fp64.

    python tools/user_model_probe.py        (needs a GPU)
"""
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
from pcgym_amd import VecEnv  # noqa: E402
from test_gpu_user_model import CSTR_BY_HAND  # noqa: E402


def run(p, B, variant=None, steps=59, reps=5):
    t0 = time.perf_counter()
    env = VecEnv(p, n_envs=B, seed=1, variant=variant, track_status=False)
    t_create = time.perf_counter() - t0
    acts = [torch.rand((env.spec.na, B), device=env.device, dtype=torch.float64) * 2 - 1 for _ in range(8)]
    if not env.spec.normalise_a:  # physical actions: spread over the action box
        lo = torch.tensor(env.spec.a_low, device=env.device)[:, None]
        hi = torch.tensor(env.spec.a_high, device=env.device)[:, None]
        acts = [lo + (a + 1) / 2 * (hi - lo) for a in acts]
    best = 1e9
    for r in range(reps):
        env.reset()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            env.step(acts[i % 8])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps * 1e3)
    env.close()
    return best, t_create


def main():
    B = 1 << 20
    base = BN.workload_params()
    for integ, kw in (("rk4", dict(substeps=1)), ("rk4", dict(substeps=4)), ("dopri5", dict(rtol=1e-8, atol=1e-8))):
        p = copy.deepcopy(base)
        p.update(integrator=integ, **kw)
        q = copy.deepcopy(p)
        q.pop("model")
        q["custom_model"] = copy.deepcopy(CSTR_BY_HAND)
        t_lean, _ = run(copy.deepcopy(p), B)            # default dispatch (pipelined lean kernel for RK4)
        t_cls, _ = run(copy.deepcopy(p), B, variant=1)  # the classic one-env-per-lane kernel: the user model's shape
        t_usr, t_c = run(q, B)
        print("%-6s %-28s built-in default %.1f us | built-in classic kernel %.1f us | user expressions %.1f us "
              "(plan creation incl. hipRTC %.2f s)" % (integ, kw, t_lean, t_cls, t_usr, t_c))


if __name__ == "__main__":
    main()
