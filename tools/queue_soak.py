"""Soak of the work-queue kernel against the classic adaptive kernel: random batch sizes (partial tiles, several tiles per
workgroup, fewer envs than lanes), models, counter modes, auto-reset; states, step counts, outputs must agree (bitwise for
the extraction model).  PCG_Q_FORCE routes every size to the queue.    python tools/queue_soak.py [iterations]"""
import copy
import os
import sys

os.environ["PCG_Q_FORCE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import scenarios as SC  # noqa: E402
from pcgym_amd import VecEnv  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(2024)
    S = SC.scenarios()
    names = ["me_canonical", "me_dist_cons", "me_reactive", "cstr_canonical", "complex_cstr_sp", "biofilm_sp"]
    for it in range(iters):
        name = names[it % len(names)]
        # (past 524,288 envs a workgroup walks several 1024-slot tiles)
        B = int(rng.choice([rng.integers(1, 600), rng.integers(600, 6000), rng.integers(100000, 400000),
                            rng.integers(600000, 1600000)], p=[0.45, 0.37, 0.13, 0.05]))
        pe = bool(rng.integers(0, 2))
        ar = bool(rng.integers(0, 2))
        p = copy.deepcopy(S[name]["env_params"])
        # the extraction scenarios alternate between the explicit pair and the Rosenbrock pair (structured W: its own
        # queue instantiation, tiles up to 2048 slots, one workgroup per CU when the batch fits one tile per CU)
        ros = name.startswith("me_") and name != "me_reactive" and (it // len(names)) % 2 == 1
        p["integrator"] = "rodas4" if ros else "dopri5"
        if ros:
            p.pop("rtol", None), p.pop("atol", None)
        p.pop("noise", None), p.pop("noise_percentage", None)
        q = VecEnv(copy.deepcopy(p), n_envs=B, seed=it, per_env_t=pe, auto_reset=ar)
        c = VecEnv(copy.deepcopy(p), n_envs=B, seed=it, per_env_t=pe, auto_reset=ar, variant=1)
        q.reset(), c.reset()
        if pe:  # spread the per-env counters so that some envs finish (and reset) inside the window
            t0 = torch.tensor(rng.integers(0, q.N - 1, B), dtype=torch.int32, device=q.device)
            q.t_env.copy_(t0), c.t_env.copy_(t0)
        steps = 2 if B > 500000 else 4 if B > 50000 else 8
        for i in range(steps):
            a = torch.tensor(rng.uniform(-1, 1, (q.spec.na, B)), device=q.device)
            if not q.spec.normalise_a:
                lo = torch.tensor(q.spec.a_low, device=q.device)[:, None]
                hi = torch.tensor(q.spec.a_high, device=q.device)[:, None]
                a = lo + (a + 1) / 2 * (hi - lo)
            q.step(a), c.step(a)
            exact = name.startswith("me_c") or name.startswith("me_d")
            same = torch.equal(q.nsteps, c.nsteps)
            if exact:
                assert same and torch.equal(q.x, c.x) and torch.equal(q.rew, c.rew), (it, name, B, pe, ar, i)
            else:
                bad = int((q.nsteps != c.nsteps).any(dim=0).sum())
                assert bad <= max(1, B // 500), (it, name, B, pe, ar, i, bad)
                assert torch.allclose(q.x, c.x, rtol=1e-9, atol=1e-12, equal_nan=True), (it, name, B, pe, ar, i)
            assert torch.equal(q.done, c.done) and torch.equal(q.status, c.status), (it, name, B, pe, ar, i)
            if pe:
                assert torch.equal(q.t_env, c.t_env)
        q.close(), c.close()
        print(it, name, p["integrator"], "B", B, "per_env_t", pe, "auto_reset", ar, "ok", flush=True)
    print("queue soak: %d configurations ok" % iters)


if __name__ == "__main__":
    main()
