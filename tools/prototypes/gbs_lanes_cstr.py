"""The ignition-front fallback on the lanes that idle beside it: explicit extrapolation (Gragg-Bulirsch-Stoer), one
sub-step sequence per lane.  (round 4, prototype only -- not in the product)

In the fix-up launch of the guarded cstr plan the heaviest env sits alone in its wave: 153 attempts x 7 right-hand sides =
1071 DEPENDENT evaluations on one lane while 63 lanes wait.  A higher-order one-lane pair does not
help (rkf78_cstr.py: 858).  Extrapolation is the method whose work is parallel by construction: lane j integrates the same
big step H with Gragg's modified midpoint rule in n_j = 2, 4, 6, ... sub-steps, the Aitken-Neville tableau over the lanes'
results (cross-lane reads) has order 2k.  What the env's chain sees per big step is the DEEPEST lane only: n_k + 1
evaluations.  Here: that depth, summed over the steps the controller takes, at the accuracy DOPRI5 1e-10 delivers.

  python tools/prototypes/gbs_lanes_cstr.py [B]
"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pcgym_amd.config import EnvSpec  # noqa: E402


def midpoint(f, x, H, n, idx):
    """Gragg's modified midpoint rule with smoothing step: n sub-steps, n + 1 evaluations."""
    h = H / n
    z0 = x
    z1 = x + h * f(x, idx)
    for _ in range(n - 1):
        z0, z1 = z1, z0 + 2 * h * f(z1, idx)
    return 0.5 * (z0 + z1 + h * f(z1, idx))


def gbs(f, x0, dt, seq, tol, safety=0.9):
    """per-env adaptive extrapolation over [0, dt] with a fixed column count k = len(seq): returns x, big steps attempted,
    sequential depth in RHS evaluations (deepest lane per attempt)."""
    k = len(seq)
    B = x0.shape[1]
    x = x0.copy(); t = np.zeros(B); h = np.full(B, dt)
    att = np.zeros(B, dtype=np.int64)
    live = np.ones(B, dtype=bool)
    while live.any():
        idx = np.nonzero(live)[0]
        H = np.minimum(h[idx], dt - t[idx])
        xs = x[:, idx]
        T = [midpoint(f, xs, H, n, idx) for n in seq]  # one lane each
        # Aitken-Neville in H^2
        for j in range(1, k):
            for i in range(k - 1, j - 1, -1):
                r = (seq[i] / seq[i - j]) ** 2
                T[i] = T[i] + (T[i] - T[i - 1]) / (r - 1)
            if j == k - 2:
                prev = T[k - 2].copy()  # T_{k-1,k-1}
        y = T[k - 1]
        err = y - prev
        sc = tol + tol * np.maximum(np.abs(xs), np.abs(y))
        en = np.sqrt(np.mean((err / sc) ** 2, axis=0))
        en = np.where(np.isfinite(en), en, 1e10)
        acc = en <= 1.0
        att[idx] += 1
        fac = np.clip(safety * np.maximum(en, 1e-12) ** (-1.0 / (2 * k - 1)), 0.1, 4.0)
        x[:, idx[acc]] = y[:, acc]
        t[idx[acc]] += H[acc]
        h[idx] = H * fac
        live[idx[acc]] = (dt - t[idx[acc]]) > 1e-14 * dt
    return x, att, att * (seq[-1] + 1)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    rng = np.random.default_rng(4)
    p_env = bench.workload_params()
    del p_env["integrator"], p_env["substeps"]
    ref = EnvSpec(dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-13, atol=1e-13))
    d10 = EnvSpec(dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-10, atol=1e-10))
    mid, p, dt = ref.model.model_id, np.array(ref.model.param_vector()), ref.dt
    x = np.stack([rng.uniform(0.7, 1.0, B), rng.uniform(310, 350, B)])
    keep_x, keep_u = [], []
    for t in range(60):
        u = rng.uniform(295, 302, (1, B))
        x2, ns = O.integrate(d10, x, u)
        sel = ns.sum(axis=0) > 30
        keep_x.append(x[:, sel]); keep_u.append(u[:, sel])
        x = x2
    xh = np.concatenate(keep_x, axis=1); u = np.concatenate(keep_u, axis=1)
    uh = np.concatenate([u, np.tile(np.array(p[8:10])[:, None], (1, xh.shape[1]))])
    want, _ = O.integrate(ref, xh, u)
    got10, ns10 = O.integrate(d10, xh, u)
    att10 = ns10.sum(axis=0)
    print(f"{xh.shape[1]} (state, action) pairs of a {B}-env episode that take DOPRI5 1e-10 more than 30 attempts: max {att10.max()} attempts = "
          f"{7 * att10.max()} dependent RHS evaluations, worst rel err {np.nanmax(np.abs(got10 - want) / np.abs(want)):.2e}")

    def f(z, idx):
        return O.rhs(mid, p, z, uh[:, idx])

    with np.errstate(all="ignore"):
        for seq in ((2, 4, 6, 8), (2, 4, 6, 8, 10, 12), (2, 4, 6, 8, 10, 12, 14, 16), (2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24)):
            for tol in (1e-8, 1e-9, 1e-10, 1e-11):
                y, a, depth = gbs(f, xh, dt, seq, tol)
                err = np.nanmax(np.abs(y - want) / np.abs(want))
                print(f"  GBS on {len(seq):2d} lanes (order {2 * len(seq):2d}, deepest lane {seq[-1] + 1:2d} RHS) tol {tol:.0e}: big steps max {a.max():3d} mean {a.mean():5.1f}; "
                      f"dependent RHS max {depth.max():4d} mean {depth.mean():6.1f}; worst rel err {err:.2e}")
