"""The heaviest env of a Rodas4 launch on the lanes that idle beside it: extrapolated linearly implicit Euler, one
sub-step sequence per lane.  (round 4, prototype only -- not in the product)

me10_ros4 and the ME segment of the mixed shard are as long as their heaviest env: ~105 attempts x 6 stages of Rodas4,
each a right-hand side + a structured solve, dependent, on one lane (DESIGN section 3 "Rodas4").  The stiff counterpart of
gbs_lanes_cstr.py: lane j integrates the big step H with n_j = 1, 2, 3, ... linearly implicit Euler sub-steps
(I - h J) d = h f(y), J frozen at the start of the big step; the Aitken-Neville tableau (in h, not h^2) over k lanes has
order k (Deuflhard's SEULEX, fixed column).  The env's chain sees the deepest lane only: n_k (RHS + solve) per big step.

  python tools/prototypes/seulex_lanes_me10.py [B]
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


def jac(f, x, idx):
    """forward-difference Jacobian, (B, nx, nx)"""
    nx, B = x.shape
    f0 = f(x, idx)
    J = np.empty((B, nx, nx))
    for j in range(nx):
        d = 1e-7 * np.maximum(np.abs(x[j]), 1e-3)
        xp = x.copy(); xp[j] += d
        J[:, :, j] = ((f(xp, idx) - f0) / d).T
    return J, f0


def seulex(f, x0, dt, seq, tol, safety=0.9):
    k = len(seq)
    nx, B = x0.shape
    x = x0.copy(); t = np.zeros(B); h = np.full(B, dt / 4)
    att = np.zeros(B, dtype=np.int64)
    live = np.ones(B, dtype=bool)
    I = np.eye(nx)[None]
    while live.any():
        idx = np.nonzero(live)[0]
        H = np.minimum(h[idx], dt - t[idx])
        xs = x[:, idx]
        J, _ = jac(f, xs, idx)
        T = []
        for n in seq:  # one lane each
            hh = H / n
            W = I - hh[:, None, None] * J
            y = xs.copy()
            for _ in range(n):
                d = np.linalg.solve(W, (hh * f(y, idx)).T[:, :, None])[:, :, 0].T
                y = y + d
            T.append(y)
        for j in range(1, k):
            for i in range(k - 1, j - 1, -1):
                T[i] = T[i] + (T[i] - T[i - 1]) / (seq[i] / seq[i - j] - 1)
            if j == k - 2:
                prev = T[k - 2].copy()
        y = T[k - 1]
        err = y - prev
        sc = tol + tol * np.maximum(np.abs(xs), np.abs(y))
        en = np.sqrt(np.mean((err / sc) ** 2, axis=0))
        en = np.where(np.isfinite(en), en, 1e10)
        acc = en <= 1.0
        att[idx] += 1
        fac = np.clip(safety * np.maximum(en, 1e-12) ** (-1.0 / k), 0.1, 4.0)
        x[:, idx[acc]] = y[:, acc]
        t[idx[acc]] += H[acc]
        h[idx] = H * fac
        live[idx[acc]] = (dt - t[idx[acc]]) > 1e-14 * dt
    return x, att, att * seq[-1]


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(7)
    _, p_env, _, _, _ = bench.single_workload("me10_ros4")
    r4 = EnvSpec(copy.deepcopy(p_env))
    ref = EnvSpec(dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-13, atol=1e-13))
    mid, p, dt = ref.model.model_id, np.array(ref.model.param_vector()), ref.dt
    lo, hi = r4.a_low, r4.a_high
    x0 = np.array(r4.x0[: r4.nx], dtype=float) if hasattr(r4, "x0") else None
    x = np.tile(x0[:, None], (1, B)) * (1 + 0.05 * rng.uniform(-1, 1, (r4.nx, B)))
    keep_x, keep_u = [], []
    for t in range(6):
        u = lo[:, None] + rng.uniform(0, 1, (r4.na, B)) * (hi - lo)[:, None]
        x2, ns = O.integrate(r4, x, u)
        a = ns.sum(axis=0)
        sel = a >= np.sort(a)[-64]
        keep_x.append(x[:, sel]); keep_u.append(u[:, sel])
        print(f"step {t}: Rodas4 (the plan's default tolerance, end-point control) attempts mean {a.mean():.1f} p99 {np.quantile(a, 0.99):.0f} max {a.max()}", flush=True)
        x = x2
    xh = np.concatenate(keep_x, axis=1); uh = np.concatenate(keep_u, axis=1)
    want, _ = O.integrate(ref, xh, uh)
    got, ns = O.integrate(r4, xh, uh)
    att = ns.sum(axis=0)
    rel = lambda y: np.nanmax(np.abs(y - want) / np.maximum(np.abs(want), 1e-300))  # noqa: E731
    print(f"{xh.shape[1]} heaviest (state, action) pairs: Rodas4 attempts max {att.max()} mean {att.mean():.1f} = {6 * att.max()} dependent stages (RHS + solve), worst rel err {rel(got):.2e}")
    nu_full = ref.nu if hasattr(ref, "nu") else uh.shape[0]
    uu = uh if uh.shape[0] == nu_full else np.concatenate([uh, np.tile(np.array(p[-(nu_full - uh.shape[0]):])[:, None], (1, uh.shape[1]))])

    def f(z, idx):
        return O.rhs(mid, p, z, uu[:, idx])

    with np.errstate(all="ignore"):
        for seq in ((1, 2, 3, 4), (1, 2, 3, 4, 5, 6), (1, 2, 3, 4, 5, 6, 7, 8), (1, 2, 3, 4, 6, 8, 12, 16)):
            for tol in (1e-6, 1e-7, 1e-8, 1e-9):
                y, a, depth = seulex(f, xh, dt, seq, tol)
                print(f"  SEULEX on {len(seq)} lanes (deepest lane {seq[-1]:2d} sub-steps) tol {tol:.0e}: big steps max {a.max():3d} mean {a.mean():5.1f}; "
                      f"dependent (RHS + solve) max {depth.max():4d} mean {depth.mean():6.1f}; worst rel err {rel(y):.2e}", flush=True)
