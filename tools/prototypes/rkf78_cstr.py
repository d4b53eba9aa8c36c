"""Would a higher-order pair shorten the cstr's ignition-front fallback?  (round 4, prototype only -- not in the product)

The guarded cstr plan (tsit5g) hands the envs its guard does not trust to DOPRI5 at 1e-10.  The launch is as long as the
heaviest of them: 133-149 attempts x 7 right-hand sides, a dependent chain on one lane.  Here:
the same envs under Fehlberg's 7(8) pair (13 stages) -- sequential RHS evaluations of the heaviest env at equal accuracy
against a 1e-13 solve.  The tableau is written from memory and pinned by its order conditions below (row sums, the
quadrature conditions to order 8 / 7 and the tree conditions sum b_i a_ij c_j^k = 1 / ((k+1)(k+2)) ).

  python tools/prototypes/rkf78_cstr.py [B]
"""
import copy
import os
import sys
from fractions import Fraction as F

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pcgym_amd.config import EnvSpec  # noqa: E402


def tab_rkf78():
    c = [F(0), F(2, 27), F(1, 9), F(1, 6), F(5, 12), F(1, 2), F(5, 6), F(1, 6), F(2, 3), F(1, 3), F(1), F(0), F(1)]
    A = [[F(0)] * 13 for _ in range(13)]
    A[1][0] = F(2, 27)
    A[2][0], A[2][1] = F(1, 36), F(1, 12)
    A[3][0], A[3][2] = F(1, 24), F(1, 8)
    A[4][0], A[4][2], A[4][3] = F(5, 12), F(-25, 16), F(25, 16)
    A[5][0], A[5][3], A[5][4] = F(1, 20), F(1, 4), F(1, 5)
    A[6][0], A[6][3], A[6][4], A[6][5] = F(-25, 108), F(125, 108), F(-65, 27), F(125, 54)
    A[7][0], A[7][4], A[7][5], A[7][6] = F(31, 300), F(61, 225), F(-2, 9), F(13, 900)
    A[8][0], A[8][3], A[8][4], A[8][5], A[8][6], A[8][7] = F(2), F(-53, 6), F(704, 45), F(-107, 9), F(67, 90), F(3)
    A[9][0], A[9][3], A[9][4], A[9][5], A[9][6], A[9][7], A[9][8] = F(-91, 108), F(23, 108), F(-976, 135), F(311, 54), F(-19, 60), F(17, 6), F(-1, 12)
    A[10][0], A[10][3], A[10][4], A[10][5], A[10][6], A[10][7], A[10][8], A[10][9] = \
        F(2383, 4100), F(-341, 164), F(4496, 1025), F(-301, 82), F(2133, 4100), F(45, 82), F(45, 164), F(18, 41)
    A[11][0], A[11][5], A[11][6], A[11][7], A[11][8], A[11][9] = F(3, 205), F(-6, 41), F(-3, 205), F(-3, 41), F(3, 41), F(6, 41)
    A[12][0], A[12][3], A[12][4], A[12][5], A[12][6], A[12][7], A[12][8], A[12][9], A[12][11] = \
        F(-1777, 4100), F(-341, 164), F(4496, 1025), F(-289, 82), F(2193, 4100), F(51, 82), F(33, 164), F(12, 41), F(1)
    b7 = [F(41, 840), 0, 0, 0, 0, F(34, 105), F(9, 35), F(9, 35), F(9, 280), F(9, 280), F(41, 840), 0, 0]
    b8 = [0, 0, 0, 0, 0, F(34, 105), F(9, 35), F(9, 35), F(9, 280), F(9, 280), 0, F(41, 840), F(41, 840)]
    return c, A, [F(x) for x in b7], [F(x) for x in b8]


def check_tableau():
    c, A, b7, b8 = tab_rkf78()
    for i in range(13):
        assert sum(A[i]) == c[i], ("row sum", i)
    for k in range(8):
        assert sum(b * ci ** k for b, ci in zip(b8, c)) == F(1, k + 1), ("quadrature b8", k)
    for k in range(7):
        assert sum(b * ci ** k for b, ci in zip(b7, c)) == F(1, k + 1), ("quadrature b7", k)
    for k in range(1, 6):  # sum_i b_i sum_j a_ij c_j^k = 1 / ((k+1)(k+2))
        for b, top in ((b8, 6), (b7, 5)):
            if k <= top:
                assert sum(b[i] * sum(A[i][j] * c[j] ** k for j in range(13)) for i in range(13)) == F(1, (k + 1) * (k + 2)), ("tree", k)
    # c_i-weighted: sum b_i c_i a_ij c_j = 1/8
    assert sum(b8[i] * c[i] * sum(A[i][j] * c[j] for j in range(13)) for i in range(13)) == F(1, 8)
    return True


def tab_dopri5():
    c = [0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1, 1]
    A = np.zeros((7, 7))
    A[1, 0] = 1 / 5
    A[2, :2] = 3 / 40, 9 / 40
    A[3, :3] = 44 / 45, -56 / 15, 32 / 9
    A[4, :4] = 19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729
    A[5, :5] = 9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656
    A[6, :6] = 35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84
    b5 = A[6].copy()
    b4 = np.array([5179 / 57600, 0, 7571 / 16695, 393 / 640, -92097 / 339200, 187 / 2100, 1 / 40])
    return np.array(c), A, b5, b5 - b4, 5


def adaptive(f, x0, dt, A, b, e, order, tol, h0=None, safety=0.9):
    """Vectorised per-env adaptive integration over [0, dt]; elementary controller (no PI), returns x, attempts."""
    nst = len(b)
    B = x0.shape[1]
    x = x0.copy(); t = np.zeros(B); h = np.full(B, dt if h0 is None else h0)
    att = np.zeros(B, dtype=np.int64)
    live = np.ones(B, dtype=bool)
    while live.any():
        idx = np.nonzero(live)[0]
        hh = np.minimum(h[idx], dt - t[idx])
        xs = x[:, idx]
        k = []
        for i in range(nst):
            z = xs.copy()
            for j in range(i):
                if A[i][j] != 0:
                    z = z + (hh * A[i][j]) * k[j]
            k.append(f(z, idx))
        y = xs.copy(); err = np.zeros_like(xs)
        for i in range(nst):
            if b[i] != 0:
                y = y + (hh * b[i]) * k[i]
            if e[i] != 0:
                err = err + (hh * e[i]) * k[i]
        sc = tol + tol * np.maximum(np.abs(xs), np.abs(y))
        en = np.sqrt(np.mean((err / sc) ** 2, axis=0))
        en = np.where(np.isfinite(en), en, 1e10)
        acc = en <= 1.0
        att[idx] += 1
        fac = np.clip(safety * np.maximum(en, 1e-10) ** (-1.0 / order), 0.2, 5.0)
        x[:, idx[acc]] = y[:, acc]
        t[idx[acc]] += hh[acc]
        h[idx] = hh * fac
        live[idx[acc]] = (dt - t[idx[acc]]) > 1e-14 * dt
    return x, att


if __name__ == "__main__":
    check_tableau()
    print("RKF7(8) tableau: row sums, quadrature to order 8 / 7 and tree conditions hold exactly")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rng = np.random.default_rng(4)
    p_env = bench.workload_params()
    del p_env["integrator"], p_env["substeps"]
    ref = EnvSpec(dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-13, atol=1e-13))
    d10 = EnvSpec(dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-10, atol=1e-10))
    mid, p, dt = ref.model.model_id, np.array(ref.model.param_vector()), ref.dt
    # walk the episode with the oracle's DOPRI5 at 1e-10 and keep every (state, action) pair that takes more than 30 attempts:
    # the ignition front (T 410 -> 500 K, ca 0.45 -> 5e-4 within one env step) is crossed in the middle of the episode
    x = np.stack([rng.uniform(0.7, 1.0, B), rng.uniform(310, 350, B)])
    keep_x, keep_u, amax, amean = [], [], 0, []
    for t in range(60):
        u = rng.uniform(295, 302, (1, B))
        x2, ns = O.integrate(d10, x, u)
        a = ns.sum(axis=0)
        sel = a > 30
        keep_x.append(x[:, sel]); keep_u.append(u[:, sel]); amax = max(amax, int(a.max())); amean.append(a.mean())
        x = x2
    xh = np.concatenate(keep_x, axis=1); u = np.concatenate(keep_u, axis=1)
    uh = np.concatenate([u, np.tile(np.array(p[8:10])[:, None], (1, xh.shape[1]))])
    heavy = slice(None)
    want, _ = O.integrate(ref, xh, u)
    got10, ns10 = O.integrate(d10, xh, u)
    att10 = ns10.sum(axis=0)
    e10 = np.max(np.abs(got10 - want) / np.abs(want), axis=0)
    print(f"dt = {dt:.5f}, B = {B} x 60 steps: oracle DOPRI5 1e-10: attempts per env step mean {np.mean(amean):.2f} max {amax}; {xh.shape[1]} pairs above 30 attempts: "
          f"max {att10.max()} (x7 RHS = {7 * att10.max()}), worst rel err {np.nanmax(e10):.2e}")

    def f(z, idx):
        return O.rhs(mid, p, z, uh[:, idx])

    c5, A5, b5, e5, _ = tab_dopri5()
    y, a = adaptive(f, xh, dt, A5, b5, e5, 5, 1e-10)
    print(f"  this script's DOPRI5 1e-10 (elementary controller) on the same pairs: attempts max {a.max()} (RHS {7 * a.max()}), worst rel err {np.nanmax(np.abs(y - want[:, heavy]) / np.abs(want[:, heavy])):.2e}")
    c, A, b7, b8 = tab_rkf78()
    A8 = [[float(v) for v in r] for r in A]
    e8 = [float(x - y) for x, y in zip(b8, b7)]
    for prop, bb, order in (("7th-order solution", [float(v) for v in b7], 8), ("8th-order solution (local extrapolation)", [float(v) for v in b8], 8)):
        for tol in (1e-8, 1e-9, 1e-10, 1e-11):
            y, a = adaptive(f, xh, dt, A8, bb, e8, order, tol)
            err = np.nanmax(np.abs(y - want[:, heavy]) / np.abs(want[:, heavy]))
            print(f"  RKF7(8) {prop:40s} tol {tol:.0e}: attempts max {a.max():4d} mean {a.mean():6.1f} (RHS max {13 * a.max():5d}), worst rel err {err:.2e}")
