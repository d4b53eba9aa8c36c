"""Fixed-step explicit Runge-Kutta candidates for the cheap models' default plans (cstr under its guard, four_tank):
worst relative one-step error against a 1e-13 solve per scheme and RHS-evaluation count, on states sampled along
tight full-box episodes.  numpy + the C oracle's batched RHS (CPU only)."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import scenarios as SC  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pcgym_amd.config import EnvSpec  # noqa: E402

s21 = np.sqrt(21.0)


def tab_rk4():
    A = np.zeros((4, 4)); A[1, 0] = .5; A[2, 1] = .5; A[3, 2] = 1
    return A, np.array([1, 2, 2, 1]) / 6.0


def tab_dp5():
    A = np.zeros((6, 6))
    A[1, :1] = [1 / 5]
    A[2, :2] = [3 / 40, 9 / 40]
    A[3, :3] = [44 / 45, -56 / 15, 32 / 9]
    A[4, :4] = [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729]
    A[5, :5] = [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]
    return A, np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84])


def tab_tsit5():
    import ctypes as C
    a = (C.c_double * 42)(); e = (C.c_double * 7)()
    O.lib().orc_tsit5_tableau(a, e)
    T = np.array(list(a)).reshape(7, 6)
    A = np.zeros((6, 6))
    A[:, :] = T[:6, :]
    return A, T[6].copy()


def tab_cv8():
    """Cooper & Verner (1972) 11-stage order 8"""
    A = np.zeros((11, 11))
    A[1, 0] = 1 / 2
    A[2, :2] = [1 / 4, 1 / 4]
    A[3, :3] = [1 / 7, (-7 - 3 * s21) / 98, (21 + 5 * s21) / 49]
    A[4, :4] = [(11 + s21) / 84, 0, (18 + 4 * s21) / 63, (21 - s21) / 252]
    A[5, :5] = [(5 + s21) / 48, 0, (9 + s21) / 36, (-231 + 14 * s21) / 360, (63 - 7 * s21) / 80]
    A[6, :6] = [(10 - s21) / 42, 0, (-432 + 92 * s21) / 315, (633 - 145 * s21) / 90, (-504 + 115 * s21) / 70, (63 - 13 * s21) / 35]
    A[7, :7] = [1 / 14, 0, 0, 0, (14 - 3 * s21) / 126, (13 - 3 * s21) / 63, 1 / 9]
    A[8, :8] = [1 / 32, 0, 0, 0, (91 - 21 * s21) / 576, 11 / 72, (-385 - 75 * s21) / 1152, (63 + 13 * s21) / 128]
    A[9, :9] = [1 / 14, 0, 0, 0, 1 / 9, (-733 - 147 * s21) / 2205, (515 + 111 * s21) / 504, (-51 - 11 * s21) / 56, (132 + 28 * s21) / 245]
    A[10, :10] = [0, 0, 0, 0, (-42 + 7 * s21) / 18, (-18 + 28 * s21) / 45, (-273 - 53 * s21) / 72, (301 + 53 * s21) / 72, (28 - 28 * s21) / 45, (49 - 7 * s21) / 18]
    b = np.array([1 / 20, 0, 0, 0, 0, 0, 0, 49 / 180, 16 / 45, 49 / 180, 1 / 20])
    return A, b


def tab_butcher6():
    """Butcher's 7-stage order 6"""
    A = np.zeros((7, 7))
    A[1, :1] = [1 / 3]
    A[2, :2] = [0, 2 / 3]
    A[3, :3] = [1 / 12, 1 / 3, -1 / 12]
    A[4, :4] = [-1 / 16, 9 / 8, -3 / 16, -3 / 8]
    A[5, :5] = [0, 9 / 8, -3 / 8, -3 / 4, 1 / 2]
    A[6, :6] = [9 / 44, -9 / 11, 63 / 44, 18 / 11, 0, -16 / 11]
    b = np.array([11 / 120, 0, 27 / 40, 27 / 40, -4 / 15, -4 / 15, 11 / 120])
    return A, b


def check_order(A, b, name):
    """convergence order on a nonlinear test problem"""
    def f(y):
        return np.array([y[1], -np.sin(y[0]) * (1 + 0.3 * y[1] * y[1]), np.cos(y[0]) * y[2]])
    def step(y, h):
        k = []
        for i in range(len(b)):
            k.append(f(y + h * sum(A[i, j] * k[j] for j in range(i))) if i else f(y))
        return y + h * sum(b[i] * k[i] for i in range(len(b)))
    def run(n):
        y = np.array([0.7, 0.3, 1.0])
        for _ in range(n):
            y = step(y, 2.0 / n)
        return y
    ref = run(4096) if len(b) < 8 else run(512)
    e = [np.abs(run(n) - ref).max() for n in (4, 8, 16)]
    print("%-10s row sums %.1e  sum b %.1e  errors %s  orders %.2f %.2f" % (
        name, np.abs(A.sum(1)[1:] - 0).min() * 0, abs(b.sum() - 1), ["%.2e" % v for v in e], np.log2(e[0] / e[1]), np.log2(e[1] / e[2])))


def erk(mid, p, x, u, dt, n, A, b):
    h = dt / n
    x = x.copy()
    s = len(b)
    for _ in range(n):
        k = []
        for i in range(s):
            y = x.copy()
            for j in range(i):
                if A[i, j] != 0:
                    y = y + (h * A[i, j]) * k[j]
            k.append(O.rhs(mid, p, y, u))
        for i in range(s):
            if b[i] != 0:
                x = x + (h * b[i]) * k[i]
    return x


SCHEMES = [("rk4 x4", tab_rk4, 4), ("rk4 x5", tab_rk4, 5), ("dp5 x2", tab_dp5, 2), ("tsit5 x2", tab_tsit5, 2), ("dp5 x3", tab_dp5, 3), ("tsit5 x3", tab_tsit5, 3),
           ("butcher6 x1", tab_butcher6, 1), ("butcher6 x2", tab_butcher6, 2), ("cv8 x1", tab_cv8, 1), ("cv8 x2", tab_cv8, 2)]

if __name__ == "__main__":
    for nm, tb in (("rk4", tab_rk4), ("dp5", tab_dp5), ("tsit5", tab_tsit5), ("butcher6", tab_butcher6), ("cv8", tab_cv8)):
        check_order(*tb(), nm)
    rng = np.random.default_rng(0)
    B = 20000
    for model in ("cstr", "four_tank"):
        sc = copy.deepcopy(SC.scenarios()[model + "_canonical"]["env_params"])
        sc.pop("noise", None), sc.pop("noise_percentage", None)
        ref = EnvSpec(dict(sc, integrator="dopri5", rtol=1e-13, atol=1e-13))
        mid, p, dt = ref.model.model_id, np.array(ref.model.param_vector()), ref.dt
        lo, hi = np.array(sc["a_space"]["low"], dtype=float), np.array(sc["a_space"]["high"], dtype=float)
        if model == "cstr":
            x = np.stack([rng.uniform(0.7, 1.0, B), rng.uniform(310, 350, B)])
        else:
            x = np.tile(np.array(sc["x0"][:4], dtype=float)[:, None], (1, B)) * rng.uniform(0.5, 2.0, (4, B))
            lo = lo + 0.25 * (hi - lo)  # bench distribution: upper 3/4 of the box
        worst = {k[0]: 0.0 for k in SCHEMES}
        for t in range(12):
            u = rng.uniform(lo[:, None], hi[:, None], (len(lo), B))
            want, _ = O.integrate(ref, x, u)
            ok = np.ones(B, bool)
            if model == "cstr":  # guard-accepted envs of the current plan (calm at both ends, rr*dt/5 <= 1)
                plan = EnvSpec(dict(sc))
                _, ns = O.integrate(plan, x, u)
                ok = ns.sum(axis=0) == 0
            for nm, tb, n in SCHEMES:
                A, b = tb()
                got = erk(mid, p, x, u, dt, n, A, b)
                err = np.max(np.abs(got - want) / np.abs(want), axis=0)
                worst[nm] = max(worst[nm], float(err[ok].max()))
            x = want
        print(model, "dt", dt, "guard-accepted" if model == "cstr" else "")
        for nm, tb, n in SCHEMES:
            print("   %-12s %3d evaluations  worst rel err %.2e" % (nm, len(tb()[1]) * n, worst[nm]))
