"""Guard calibration for a guarded fixed-step Tsit5 plan of the cstr: worst error of accepted envs against a 1e-13 solve
as a function of the threshold on rho*h, guard evaluated at every stage state and at the end state."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools", "prototypes"))
import scenarios as SC  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pcgym_amd.config import EnvSpec  # noqa: E402
from erk_fixed import tab_tsit5, tab_rk4  # noqa: E402


def guard(p, x):
    q, V, rho_, C, dH, EAR, k0, UA = p[:8]
    ca, T = x
    kk = k0 * np.exp(-EAR / T)
    fb = (-dH / (rho_ * C)) * kk * ca * EAR / (T * T)
    base = q / V + UA / (rho_ * C * V)
    return fb - base, kk + fb + base


def run(tsim, nsub, tab, every_stage, steps, B, rng):
    sc = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
    sc.pop("noise", None), sc.pop("noise_percentage", None)
    sc["tsim"] = tsim
    ref = EnvSpec(dict(sc, integrator="dopri5", rtol=1e-13, atol=1e-13))
    mid, p, dt = ref.model.model_id, np.array(ref.model.param_vector()), ref.dt
    A, b = tab()
    h = dt / nsub
    x = np.stack([rng.uniform(0.7, 1.0, B), rng.uniform(310, 350, B)])
    rec = []
    for t in range(steps):
        u = rng.uniform(295, 302, (1, B))
        uu = np.concatenate([u, np.tile(np.array(p[8:10])[:, None], (1, B))])
        want, _ = O.integrate(ref, x, u)
        y = x.copy()
        gmax = np.full(B, -np.inf); rmax = np.zeros(B)
        for s in range(nsub):
            k = []
            for i in range(len(b)):
                z = y.copy()
                for j in range(i):
                    if A[i, j] != 0:
                        z = z + (h * A[i, j]) * k[j]
                if every_stage or i == 0:
                    g, r = guard(p, z)
                    gmax = np.maximum(gmax, np.where(np.isnan(g), np.inf, g)); rmax = np.maximum(rmax, np.where(np.isnan(r), np.inf, r))
                k.append(O.rhs(mid, p, z, uu))
            for i in range(len(b)):
                if b[i] != 0:
                    y = y + (h * b[i]) * k[i]
        g, r = guard(p, y)
        gmax = np.maximum(gmax, np.where(np.isnan(g), np.inf, g)); rmax = np.maximum(rmax, np.where(np.isnan(r), np.inf, r))
        err = np.max(np.abs(y - want) / np.abs(want), axis=0)
        rec.append((gmax, rmax * h, err))
        x = want
    g = np.concatenate([r[0] for r in rec]); rh = np.concatenate([r[1] for r in rec]); e = np.concatenate([r[2] for r in rec])
    return g, rh, np.where(np.isnan(e), np.inf, e)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for tsim, steps in ((26.0, 40), (52.0, 20)):
        for name, tab, nsub, every in (("rk4 x5 (starts)", tab_rk4, 5, False), ("tsit5 x2 (starts)", tab_tsit5, 2, False), ("tsit5 x2 (stages)", tab_tsit5, 2, True),
                                       ("tsit5 x3 (stages)", tab_tsit5, 3, True)):
            if tsim > 30:
                nsub = 2 * nsub
            g, rh, e = run(tsim, nsub, tab, every, steps, 6000, np.random.default_rng(1))
            calm = g <= 0
            line = []
            for c in (1.5, 2.5, 4.0, 6.0, 10.0, 20.0, 1e9):
                ok = calm & (rh <= c)
                line.append("c=%.1f: %.1e (%.0f%%)" % (c, e[ok].max() if ok.any() else 0, 100 * ok.mean()))
            print("tsim %4.1f %-20s calm %.0f%%  worst calm err %.1e | %s" % (tsim, name, 100 * calm.mean(), e[calm].max(), "  ".join(line)))
