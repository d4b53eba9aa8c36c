"""Calibration of T5G_EST_TOL (the embedded-estimate threshold of the guarded Tsit5 plan, oracle/pcg_oracle.c: t5g): worst
error of TRUSTED envs against a 1e-13 solve and the share of escalated envs, on (a) the canonical closed loop, (b) full-box
episodes, (c) a deliberately wide box of states / jacket temperatures / step sizes, (d) the two points of ADVICE r3."""
import copy
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import scenarios as SC  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pcgym_amd.config import EnvSpec  # noqa: E402

O.build()
lib = O._load() if hasattr(O, "_load") else None


def spec(tsim, **kw):
    sc = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
    sc.pop("noise", None), sc.pop("noise_percentage", None)
    sc["tsim"] = tsim
    sc.update(kw)
    return EnvSpec(sc)


def errs(plan, ref, x, u):
    want, _ = O.integrate(ref, x, u)
    got, ns = O.integrate(plan, x, u)
    # scaled by the reference's own tolerances (CasADi CVODES defaults: reltol 1e-6, abstol 1e-8): <= 1 is "inside its class"
    err = np.max(np.abs(got - want) / (1e-6 * np.abs(want) + 1e-8), axis=0)
    return want, err, ns.sum(axis=0) > 0


def main():
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libpcg_oracle.so"))
    L.orc_set_t5g_est_tols.argtypes = [ctypes.c_double, ctypes.c_double]
    rng = np.random.default_rng(0)
    for tol, at in ((2e-7, 2e-7), (2e-7, 2e-9), (5e-7, 5e-9), (1e-6, 1e-8), (2e-6, 2e-8), (1e-7, 1e-9)):
        L.orc_set_t5g_est_tols(tol, at)
        out = [f"rtol {tol:7.1e} atol {at:7.1e}"]
        for tsim, lab in ((26.0, "26/60"), (1.0, "1/60")):
            ref, plan = spec(tsim, integrator="dopri5", rtol=1e-13, atol=1e-13), spec(tsim, integrator="tsit5g")
            B = 4000
            # (a) canonical closed loop
            x = np.stack([np.full(B, 0.8), np.full(B, 330.0)])
            esc_a = 0.0
            for t in range(20):
                u = rng.uniform(295, 302, (1, B))
                x, e, esc = errs(plan, ref, x, u)
                esc_a = max(esc_a, esc.mean())
            # (b) full-box episodes
            x = np.stack([rng.uniform(0.7, 1.0, B), rng.uniform(310, 350, B)])
            wb, fb = 0.0, []
            for t in range(12 if tsim > 2 else 60):
                u = rng.uniform(295, 302, (1, B))
                x, e, esc = errs(plan, ref, x, u)
                wb = max(wb, e[~esc].max())
                fb.append(esc.mean())
            out.append(f"dt {lab}: loop esc {esc_a:.4f} | box acc-err {wb:.1e} esc {np.mean(fb):.3f}")
        # (c) wide box, three step sizes
        ww, fw = 0.0, []
        for tsim in (1.0, 5.0, 26.0):
            ref, plan = spec(tsim, integrator="dopri5", rtol=1e-13, atol=1e-13), spec(tsim, integrator="tsit5g")
            B = 20000
            x = np.stack([rng.uniform(0.0, 1.2, B) ** 2, rng.uniform(290, 600, B)])
            u = rng.uniform(280, 320, (1, B))
            _, e, esc = errs(plan, ref, x, u)
            ok = np.isfinite(e)
            ww = max(ww, e[~esc & ok].max())
            fw.append(esc.mean())
        out.append(f"wide acc-err {ww:.1e} esc {np.mean(fw):.3f}")
        # (d) ADVICE r3's points
        pts = []
        for (ca, T, Tc, tsim) in ((0.0036, 375.5, 285.7, 26.0), (0.005, 380.0, 300.0, 5.0)):
            ref, plan = spec(tsim, integrator="dopri5", rtol=1e-13, atol=1e-13), spec(tsim, integrator="tsit5g")
            _, e, esc = errs(plan, ref, np.array([[ca], [T]]), np.array([[Tc]]))
            pts.append(f"{e[0]:.1e}{'E' if esc[0] else 'A'}")
        out.append("advice " + " ".join(pts))
        print(" | ".join(out), flush=True)


if __name__ == "__main__":
    main()
