"""Calibration of the oracle's SEULEX-8 (the heavy envs of a Rodas4 plan) on the me10_ros4 workload: accuracy against a
1e-13 solve and big-step counts over the tunables; which envs a cost key would pick.  CPU only (oracle).

  python tools/prototypes/seulex8_calib.py [B] [steps]
"""
import copy
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pcgym_amd.config import EnvSpec  # noqa: E402


def seulex(spec, x, u):
    cfg, keep = spec.to_cfg()
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    u = np.ascontiguousarray(u, dtype=np.float64)
    ns = np.zeros((2, x.shape[1]), dtype=np.int32)
    fn = O.lib().orc_seulex8
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = fn(C.byref(cfg), x.shape[1], O._p(x), O._p(u), O._p(ns))
    assert rc == 0, rc
    return x, ns


def episode(B, steps, seed=7):
    rng = np.random.default_rng(seed)
    _, p_env, _, _, _ = bench.single_workload("me10_ros4")
    r4 = EnvSpec(copy.deepcopy(p_env))
    ref = EnvSpec(dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-13, atol=1e-13))
    lo, hi = r4.a_low, r4.a_high
    x0 = np.array(r4.x0[: r4.nx], dtype=float)
    x = np.tile(x0[:, None], (1, B)) * (1 + 0.05 * rng.uniform(-1, 1, (r4.nx, B)))
    X, U, A = [], [], []
    for t in range(steps):
        u = lo[:, None] + rng.uniform(0, 1, (r4.na, B)) * (hi - lo)[:, None]
        x2, ns = O.integrate(r4, x, u)
        X.append(x); U.append(u); A.append(ns.sum(axis=0))
        x = x2
    return r4, ref, np.concatenate(X, 1), np.concatenate(U, 1), np.concatenate(A)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    r4, ref, X, U, A = episode(B, steps)
    print(f"{X.shape[1]} (state, action) pairs; Rodas4 attempts mean {A.mean():.1f} p90 {np.quantile(A, .9):.0f} p99 {np.quantile(A, .99):.0f} max {A.max()}")
    for thr in (40, 50, 60, 70, 80):
        print(f"  attempts >= {thr}: {np.mean(A >= thr) * 100:.2f} % of the envs, {A[A >= thr].sum() / A.sum() * 100:.1f} % of the attempts")
    sel = A >= 45
    xh, uh, ah = X[:, sel], U[:, sel], A[sel]
    nu_full = 4
    p = np.array(ref.model.param_vector())
    uu = np.concatenate([uh, np.tile(p[-2:][:, None], (1, uh.shape[1]))]) if uh.shape[0] < nu_full else uh
    want, _ = O.integrate(ref, xh, uu if False else uh)
    got, _ = O.integrate(r4, xh, uh)
    rel = lambda y: np.abs(y - want) / np.maximum(np.abs(want), 1e-300)  # noqa: E731
    print(f"{sel.sum()} pairs with >= 45 attempts: Rodas4 worst rel err {np.nanmax(rel(got)):.2e}")
    lib = O.lib()
    lib.orc_set_seulex.argtypes = [C.c_double] * 5 + [C.c_int]
    lib.orc_set_seulex.restype = None
    print("tol  h0  safety facmax facmin ep | big steps max mean (rej mean) | worst err | corr(big steps, r4 attempts)")
    for ep in (1, 0):
        for tol in (2.0, 4.0, 8.0):
            for h0 in (2.0, 8.0, 32.0):
                for safety, facmax in ((0.8, 2.0), (0.8, 4.0), (0.9, 4.0)):
                    lib.orc_set_seulex(tol, h0, safety, facmax, 0.1, ep)
                    y, ns = seulex(r4, xh, uh)
                    a = ns.sum(axis=0)
                    e = np.nanmax(rel(y), axis=0)
                    print(f"{tol:4.1f} {h0:4.0f} {safety:.1f} {facmax:.0f} 0.1 {ep} | {a.max():3d} {a.mean():5.1f} ({ns[1].mean():4.2f}) | {np.nanmax(e):.2e} | nan {np.isnan(y).any()}", flush=True)
