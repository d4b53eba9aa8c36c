"""Rodas3 (stiff-capable integrator) on the GPU: parity with the oracle twin on every LSODA fixture, cost against the
explicit pair at the models' canonical stiffness, and a stiffened extraction column where the explicit pair gives up.

    python tools/rodas_probe.py            (needs a GPU; prints a table)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import helpers as H  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only: this is a test tool)
from pcgym_amd import _abi as abi  # noqa: E402
from test_gpu_parity import _plan_for  # noqa: E402
from test_oracle_golden import TIGHT_CASES, _spec_for_integration  # noqa: E402


def integrate(spec, xs, us, reps=1):
    lib, plan = _plan_for(spec, torch)
    x0 = torch.tensor(xs, device="cuda")
    u = torch.tensor(us, device="cuda")
    ns = torch.zeros((2, x0.shape[1]), dtype=torch.int32, device="cuda")
    best = 1e30
    for _ in range(reps):
        x = x0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.pcg_integrate(plan, x.shape[1], x.data_ptr(), u.data_ptr(), ns.data_ptr(), None)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        assert rc == 0, rc
    lib.pcg_plan_destroy(plan)
    return x.cpu().numpy(), ns.cpu().numpy(), best


def main():
    print("== parity with the oracle twin (rtol 1e-6, atol 1e-8) ==")
    for fix, model, *_ in TIGHT_CASES:
        g = H.gold("tight_" + fix)
        spec = _spec_for_integration(model, float(g["dt"]), g["u"].shape[1], integrator="rodas3", rtol=1e-6, atol=1e-8,
                                     max_steps=100000)
        xs, us = g["x"].T.copy(), g["u"].T.copy()
        got, ns, _ = integrate(spec, xs, us)
        want, ns_o = O.integrate(spec, xs, us)
        sc = np.maximum(np.abs(want), 1e-6 * np.max(np.abs(want), axis=1, keepdims=True))
        err = np.nanmax(np.abs(got - want) / sc)
        t = g["xf"].T
        st = np.maximum(np.abs(t), 1e-6 * np.max(np.abs(t), axis=1, keepdims=True))
        print("%-32s nx %2d  same counts %.3f  max err vs oracle %.1e  vs LSODA %.1e  steps %.0f+%.0f" % (
            fix, spec.nx, np.mean(np.all(ns == ns_o, axis=0)), err, np.nanmax(np.abs(got - t) / st), ns[0].mean(), ns[1].mean()))
    print("== cost at canonical stiffness, B = 65536 (tiled fixture samples) ==")
    for fix, model in [("cstr", "cstr"), ("multistage_extraction", "multistage_extraction"),
                       ("multistage_extraction_reactive", "multistage_extraction_reactive"),
                       ("crystallization", "crystallization"), ("four_tank", "four_tank")]:
        g = H.gold("tight_" + fix)
        B = 65536
        reps = B // g["x"].shape[0] + 1
        xs = np.tile(g["x"].T, (1, reps))[:, :B].copy()
        us = np.tile(g["u"].T, (1, reps))[:, :B].copy()
        row = []
        for integ in ("dopri5", "rodas3"):
            spec = _spec_for_integration(model, float(g["dt"]), g["u"].shape[1], integrator=integ, rtol=1e-6, atol=1e-8,
                                         max_steps=100000)
            _, ns, tt = integrate(spec, xs, us, reps=3)
            row.append("%s %.2f ms (%.0f+%.0f steps)" % (integ, tt * 1e3, ns[0].mean(), ns[1].mean()))
        print("%-32s %s" % (fix, " | ".join(row)))
    print("== stiffened extraction column: hold-ups / S (|lambda| dt ~ 240 S at the top of the action box) ==")
    g = H.gold("tight_multistage_extraction")
    B = 4096
    reps = B // g["x"].shape[0] + 1
    xs = np.tile(g["x"].T, (1, reps))[:, :B].copy()
    us = np.tile(g["u"].T, (1, reps))[:, :B].copy()
    for S in (1, 10, 100, 1000):
        row = []
        res = {}
        for integ in ("dopri5", "rodas3"):
            spec = _spec_for_integration("multistage_extraction", float(g["dt"]), g["u"].shape[1], integrator=integ,
                                         rtol=1e-6, atol=1e-8, max_steps=20000)
            spec.model.parameters["Vl"] = 5.0 / S  # a private copy of the registry entry (models.get_model)
            spec.model.parameters["Vg"] = 5.0 / S
            got, ns, tt = integrate(spec, xs, us, reps=2)
            res[integ] = got
            row.append("%s %.2f ms, %.0f+%.0f steps, failed %.3f" % (integ, tt * 1e3, ns[0].mean(), ns[1].mean(),
                                                                      np.mean(np.isnan(got).any(axis=0))))
        both = ~(np.isnan(res["dopri5"]).any(axis=0) | np.isnan(res["rodas3"]).any(axis=0))
        d = np.max(np.abs(res["dopri5"] - res["rodas3"])[:, both]) if both.any() else float("nan")
        print("S = %4d  %s | max |dp5 - rodas3| %.1e" % (S, " | ".join(row), d))


if __name__ == "__main__":
    main()
