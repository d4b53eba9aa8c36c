#!/usr/bin/env python3
"""The accuracy contract of the extraction model's DEFAULT plan (integrator 'rodas5', rtol = atol = 8e-8 / max(1, dt),
end-point exponents under the step cap) beyond the sample it was calibrated on (VERDICT r5 "next" 7).

CPU only (the oracle twin of the kernels is bit-identical in step sequence and state): ~1e5 env steps from 200-step
random-action episodes -- X0 ~ N(0.6, 0.02) and Y6 ~ N(0.05, 0.01) per env and step (configs[4]'s disturbances), set-point
changes, eq_exponent in {1.5, 2, 3}, dt in {0.2, 1, 2, 5} -- each step compared with a 1e-13 solve FROM THE SAME STATE under
the same held input.  The error is reported in units of the reference's own tolerances, |diff| / (1e-6 |x| + 1e-8)
(CasADi CVODES defaults, integrator.py:163-182): the bar the cstr default plan is held to is 3 units.

    python tools/rodas5_contract.py [envs per configuration, default 42] > profiles/r6/rodas5_contract.txt
"""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import numpy as np
import scenarios as SC
from oracle import oracle as O
from pcgym_amd.config import EnvSpec

B = int(sys.argv[1]) if len(sys.argv) > 1 else 42
N = 201
NT = min(8, os.cpu_count() or 1)


def registry_object(model, **over):
    """an object the way the reference's registry classes look to make_env (pcgym.py:150-153): class name + info()"""
    from pcgym_amd.models import get_model

    mi = get_model(model)
    info = {"parameters": {**mi.parameters, **over}, "states": list(mi.states), "inputs": list(mi.inputs),
            "disturbances": list(mi.disturbances)}
    return type(model, (), {"info": lambda self: info, "int_method": "hip"})()


def params(dt, **kw):
    p = copy.deepcopy(SC.scenarios()["me_dist_cons"]["env_params"])
    for k in ("constraints", "done_on_cons_vio", "r_penalty"):
        p.pop(k, None)
    k3 = N // 3
    p.update(N=N, tsim=N * dt, SP={"X5": [0.3] * k3 + [0.4] * k3 + [0.3] * (N - 2 * k3)},
             disturbances={"X0": np.full(N, 0.6), "Y6": np.full(N, 0.05)},
             disturbance_bounds={"low": np.array([0.5, 0.0]), "high": np.array([0.8, 0.1])},
             gaussian_disturbances={"X0": 0.02, "Y6": 0.01}, normalise_a=True, normalise_o=True)
    p.update(kw)
    return p


print(__doc__.split("\n\n")[0])
print(f"# {B} envs x {N - 1} steps per configuration; reference: dopri5 rtol = atol = 1e-13 from the same state; oracle threads {NT}")
print(f"{'eq_exponent':>11s} {'dt':>5s} {'rtol':>9s} {'env steps':>10s} {'attempts':>9s} {'worst units':>12s} {'99.9 %':>8s} {'99 %':>8s} {'median':>8s} {'worst rel':>10s}")
allu, t0 = [], time.time()
for expo in (1.5, 2.0, 3.0):
    for dt in (0.2, 1.0, 2.0, 5.0):
        cm = {} if expo == 2.0 else {"custom_model": registry_object("multistage_extraction", eq_exponent=expo)}
        s1 = EnvSpec(params(dt, **cm))  # the default plan: integrator and tolerance are the model's own
        s2 = EnvSpec(params(dt, integrator="dopri5", rtol=1e-13, atol=1e-13, **cm))
        assert s1.integrator == "rodas5", s1.integrator
        assert s1.model.parameters["eq_exponent"] == expo and s2.model.parameters["eq_exponent"] == expo
        e1 = O.OracleEnv(s1, B, seed=77, n_threads=NT)
        e2 = O.OracleEnv(s2, B, seed=77, n_threads=NT)
        e1.reset(), e2.reset()
        rng = np.random.default_rng(int(expo * 10 + dt * 100))
        units, rel, att = [], [], []
        for t in range(N - 1):
            a = rng.uniform(-1, 1, (s1.na, B))
            if t % 7 == 0:  # corners of the action box, regularly
                a[:, : min(4, B)] = np.array([[-1, 1, -1, 1], [-1, -1, 1, 1]])[:, : min(4, B)]
            e2.x[:] = e1.x
            e2.t = e1.t
            e1.step(a), e2.step(a)
            d = np.abs(e1.x - e2.x)
            units.append((d / (1e-6 * np.abs(e2.x) + 1e-8)).max(axis=0))
            rel.append((d / np.maximum(np.abs(e2.x), 1e-12)).max(axis=0))
            att.append(e1.nsteps.sum(axis=0))
        u, r, at = np.concatenate(units), np.concatenate(rel), np.concatenate(att)
        assert np.isfinite(u).all()
        allu.append(u)
        print(f"{expo:11.1f} {dt:5.1f} {s1.rtol:9.2e} {u.size:10d} {at.mean():9.2f} {u.max():12.3f} {np.quantile(u, 0.999):8.3f} "
              f"{np.quantile(u, 0.99):8.3f} {np.median(u):8.4f} {r.max():10.2e}", flush=True)
u = np.concatenate(allu)
print(f"\n# all {u.size} env steps: worst {u.max():.3f} units; histogram of the error in units of (1e-6 |x| + 1e-8):")
edges = [0, 0.01, 0.03, 0.1, 0.3, 0.5, 0.7, 1.0, 1.5, 2.0, 3.0, 1e9]
h, _ = np.histogram(u, bins=edges)
for lo, hi, n in zip(edges[:-1], edges[1:], h):
    print(f"  [{lo:5.2f}, {hi if hi < 1e8 else float('inf'):5.2f})  {n:8d}  {n / u.size:8.5f}")
print(f"# bar: worst <= 3 units -- {'MET' if u.max() <= 3.0 else 'NOT MET'}   ({time.time() - t0:.0f} s)")
