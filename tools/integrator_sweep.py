#!/usr/bin/env python3
"""Every registry scenario x {rodas3, rodas4, rodas5} x {lock-stepped, per-env t}: three env steps through the general step
kernel against the oracle (one-step comparisons from a common state).  Found in round 5: the 24-state model's lock-stepped
Rosenbrock kernels (1400 spilled SGPRs, all 512 vector registers in use) returned a garbage x[2] (or x[0] with scalar spills
in memory; correct at -O1) -- the attempt of models with more than 16 states is now written with loops that stay loops
(pcg_integrators.hpp: ros_try_rolled), and all 114 combinations agree with the oracle."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import scenarios as SC
from oracle import oracle as O
from pcgym_amd import VecEnv

# usage: integrator_sweep.py [integrator ...]   (default: the three Rosenbrock integrators; any of config.INTEGRATOR_IDS)
INTEGS = tuple(a for a in sys.argv[1:] if not a.startswith("-")) or ("rodas3", "rodas4", "rodas5")
VARIANT = 1 if "--classic" in sys.argv else None  # --classic: PCG_OPT_VARIANT 1 (the one-env-per-lane kernels)
S = SC.scenarios()
seen, bad = set(), 0
for name, sc in S.items():
    p0 = sc["env_params"]
    model = p0.get("model")
    if model is None:
        continue
    if model in seen or p0.get("custom_model") is not None:
        continue
    seen.add(model)
    for integ in INTEGS:
        for pe in (False, True):
            p = copy.deepcopy(p0)
            p.update(integrator=integ, rtol=1e-6, atol=1e-8)
            if integ in ("rk4", "cv8", "rk4g", "tsit5g"):
                p.pop("rtol"), p.pop("atol")
            for k in ("uncertainty_percentages", "uncertainty_bounds", "distribution"):
                p.pop(k, None)
            try:
                env = VecEnv(copy.deepcopy(p), n_envs=130, seed=3, per_env_t=pe, **({"variant": VARIANT} if VARIANT else {}))
            except Exception as e:  # noqa: BLE001
                print(f"{model:32s} {integ} per_env_t {pe}: skipped ({type(e).__name__})")
                continue
            spec = env.spec
            orc = O.OracleEnv(spec, 130, seed=3, per_env_t=pe)
            env.reset(); orc.reset()
            rng = np.random.default_rng(1)
            worst, same = 0.0, 1.0
            for i in range(3):
                a = rng.uniform(-1, 1, (spec.na, 130))
                if not spec.normalise_a:
                    a = (a + 1) * (spec.a_high - spec.a_low)[:, None] / 2 + spec.a_low[:, None]
                env.step(torch.tensor(a, device=env.device)); orc.step(a)
                xg = env.x.cpu().numpy()
                ok = np.isfinite(orc.x).all(axis=0)
                xs = np.maximum(np.abs(orc.x[:, ok]), 1e-6 * np.max(np.abs(orc.x[:, ok]), axis=1, keepdims=True))
                worst = max(worst, float(np.max(np.abs(xg[:, ok] - orc.x[:, ok]) / xs)))
                if env.nsteps is not None and orc.nsteps is not None:
                    same = min(same, float(np.mean(np.all(env.nsteps.cpu().numpy() == orc.nsteps, axis=0))))
                env.x.copy_(torch.tensor(orc.x, device=env.device))
            flag = "" if worst <= 1e-4 else "   <-- BAD"
            bad += worst > 1e-4
            print(f"{model:32s} nx {spec.nx:2d} {integ} per_env_t {str(pe):5s}: worst rel diff {worst:.2e}, identical step sequences {same:.3f}{flag}", flush=True)
            env.close()
print("bad combinations:", bad)
sys.exit(1 if bad else 0)
