"""The default cstr plan (round 3: guarded Tsit5 x 2 -- first guarded RK4 x 5 -- with the adaptive pair at 1e-10 as fallback; round 2: the adaptive pair
for every env) against the explicit opt-ins, B = 2^20 (needs a GPU).  Canonical closed loop: x0 = (0.8, 330 K), random
jacket temperatures, 59 steps."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402,F401

import scenarios as SC  # noqa: E402
from tools.user_model_probe import run  # noqa: E402

B = 1 << 20
base = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
base.pop("noise", None), base.pop("noise_percentage", None)
for tag, kw in (("default (tsit5g: guarded Tsit5 x2, fallback dopri5 1e-10)", {}), ("rk4g (guarded RK4 x5, fallback dopri5 1e-10)", dict(integrator="rk4g")), ("round-2 default (dopri5, rtol = atol = 1e-10)", dict(integrator="dopri5")), ("dopri5 1e-8 (the reference's jax path)", dict(integrator="dopri5", rtol=1e-8, atol=1e-8)),
                ("rk4 x4 (opt-in, canonical closed loop)", dict(integrator="rk4")), ("rk4 x1", dict(integrator="rk4", substeps=1))):
    p = copy.deepcopy(base)
    p.update(kw)
    t, _ = run(p, B, steps=59, reps=4)
    print("%-45s %.1f us per step of 2^20 envs = %.2e env-steps/s" % (tag, t, B / t * 1e6))
