"""Default-path cost of the registry models that have a recorded scenario: us per step of B envs at each model's default
integrator settings, through the default dispatch, the classic one-env-per-lane kernel (PCG_OPT_VARIANT 1) and -- for
adaptive plans -- the work-queue kernel forced on (PCG_Q_FORCE).  Guards against a routing that is slower than the
plain kernel (round 2 found one: cstr DOPRI5 through the queue).

    python tools/registry_sweep.py [B]        (needs a GPU)
"""
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

CASES = ["cstr_canonical", "cstr_dist_both", "cstr_cons_pen_norm", "four_tank_canonical", "me_canonical", "me_dist_cons",
         "me_reactive", "cryst_adelta", "complex_cstr_sp", "photo_batch_reward", "distillation_sp", "first_order_sp",
         "biofilm_sp", "heat_exchanger_sp"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
    S = SC.scenarios()
    print("B = %d; us per step: default dispatch | classic kernel | work queue forced" % B)
    for name in CASES:
        p = copy.deepcopy(S[name]["env_params"])
        p.pop("noise", None), p.pop("noise_percentage", None)
        from pcgym_amd.config import EnvSpec

        spec = EnvSpec(copy.deepcopy(p))
        steps = min(spec.N - 1, 40)
        os.environ.pop("PCG_Q_FORCE", None)
        t_def, _ = run(copy.deepcopy(p), B, steps=steps, reps=3)
        t_cls, _ = run(copy.deepcopy(p), B, variant=1, steps=steps, reps=3)
        t_q = float("nan")
        if spec.integrator == "dopri5":
            os.environ["PCG_Q_FORCE"] = "1"
            t_q, _ = run(copy.deepcopy(p), B, steps=steps, reps=3)
            os.environ.pop("PCG_Q_FORCE", None)
        flag = "  <-- default slower than classic" if t_def > 1.08 * t_cls else ""
        print("%-22s %-30s %-7s %9.1f | %9.1f | %9.1f%s" % (name, spec.model.name, spec.integrator, t_def, t_cls, t_q, flag))


if __name__ == "__main__":
    main()
