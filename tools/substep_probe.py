"""Lean pipelined kernel (two envs per lane) vs the classic one-env-per-lane kernel as the RK4 sub-step count grows.
    python tools/substep_probe.py        (needs a GPU)"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import bench as BN  # noqa: E402
import scenarios as SC  # noqa: E402
from tools.user_model_probe import run  # noqa: E402


def main():
    B = 1 << 20
    for name, base in (("cstr", BN.workload_params()), ("four_tank", copy.deepcopy(SC.scenarios()["four_tank_canonical"]["env_params"]))):
        for n in (1, 2, 4, 8, 16):
            p = copy.deepcopy(base)
            p.update(integrator="rk4", substeps=n)
            p.pop("noise", None), p.pop("noise_percentage", None)
            t_lean, _ = run(copy.deepcopy(p), B)
            t_cls, _ = run(copy.deepcopy(p), B, variant=1)
            print("%-10s substeps %2d  default (pipelined, 2 envs/lane) %.1f us | classic %.1f us" % (name, n, t_lean, t_cls))


if __name__ == "__main__":
    main()
