#!/usr/bin/env python3
"""Which lanes and which components does the (unrolled) lock-stepped Rosenbrock step kernel of the 24-state model get wrong,
and what does the wrong value look like?  One env step of heat_exchanger under rodas4, lock-stepped and per-env counters,
against the oracle (PCGYM_HIP_LIB selects the library build)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import test_gpu_sweeps as T
from oracle import oracle as O
from pcgym_amd import VecEnv

for integ in ("rodas4", "rodas5"):
    for pe in (False, True):
        B = 256
        p = T._params("heat_exchanger", integ, "lean")
        env = VecEnv(copy.deepcopy(p), n_envs=B, seed=3, per_env_t=pe, variant=1)
        orc = O.OracleEnv(env.spec, B, seed=3, per_env_t=pe)
        env.reset(), orc.reset()
        a = T._actions(env.spec, np.random.default_rng(1), B)
        env.step(torch.tensor(a, device=env.device)), orc.step(a)
        xg, xo = env.x.cpu().numpy(), orc.x
        rel = np.abs(xg - xo) / np.maximum(np.abs(xo), 1e-9)
        badc = np.where(rel.max(axis=1) > 1e-6)[0]
        badl = np.where(rel.max(axis=0) > 1e-6)[0]
        same_steps = float(np.mean(np.all(env.nsteps.cpu().numpy() == orc.nsteps, axis=0)))
        print(f"{integ} per_env_t={pe}: wrong components {badc.tolist()}, wrong lanes {len(badl)} of {B}"
              f" (lane mod 32: {sorted(set((badl % 32).tolist()))[:40]}), identical step sequences {same_steps:.3f}")
        if len(badl):
            l = badl[0]
            c = badc[0]
            print(f"   lane {l} component {c}: got {xg[c, l]!r} want {xo[c, l]!r}; got as bits {np.float64(xg[c, l]).view(np.uint64):#018x};"
                  f" the lane's other components near the wrong value: {[i for i in range(xg.shape[0]) if abs(xg[i, l] - xg[c, l]) < 1e-9 and i != c]}")
        env.close()
