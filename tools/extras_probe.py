#!/usr/bin/env python3
"""cstr at B = 2^20, one RK4 step per env step: lean path vs the feature-masked pipelined kernels (pcg_step_feat.hpp) and
the classic one-env-per-lane kernel (PCG_OPT_VARIANT 1) with noise / constraints / tracking reward / Gaussian
disturbance / per-env t switched on.  Run on the GPU box.  PROBE_STATUS=0 drops the status byte."""
import os, sys, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as BN
from pcgym_amd import VecEnv

def run(label, extra, per_env_t=False, variant=None, status=None):
    B = 1 << 20
    p = BN.workload_params(B); p.update(extra)
    if variant is None:
        run(label + " [classic]", extra, per_env_t, variant=1)
    env = VecEnv(p, n_envs=B, seed=1, per_env_t=per_env_t, auto_reset=True, variant=variant or 0,
                 track_status=bool(int(os.environ.get("PROBE_STATUS", "1"))) if status is None else status); env.reset()
    acts = 2 * torch.rand((16, 1, B), device=env.device, dtype=torch.float64) - 1
    W, K = int(os.environ.get("PROBE_WARM", 600)), int(os.environ.get("PROBE_STEPS", 3000))
    for i in range(W): env.step(acts[i % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): env.step(acts[i % 16])
    torch.cuda.synchronize(); w = time.perf_counter() - t0
    bpe = env.bytes_per_env_step
    print(f"{label:34s} {w/K*1e6:7.2f} us/step  {B*K/w:.3e} env-steps/s  {bpe} B/env-step -> {bpe*B*K/w/1e12:.2f} TB/s")

def cons(x, u):
    return np.array([x[1] - 340.0, 300.0 - x[1]])

run("lean, no status byte (pipe kernel)", {}, variant=0, status=False)
run("lean + status byte (feat mask 0)", {})
run("noise 0.1 %", {"noise": True, "noise_percentage": 0.001})
run("constraints (2 rows)", {"constraints": cons, "done_on_cons_vio": False, "r_penalty": True})
run("tracking reward (sp_track)", {"custom_reward": {"kind": "sp_track", "R": 0.1}})
run("per-env t", {}, per_env_t=True)
import numpy as _np
run("gaussian disturbance Ti", {"disturbances": {"Ti": _np.full(60, 350.0)}, "disturbance_bounds": {"low": _np.array([320.0]),
    "high": _np.array([360.0])}, "gaussian_disturbances": {"Ti": 2.0},
    "o_space": {"low": _np.array([0.7, 300.0, 0.8]), "high": _np.array([1.0, 350.0, 0.9])}})
run("noise + tracking reward", {"noise": True, "noise_percentage": 0.001, "custom_reward": {"kind": "sp_track", "R": 0.1}})
run("noise + constraints + per-env t", {"noise": True, "noise_percentage": 0.001, "constraints": cons,
    "done_on_cons_vio": True, "r_penalty": True}, per_env_t=True)
