#!/usr/bin/env python3
"""Secondary measurements: BASELINE.json configs[2..4] on ONE MI355X (the headline config is bench.py).

  config 3  multistage_extraction (5-stage nx=10) and the reactive 20-state variant, B = 262,144,
            adaptive DOPRI5 rtol = atol = 1e-8, (L,G) per env over the full action box (stiff)
  config 4  crystallization nx=7, B = 262,144, RK4 n_sub = 32 per dt = 1, a_delta on
  config 5  mixed {cstr, four_tank, ME} with set-point changes + Gaussian disturbances, one shard
            (3 plans on 3 streams; the 8-GPU run replicates this shard per rank)
  extra     four_tank B = 2^20 RK4 (HBM-bound), fused rollout (pcg_rollout) on the cstr workload

Prints one JSON object per line.  These kernels are fp64-VALU-bound (except four_tank / cstr): the
figure of merit is RHS evaluations/s and an algorithmic fp64 FLOP/s against the 78.6 TFLOP/s vector
peak, with HBM GB/s for completeness.  Flop counts per RHS evaluation are the SURVEY.md section 8a ones.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np
import torch

import scenarios as SC
from pcgym_amd import VecEnv

FP64_PEAK_TFLOPS = 78.6
FLOP_PER_RHS = {"cstr": 16 + 25 + 10, "four_tank": 20 + 4 * 12, "multistage_extraction": 65,
                "multistage_extraction_reactive": 150, "crystallization": 60 + 4 * 25 + 12 + 6 * 10}  # 2 log + 2 exp + sqrt + 6 div


def timed_steps(env, acts, K, W=5):
    dev = env.device
    for i in range(W):
        env.step(acts[i % len(acts)])
        if env.t == env.N - 1:
            env.reset()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(K):
        env.step(acts[i % len(acts)])
        if env.t == env.N - 1:
            env.reset()
    e1.record()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, e0.elapsed_time(e1) * 1e-3


def report(name, env, wall, K, extra=None):
    B = env.B
    steps_s = B * K / wall
    d = {"config": name, "model": env.spec.model.name, "B": B, "steps": K, "integrator": env.spec.integrator,
         "env_steps_per_s": steps_s, "ms_per_step": wall / K * 1e3,
         "alg_bytes_per_env_step": env.bytes_per_env_step,
         "hbm_GBps": env.bytes_per_env_step * steps_s / 1e9}
    if extra:
        d.update(extra)
    print(json.dumps(d), flush=True)
    return d


def me_like(name, B, lds_stages=False, K=20):
    sc = SC.scenarios()[name]
    p = dict(sc["env_params"])
    p.update(integrator="dopri5", rtol=1e-8, atol=1e-8)
    env = VecEnv(p, n_envs=B, seed=3, lds_stages=lds_stages)
    env.reset()
    gen = torch.Generator(device=env.device).manual_seed(7)
    x0 = env.x * (1 + 0.05 * (2 * torch.rand(env.x.shape, generator=gen, device=env.device, dtype=torch.float64) - 1))
    env.x.copy_(x0)
    acts = [2 * torch.rand((env.spec.na, B), generator=gen, device=env.device, dtype=torch.float64) - 1 for _ in range(8)]
    wall, _ = timed_steps(env, acts, K, W=3)
    ns = env.nsteps.to(torch.float64)
    acc, rej = ns[0].mean().item(), ns[1].mean().item()
    rhs_per_step = 2 + 6 * (acc + rej) + acc * 0  # FSAL: 6 new evaluations per attempted step (+2 for h0)
    rhs_s = rhs_per_step * B * K / wall
    fl = FLOP_PER_RHS[env.spec.model.name] + 2 * 6 * env.spec.nx  # + RK combination per stage
    report(("config3 " if "reactive" not in name else "config3' ") + name + (" lds-stages" if lds_stages else ""), env, wall, K,
           {"accepted_steps_mean": acc, "rejected_steps_mean": rej, "rhs_evals_per_s": rhs_s,
            "alg_fp64_TFLOPs": rhs_s * fl / 1e12, "frac_fp64_vector_peak": rhs_s * fl / 1e12 / FP64_PEAK_TFLOPS,
            "max_accepted": int(ns[0].max().item()), "finite": bool(torch.isfinite(env.x).all().item())})
    env.close()


def cryst(B, K=20):
    sc = SC.scenarios()["cryst_adelta"]
    p = dict(sc["env_params"])
    p.update(integrator="rk4", substeps=32)
    env = VecEnv(p, n_envs=B, seed=3)
    env.reset()
    gen = torch.Generator(device=env.device).manual_seed(7)
    x = env.x.clone()
    x[:5] *= 1 + 0.01 * (2 * torch.rand((5, B), generator=gen, device=env.device, dtype=torch.float64) - 1)
    x[5] = torch.sqrt(x[2] * x[0] / x[1] ** 2 - 1)
    x[6] = x[1] / x[0]
    env.x.copy_(x)
    acts = [0.3 * (2 * torch.rand((1, B), generator=gen, device=env.device, dtype=torch.float64) - 1) - 0.2 for _ in range(8)]
    wall, _ = timed_steps(env, acts, K, W=3)
    rhs_s = 4 * 32 * B * K / wall
    fl = FLOP_PER_RHS["crystallization"] + 2 * 7
    report("config4 crystallization rk4 n_sub=32 a_delta", env, wall, K,
           {"rhs_evals_per_s": rhs_s, "alg_fp64_TFLOPs": rhs_s * fl / 1e12,
            "frac_fp64_vector_peak": rhs_s * fl / 1e12 / FP64_PEAK_TFLOPS,
            "finite": bool(torch.isfinite(env.x).all().item())})
    env.close()


def four_tank(B, K=200):
    p = dict(SC.scenarios()["four_tank_canonical"]["env_params"])
    env = VecEnv(p, n_envs=B, seed=3)
    env.reset()
    gen = torch.Generator(device=env.device).manual_seed(7)
    # pump voltages in the upper 3/4 of the action box: with the full box ~0.05 % of the envs drain tank 3
    # and sqrt(2 g h) of a negative level is NaN -- in the reference just the same
    acts = [1.5 * torch.rand((2, B), generator=gen, device=env.device, dtype=torch.float64) - 0.5 for _ in range(16)]
    wall, _ = timed_steps(env, acts, K)
    report("four_tank rk4 n_sub=4 (canonical dt=1000/60)", env, wall, K, {"finite": bool(torch.isfinite(env.x).all().item())})
    env.close()


def mixed(B_total, K=60):
    """config 5, one shard: ceil(B/3) envs of each of cstr / four_tank / ME, each on its own stream."""
    Bm = B_total // 3
    specs = []
    p = dict(SC.scenarios()["cstr_dist_Ti"]["env_params"])
    p.update(gaussian_disturbances={"Ti": 2.0})
    specs.append(p)
    specs.append(dict(SC.scenarios()["four_tank_canonical"]["env_params"]))
    p = dict(SC.scenarios()["me_dist_cons"]["env_params"])
    p.pop("constraints"), p.pop("done_on_cons_vio"), p.pop("r_penalty")
    p.update(gaussian_disturbances={"X0": 0.02}, normalise_a=True, normalise_o=True)
    specs.append(p)
    from pcgym_amd import MixedVecEnv

    mixed_env = MixedVecEnv([(p, Bm) for p in specs], seed=5)
    envs = mixed_env.envs
    mixed_env.reset()
    gen = torch.Generator(device="cuda").manual_seed(11)
    acts = []
    for e in envs:
        a = 2 * torch.rand((4, e.spec.na, Bm), generator=gen, device="cuda", dtype=torch.float64) - 1
        if e.spec.model.name.startswith("multistage"):
            a = a * 0.2 - 0.7  # moderate flows (|lambda| dt of a few tens): the paper's operating range
        acts.append(a)
    torch.cuda.synchronize()

    def one_round(i):
        mixed_env.step([a[i % 4] for a in acts])
        if envs[0].t == envs[0].N - 1:  # the three configurations share N = 60
            mixed_env.reset()

    for i in range(3):
        one_round(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        one_round(i)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    by = sum(e.bytes_per_env_step * e.B for e in envs)
    print(json.dumps({"config": "config5 mixed {cstr+Ti~N, four_tank, ME+X0~N} one shard, 3 streams",
                      "B": 3 * Bm, "steps": K, "env_steps_per_s": 3 * Bm * K / wall, "ms_per_step": wall / K * 1e3,
                      "hbm_GBps": by * K / wall / 1e9,
                      "finite": all(bool(torch.isfinite(e.x).all().item()) for e in envs)}), flush=True)
    for e in envs:
        e.close()


def fused_rollout(B, T=59, reps=10):
    import bench

    p = bench.workload_params(B)
    env = VecEnv(p, n_envs=B, seed=1)
    gen = torch.Generator(device=env.device).manual_seed(7)
    acts = 2 * torch.rand((T, 1, B), generator=gen, device=env.device, dtype=torch.float64) - 1
    for collect in (False, True):
        env.reset()
        env.rollout(acts, collect_obs=collect, collect_rew=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            env.reset()
            env.rollout(acts, collect_obs=collect, collect_rew=True)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        bpe = 8 * (1 + 1) + (8 * 3 if collect else 0)  # per env-step: action in, reward out (+ obs out); x stays in registers
        print(json.dumps({"config": "fused open-loop rollout (pcg_rollout), cstr bench workload, obs %s" %
                          ("stored every step" if collect else "of the last step only"),
                          "B": B, "T": T, "env_steps_per_s": B * T * reps / wall, "us_per_env_step_launch_equiv": wall / (T * reps) * 1e6,
                          "alg_bytes_per_env_step": bpe, "hbm_GBps": bpe * B * T * reps / wall / 1e9}), flush=True)
    env.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["me", "mer", "cryst", "four_tank", "mixed", "rollout"]
    B18 = 1 << 18
    if "me" in which:
        me_like("me_canonical", B18)
    if "mer" in which:
        me_like("me_reactive", B18)
        me_like("me_reactive", B18, lds_stages=True)
    if "cryst" in which:
        cryst(B18)
    if "four_tank" in which:
        four_tank(1 << 20)
    if "mixed" in which:
        mixed(1 << 20)
    if "rollout" in which:
        fused_rollout(1 << 20)
