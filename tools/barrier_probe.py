#!/usr/bin/env python3
"""What the per-step barrier costs the chain-bound plans (VERDICT r5 "next" 3): over one episode of the bench's workload,
on the CPU oracle (bit-identical step sequences to the kernels),

    sum_t max_i attempts(i, t)      what a launch per step pays: every step waits for its heaviest env
    max_i sum_t attempts(i, t)      what a barrier-free episode would pay: the heaviest env's own chain

and the same at the granularity the hardware actually synchronises on -- a wave of 64 lanes (one env per lane for the whole
episode: the fused rollout) and a workgroup tile.   usage: barrier_probe.py [cstr_safe|me10_ros5|...] [log2 B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import numpy as np
import bench
from oracle import oracle as O
from pcgym_amd.config import EnvSpec

wl = sys.argv[1] if len(sys.argv) > 1 else "cstr_safe"
B = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 16)
name, params, _, _, _ = bench.single_workload(wl)
spec = EnvSpec(params)
env = O.OracleEnv(spec, B, seed=1234, n_threads=min(8, os.cpu_count() or 1))
env.reset()
rng = np.random.default_rng(0)
T = spec.N - 1
att = np.zeros((T, B), dtype=np.int64)
for t in range(T):
    a = bench.act_box(spec) * rng.uniform(-1, 1, (spec.na, B)) + bench.act_shift(spec)
    env.step(a)
    att[t] = env.nsteps.sum(axis=0)
    if wl.startswith("cstr") and spec.integrator in ("rk4g", "tsit5g"):
        pass
print(f"# {name}: B = {B}, T = {T} steps, integrator {spec.integrator}; attempts of the adaptive pair per env step "
      f"(guarded plans: 0 for an env the fixed step was trusted for)")
print(f"mean attempts per env step {att.mean():.2f}; envs with any adaptive attempt: {np.mean(att.sum(axis=0) > 0):.3f}")
per_step_max = att.max(axis=1)
print(f"sum_t max_i  = {per_step_max.sum():6d}   (per step: min {per_step_max.min()}, median {int(np.median(per_step_max))}, max {per_step_max.max()})")
print(f"max_i sum_t  = {att.sum(axis=0).max():6d}   ratio {per_step_max.sum() / max(att.sum(axis=0).max(), 1):.2f}")
print(f"mean_i sum_t = {att.sum(axis=0).mean():8.1f}")
for w in (64, 256, 1024):
    g = att[:, : B // w * w].reshape(T, -1, w)
    lock = g.max(axis=2).sum(axis=0)          # a group that steps in lock-step pays the per-step max of its members
    free = g.sum(axis=0).max(axis=1)          # its heaviest member's own chain
    bal = g.sum(axis=(0, 2)) / w              # perfectly balanced within the group
    print(f"groups of {w:5d}: lock-stepped group cost mean {lock.mean():7.1f} max {lock.max():5d} | heaviest member mean {free.mean():7.1f} max {free.max():5d}"
          f" | balanced mean {bal.mean():7.1f} max {bal.max():7.1f}")

# --- would SORTING the envs by how hot they are rescue a wave that steps its 64 lanes together? --------------------------------
# (the fused rollout as it exists: one env per lane for the whole episode, the wave pays the per-step maximum of its lanes)
tot = att.sum(axis=0)
for name_, order in (("as drawn", np.arange(B)), ("sorted by the episode's total (hindsight)", np.argsort(tot)),
                     ("sorted by the first 5 steps' attempts", np.argsort(att[:5].sum(axis=0), kind="stable"))):
    g = att[:, order][:, : B // 64 * 64].reshape(T, -1, 64)
    lock = g.max(axis=2).sum(axis=0)
    print(f"waves of 64, envs {name_:45s}: wave cost mean {lock.mean():7.1f} (balanced {att.mean() * T:6.1f}), "
          f"sum over waves / balanced = {lock.sum() * 64 / max(att.sum(), 1):.2f}")
# persistence: does an env that needed the adaptive pair at step t need it at t + 1?
hot = att > 0
print(f"P(hot at t+1 | hot at t) = {(hot[1:] & hot[:-1]).sum() / max(hot[:-1].sum(), 1):.3f},  P(hot at t+1 | calm at t) = "
      f"{(hot[1:] & ~hot[:-1]).sum() / max((~hot[:-1]).sum(), 1):.3f},  share of hot env steps {hot.mean():.3f}")
