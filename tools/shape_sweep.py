#!/usr/bin/env python3
"""Every registry model x integrator: the other entry points of a plan against its own step() launches, on the GPU --
  rollout   pcg_rollout (T steps, state in registers) == T pcg_step launches, bitwise (state, observations, rewards, status);
  graph     a HIP graph of the T steps (pcg_graph_*) == the launches, bitwise;
  autoreset pcg_step_autoreset through an episode end == step + reset (state of the new episode, rewards, done).
Companion of tools/integrator_sweep.py (which holds the step kernels against the oracle).   usage: shape_sweep.py [integrator ...]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import scenarios as SC
from pcgym_amd import VecEnv

INTEGS = tuple(a for a in sys.argv[1:] if not a.startswith("-")) or ("rk4", "cv8", "dopri5", "tsit5", "rodas3", "rodas4", "rodas5")
S = SC.scenarios()
seen, bad, n = set(), 0, 0
B, T = 200, 5
TOL = 1e-9  # bitwise for most shapes; the fused rollout of some models contracts differently (reported as a number)


def close(a, b):
    """-> (ok, max relative difference): NaN / inf patterns must match, finite entries to TOL"""
    a, b = a.double(), b.double()
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    if not torch.equal(fa, fb):
        return False, float("inf")
    if not fa.any():
        return True, 0.0
    d = ((a[fa] - b[fa]).abs() / b[fa].abs().clamp_min(1e-9)).max().item()
    return d <= TOL, d

for name, sc in S.items():
    p0 = sc["env_params"]
    model = p0.get("model")
    if model is None or model in seen or p0.get("custom_model") is not None:
        continue
    seen.add(model)
    for integ in INTEGS:
        p = copy.deepcopy(p0)
        p.update(integrator=integ, rtol=1e-6, atol=1e-8, N=T + 2, tsim=float(p0["tsim"]) * (T + 2) / p0["N"])
        for k in ("uncertainty_percentages", "uncertainty_bounds", "distribution"):
            p.pop(k, None)
        if integ in ("rk4", "cv8"):
            p.pop("rtol"), p.pop("atol")
        for k in ("SP", "disturbances"):
            if p.get(k):
                p[k] = {kk: list(np.asarray(v, dtype=float)[: T + 2]) for kk, v in p[k].items()}
        try:
            envs = [VecEnv(copy.deepcopy(p), n_envs=B, seed=5) for _ in range(3)]
            ar = VecEnv(copy.deepcopy(p), n_envs=B, seed=5, auto_reset=True)
        except Exception as e:  # noqa: BLE001
            print(f"{model:32s} {integ:7s}: skipped ({type(e).__name__}: {str(e)[:60]})")
            continue
        spec = envs[0].spec
        gen = torch.Generator(device="cuda").manual_seed(7)
        acts = 2 * torch.rand((T + 4, spec.na, B), generator=gen, device="cuda", dtype=torch.float64) - 1
        if not spec.normalise_a:
            lo = torch.tensor(spec.a_low, device="cuda")[None, :, None]; hi = torch.tensor(spec.a_high, device="cuda")[None, :, None]
            acts = (acts + 1) * (hi - lo) / 2 + lo
        for e in envs + [ar]:
            e.reset()
        res = []
        e_step, e_roll, e_graph = envs
        obs_s, rew_s = [], []
        for i in range(T):
            o, r, d, _, _ = e_step.step(acts[i])
            obs_s.append(e_step.obs_soa.clone()); rew_s.append(r.clone())
        msg = []
        try:
            oq, rq = e_roll.rollout(acts[:T], collect_obs=True, collect_rew=True)
            cs = [close(e_roll.x, e_step.x)] + [close(rq[i], rew_s[i]) for i in range(T)] + [close(oq[i], obs_s[i]) for i in range(T)]
            ok = all(c[0] for c in cs) and torch.equal(e_roll.status, e_step.status)
            dm = max(c[1] for c in cs)
            msg.append("rollout " + ("ok" if ok else "DIFFERS") + (f" ({dm:.1e})" if dm > 0 else ""))
            bad += not ok
        except Exception as e:  # noqa: BLE001
            msg.append("rollout n/a")
        try:
            g = e_graph.capture_steps([acts[i] for i in range(T)])
            g.replay()
            torch.cuda.synchronize()
            cs = [close(e_graph.x, e_step.x), close(e_graph.rew, rew_s[-1])]
            ok = all(c[0] and c[1] == 0.0 for c in cs) and torch.equal(e_graph.status, e_step.status)  # the same kernels: bitwise
            msg.append("graph " + ("ok" if ok else "DIFFERS"))
            bad += not ok
        except Exception as e:  # noqa: BLE001
            msg.append(f"graph n/a ({type(e).__name__})")
        # auto-reset through the episode end (N - 1 = T + 1 steps), then one step of the next episode
        ok = True
        ref = VecEnv(copy.deepcopy(p), n_envs=B, seed=5)
        ref.reset()
        for i in range(T + 2):
            o, r, d, _, _ = ar.step(acts[i])
            if ref.t == ref.N - 1:
                ref.reset()  # the next episode: the episode counter keys the reset stream on both sides
            o2, r2, d2, _, _ = ref.step(acts[i])
            same_x = close(ar.x, ref.x)[1] == 0.0 if ref.t != ref.N - 1 else True  # (at the episode end `ar` already holds the NEW x0)
            ok = ok and close(r, r2)[1] == 0.0 and torch.equal(d, d2) and same_x
        msg.append("autoreset " + ("ok" if ok else "DIFFERS"))
        bad += not ok
        n += 1
        print(f"{model:32s} nx {spec.nx:2d} {integ:7s}: " + ", ".join(msg), flush=True)
        for e in envs + [ar, ref]:
            e.close()
print("combinations:", n, "differences:", bad)
sys.exit(1 if bad else 0)
