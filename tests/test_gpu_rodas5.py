"""GPU: the fifth-order stiff pair (PCG_INT_RODAS5) through the C ABI against its oracle twin -- the cases of
tests/test_gpu_rodas4.py under the other tableau.

Dense path (any model): identical step sequences for >= 90 % of the samples, states to the conditioning of W.  Structured path
(the 10-state extraction cascade, registers only; classic and work-queue kernels, both counter modes, fused auto-reset, fused
rollout, the cooperative rule switched on): identical step sequences for EVERY env and states to round-off, also at BASELINE
configs[2]'s full size, where the accuracy against a 1e-13 solve is checked too."""
import copy

import numpy as np
import pytest

import helpers as H
import scenarios as SC
from test_gpu_rodas4 import _step_pair, _torch

pytestmark = pytest.mark.gpu

ROS5 = dict(integrator="rodas5", rtol=1e-6, atol=1e-8)
INT_CASES = [
    ("cstr", "cstr", ROS5, 5e-8),
    ("four_tank", "four_tank", ROS5, 5e-8),
    ("crystallization", "crystallization", ROS5, 5e-8),
    ("distillation_column", "distillation_column", ROS5, 5e-8),
    ("multistage_extraction_reactive", "multistage_extraction_reactive", ROS5, 5e-8),
    ("heat_exchanger", "heat_exchanger", ROS5, 5e-8),
    # structured W in registers: bit-exact class
    ("multistage_extraction", "multistage_extraction", ROS5, 1e-11),
    ("multistage_extraction_d", "multistage_extraction", dict(integrator="rodas5", rtol=8e-8, atol=8e-8), 1e-11),
]


@pytest.mark.parametrize("fix,model,kw,tol", INT_CASES)
def test_integrate_vs_oracle(fix, model, kw, tol):
    torch = _torch()
    from oracle import oracle as O
    from test_gpu_parity import _plan_for
    from test_oracle_golden import _spec_for_integration

    g = H.gold("tight_" + fix)
    spec = _spec_for_integration(model, float(g["dt"]), g["u"].shape[1], **kw)
    lib, plan = _plan_for(spec, torch)
    xs, us = g["x"].T.copy(), g["u"].T.copy()
    x = torch.tensor(xs, device="cuda")
    u = torch.tensor(us, device="cuda")
    ns = torch.zeros((2, x.shape[1]), dtype=torch.int32, device="cuda")
    assert lib.pcg_integrate(plan, x.shape[1], x.data_ptr(), u.data_ptr(), ns.data_ptr(), None) == 0
    torch.cuda.synchronize()
    lib.pcg_plan_destroy(plan)
    got = x.cpu().numpy()
    want, ns_o = O.integrate(spec, xs, us)
    if model == "multistage_extraction":
        H.adaptive_check(model, got, want, ns.cpu().numpy(), ns_o, fix, tol=tol)
    else:  # (dense path: see tests/test_gpu_rodas4.py)
        xs_ = np.maximum(np.abs(want), 1e-6 * np.max(np.abs(want), axis=1, keepdims=True))
        ex = np.max(np.abs(got - want) / xs_, axis=0)
        same = np.all(ns.cpu().numpy() == ns_o, axis=0)
        assert same.mean() >= 0.9, (fix, same.mean())
        assert ex[same].max() <= tol, (fix, ex[same].max())
        assert ex.max() <= 100 * kw["rtol"], (fix, ex.max())
    t = g["xf"].T
    assert np.all(np.abs(got - t) <= 3e-4 * np.abs(t) + 1e-6)


@pytest.mark.parametrize("name", ["me_canonical", "me_dist_cons"])
@pytest.mark.parametrize("per_env_t", [False, True])
@pytest.mark.parametrize("kernel", ["queue", "classic"])
@pytest.mark.parametrize("coop", [False, True])
def test_step_vs_oracle_structured(name, per_env_t, kernel, coop, monkeypatch):
    """full step tuples of the extraction scenarios, 12 steps WITHOUT re-synchronisation, through the work-queue kernel (forced:
    thin tiles too) and the classic kernel; with the cooperative rule on (threshold 30: SEULEX-8 takes the heavy envs, eight
    lanes each in the queue kernel) and off (the default under this pair)"""
    B = 1500 if kernel == "queue" else 700
    kw = dict(per_env_t=per_env_t)
    if kernel == "classic":
        kw["variant"] = 1
    params = dict(integrator="rodas5")
    if coop:
        params["cooperative"] = {"thr": 30}
    torch, env, orc = _step_pair(name, B, 11, monkeypatch, kernel == "queue", params=params, **kw)
    assert env.spec.integrator == "rodas5" and (env.spec.coop_thr == 30.0) == coop
    rng = np.random.default_rng(3)
    for i in range(12):
        a = rng.uniform(-1, 1, (env.spec.na, B))
        if not env.spec.normalise_a:
            a = (a + 1) * (env.spec.a_high - env.spec.a_low)[:, None] / 2 + env.spec.a_low[:, None]
        o, r, d, _, _ = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
        H.adaptive_check("multistage_extraction", env.x.cpu().numpy(), orc.x, env.nsteps.cpu().numpy(), orc.nsteps,
                         (name, kernel, per_env_t, coop, i), tol=1e-11)
        assert np.max(np.abs(env.obs_soa.cpu().numpy() - orc.obs) / np.maximum(np.abs(orc.obs), 1e-3)) <= 1e-10
        assert np.allclose(r.cpu().numpy(), rc, rtol=1e-9, atol=1e-10)
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), dc) and not env.status.any()
        if env.spec.ncon:
            assert np.array_equal(env.viol.cpu().numpy(), orc.viol)
    env.close()


def test_queue_equals_classic_and_is_order_independent(monkeypatch):
    torch = _torch()
    from pcgym_amd import VecEnv

    monkeypatch.setenv("PCG_Q_FORCE", "1")
    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p.update(integrator="rodas5")
    B = 30000
    q, cl, q2 = VecEnv(p, n_envs=B, seed=1), VecEnv(p, n_envs=B, seed=1, variant=1), VecEnv(p, n_envs=B, seed=1)
    gen = torch.Generator(device="cuda").manual_seed(2)
    for e in (q, cl, q2):
        e.reset()
    x0 = q.x * (1 + 0.05 * (2 * torch.rand(q.x.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
    perm = torch.randperm(B, generator=gen, device="cuda")
    q.x.copy_(x0)
    cl.x.copy_(x0)
    q2.x.copy_(x0[:, perm])
    for i in range(3):
        a = 2 * torch.rand((2, B), generator=gen, device="cuda", dtype=torch.float64) - 1
        q.step(a)
        cl.step(a)
        q2.step(a[:, perm])
        assert torch.equal(q.x, cl.x) and torch.equal(q.nsteps, cl.nsteps) and torch.equal(q.rew, cl.rew), i
        assert torch.equal(q.x[:, perm], q2.x) and torch.equal(q.nsteps[:, perm], q2.nsteps), i
    for e in (q, cl, q2):
        e.close()


def test_autoreset_in_the_same_launch(monkeypatch):
    torch, env, orc = _step_pair("me_canonical", 900, 70, monkeypatch, True, auto_reset=True,
                                 params=dict(integrator="rodas5", N=7, tsim=7.0, SP={"X5": [0.3] * 7}))
    N = env.N
    for i in range(2 * (N - 1) + 2):
        a = np.random.default_rng(i).uniform(-1, 1, (2, env.B))
        o, r, d, _, _ = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
        rc, dc = rc.copy(), dc.copy()
        if orc.t == N - 1:
            orc.reset()
        assert np.allclose(r.cpu().numpy(), rc, rtol=1e-9, atol=1e-10) and np.array_equal(d.cpu().numpy().astype(np.uint8), dc)
        assert np.max(np.abs(env.x.cpu().numpy() - orc.x) / np.maximum(np.abs(orc.x), 1e-6)) <= 1e-11, i
        assert env.t == orc.t
    env.close()


def test_full_size_configs2_rodas5():
    """BASELINE configs[2] at its size (B = 262,144) under the model's default plan: oracle agreement on a slice of 2048 envs
    (identical step sequences, round-off), the slice within 1e-6 of a 1e-13 solve of the same steps, the attempts per env
    step, and the steady-state solute balance after holding the input."""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd.config import EnvSpec

    B = 1 << 18
    gen = torch.Generator(device="cuda").manual_seed(99)
    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p.update(integrator="rodas5", N=200, tsim=200.0, SP={"X5": [0.3] * 200})
    env = VecEnv(p, n_envs=B)
    assert env.spec.rtol == 8e-8 and env.spec.ep_kmax == 16 and env.spec.coop_thr == 0.0
    env.reset()
    x0 = env.x * (1 + 0.05 * (2 * torch.rand(env.x.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
    env.x.copy_(x0)
    n_or = 2048
    orc = O.OracleEnv(env.spec, n_or, n_threads=8)
    orc.reset()
    orc.x[:] = x0[:, :n_or].cpu().numpy()
    pt = copy.deepcopy(p)
    pt.update(integrator="dopri5", rtol=1e-13, atol=1e-13)
    tru = O.OracleEnv(EnvSpec(pt), n_or, n_threads=8)
    tru.reset()
    worst = 0.0
    for i in range(3):
        a = 2 * torch.rand((2, B), generator=gen, device="cuda", dtype=torch.float64) - 1
        an = a[:, :n_or].cpu().numpy()
        tru.x[:] = orc.x  # one-step truth from the common state
        tru.t = orc.t
        env.step(a)
        orc.step(an)
        tru.step(an)
        H.adaptive_check("multistage_extraction", env.x[:, :n_or].cpu().numpy(), orc.x,
                         env.nsteps[:, :n_or].cpu().numpy(), orc.nsteps, ("configs[2] rodas5", i), tol=1e-11)
        worst = max(worst, float(np.max(np.abs(orc.x - tru.x) / np.abs(tru.x))))
    assert worst <= 1e-6, worst
    assert not env.status.any() and torch.isfinite(env.x).all()
    att = env.nsteps.to(torch.float64).sum(dim=0)
    assert att.mean().item() <= 16.0 and att.max().item() <= 60, (att.mean().item(), att.max().item())  # the fourth-order pair: ~21 / ~100
    for i in range(150):
        env.step(a)
    lo, hi = torch.tensor([5.0, 10.0], device="cuda"), torch.tensor([500.0, 1000.0], device="cuda")
    LG = (a + 1) * ((hi - lo) / 2)[:, None] + lo[:, None]
    L, G = LG[0], LG[1]
    bal = L * (0.6 - env.x[8]) - G * (env.x[1] - 0.05)
    assert (bal.abs() <= 10 * 8e-8 * (L + G)).all(), (bal.abs() / (8e-8 * (L + G))).max().item()
    env.close()


def test_fused_rollout_equals_stepping():
    torch = _torch()
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p.update(integrator="rodas5")
    B, T = 700, 6
    e1, e2 = VecEnv(p, n_envs=B, seed=3), VecEnv(p, n_envs=B, seed=3)
    e1.reset()
    e2.reset()
    gen = torch.Generator(device="cuda").manual_seed(5)
    acts = 2 * torch.rand((T, 2, B), generator=gen, device="cuda", dtype=torch.float64) - 1
    obs_seq, rew_seq = e2.rollout(acts, collect_obs=True, collect_rew=True)
    for i in range(T):
        o, r, d, _, _ = e1.step(acts[i])
        assert torch.equal(o.t().contiguous(), obs_seq[i]) and torch.equal(r, rew_seq[i]), i
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.status, e2.status)
    e1.close()
    e2.close()
