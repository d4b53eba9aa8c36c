"""GPU: the fourth-order stiff integrator (PCG_INT_RODAS4) through the C ABI against its oracle twin.

Dense path (forward-difference Jacobian + per-lane LU in LDS, any model): identical step sequences, states to the
conditioning of W.  Structured path (the 10-state extraction cascade, registers only; classic and work-queue kernels,
both counter modes, fused auto-reset): the arithmetic is an exactly specified operation sequence on both sides --
identical step sequences for EVERY env and states to round-off (helpers.BIT_EXACT_RHS), also at BASELINE configs[2]'s
full size, where the accuracy against a 1e-13 solve is checked too."""
import copy
import ctypes as C

import numpy as np
import pytest

import helpers as H
import scenarios as SC

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    return torch


ROS4 = dict(integrator="rodas4", rtol=1e-6, atol=1e-8)
INT_CASES = [
    ("first_order_system", "first_order_system", ROS4, 5e-8),
    ("cstr", "cstr", ROS4, 5e-8),
    ("cstr_d", "cstr", dict(integrator="rodas4", rtol=1e-7, atol=1e-9), 5e-8),
    ("four_tank", "four_tank", ROS4, 5e-8),
    ("crystallization", "crystallization", ROS4, 5e-8),
    ("distillation_column", "distillation_column", ROS4, 5e-8),
    ("biofilm_reactor", "biofilm_reactor", ROS4, 5e-8),
    ("multistage_extraction_reactive", "multistage_extraction_reactive", ROS4, 5e-8),
    ("heat_exchanger", "heat_exchanger", ROS4, 5e-8),
    ("polymerisation_reactor", "polymerisation_reactor", ROS4, 5e-8),
    # structured W in registers: bit-exact class
    ("multistage_extraction", "multistage_extraction", ROS4, 1e-11),
    ("multistage_extraction_d", "multistage_extraction", dict(integrator="rodas4", rtol=3e-8, atol=3e-8), 1e-11),
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
    if model == "multistage_extraction":  # structured W: exactly specified arithmetic, every sample identical
        H.adaptive_check(model, got, want, ns.cpu().numpy(), ns_o, fix, tol=tol)
    else:
        # dense path: the difference quotients of the Jacobian divide the last-bit differences of two RHS evaluations
        # (OCML vs glibc exp, contraction) by ~1.5e-8 |x_j| -- E moves by ~1e-7 relative, and a sample whose decision
        # lands that close to a threshold takes another, equally valid, step sequence (the igniting cstr samples: 2 of
        # 24).  Identical sequences agree to the conditioning of W; the others to the integrator's accuracy class.
        xs_ = np.maximum(np.abs(want), 1e-6 * np.max(np.abs(want), axis=1, keepdims=True))
        ex = np.max(np.abs(got - want) / xs_, axis=0)
        same = np.all(ns.cpu().numpy() == ns_o, axis=0)
        assert same.mean() >= 0.9, (fix, same.mean())
        assert ex[same].max() <= tol, (fix, ex[same].max())
        assert ex.max() <= 100 * kw["rtol"], (fix, ex.max())
    t = g["xf"].T
    assert np.all(np.abs(got - t) <= 1e-4 * np.abs(t) + 1e-6)  # fourth-order pair at 1e-6: its accuracy class


def _step_pair(name, B, seed, monkeypatch, force_queue, **kw):
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    if force_queue:
        monkeypatch.setenv("PCG_Q_FORCE", "1")
    p = copy.deepcopy(SC.scenarios()[name]["env_params"])
    p.update(integrator="rodas4")
    p.update(kw.pop("params", {}))
    env = VecEnv(p, n_envs=B, seed=seed, **kw)
    orc = O.OracleEnv(env.spec, B, seed=seed, per_env_t=kw.get("per_env_t", False))
    env.reset()
    orc.reset()
    return torch, env, orc


@pytest.mark.parametrize("name", ["me_canonical", "me_dist_cons"])
@pytest.mark.parametrize("per_env_t", [False, True])
@pytest.mark.parametrize("kernel", ["queue", "classic"])
def test_step_vs_oracle_structured(name, per_env_t, kernel, monkeypatch):
    """full step tuples (state, observation, reward, done, status, step counts) of the extraction scenarios, 12 steps
    WITHOUT re-synchronisation, through the work-queue kernel (forced: thin tiles too) and the classic kernel"""
    B = 1500 if kernel == "queue" else 700
    kw = dict(per_env_t=per_env_t)
    if kernel == "classic":
        kw["variant"] = 1
    torch, env, orc = _step_pair(name, B, 11, monkeypatch, kernel == "queue", **kw)
    rng = np.random.default_rng(3)
    for i in range(12):
        a = rng.uniform(-1, 1, (env.spec.na, B))
        if not env.spec.normalise_a:
            a = (a + 1) * (env.spec.a_high - env.spec.a_low)[:, None] / 2 + env.spec.a_low[:, None]
        o, r, d, _, _ = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
        H.adaptive_check("multistage_extraction", env.x.cpu().numpy(), orc.x, env.nsteps.cpu().numpy(), orc.nsteps,
                         (name, kernel, per_env_t, i), tol=1e-11)
        assert np.max(np.abs(env.obs_soa.cpu().numpy() - orc.obs) / np.maximum(np.abs(orc.obs), 1e-3)) <= 1e-10
        assert np.allclose(r.cpu().numpy(), rc, rtol=1e-9, atol=1e-10)
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), dc) and not env.status.any()
        if env.spec.ncon:
            assert np.array_equal(env.viol.cpu().numpy(), orc.viol)
    env.close()


def test_queue_equals_classic_and_is_order_independent(monkeypatch):
    """the work-queue kernel against the classic one (bitwise), and against itself on a permuted batch"""
    torch = _torch()
    from pcgym_amd import VecEnv

    monkeypatch.setenv("PCG_Q_FORCE", "1")
    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p.update(integrator="rodas4")
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
                                 params=dict(N=7, tsim=7.0, SP={"X5": [0.3] * 7}))
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


def test_full_size_configs2_rodas4():
    """BASELINE configs[2] at its size (B = 262,144) with the stiff pair: oracle agreement on a slice of 2048 envs
    (identical step sequences, round-off), the slice within 1e-6 of a 1e-13 solve of the same steps, lane independence,
    and the steady-state solute balance after holding the input."""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd.config import EnvSpec

    B = 1 << 18
    gen = torch.Generator(device="cuda").manual_seed(99)
    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p.update(integrator="rodas4", N=200, tsim=200.0, SP={"X5": [0.3] * 200})
    env = VecEnv(p, n_envs=B)
    assert env.spec.rtol == 3e-8 and env.spec.ep_kmax == 10
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
                         env.nsteps[:, :n_or].cpu().numpy(), orc.nsteps, ("configs[2] rodas4", i), tol=1e-11)
        worst = max(worst, float(np.max(np.abs(orc.x - tru.x) / np.abs(tru.x))))
    assert worst <= 1e-6, worst
    assert not env.status.any() and torch.isfinite(env.x).all()
    att = env.nsteps.to(torch.float64).sum(dim=0)
    assert att.mean().item() <= 30.0, att.mean().item()  # the explicit pair: ~72
    for i in range(150):
        env.step(a)
    lo, hi = torch.tensor([5.0, 10.0], device="cuda"), torch.tensor([500.0, 1000.0], device="cuda")
    LG = (a + 1) * ((hi - lo) / 2)[:, None] + lo[:, None]
    L, G = LG[0], LG[1]
    bal = L * (0.6 - env.x[8]) - G * (env.x[1] - 0.05)
    assert (bal.abs() <= 10 * 3e-8 * (L + G)).all(), (bal.abs() / (3e-8 * (L + G))).max().item()
    env.close()


def test_fused_rollout_equals_stepping_and_dense_path_is_refused():
    """pcg_rollout with the structured pair (state in registers over T steps) == T pcg_step launches, bitwise; the
    dense-W path (LDS matrices) steps through pcg_step only"""
    torch = _torch()
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p.update(integrator="rodas4")
    B, T = 700, 6
    e1, e2 = VecEnv(p, n_envs=B, seed=3), VecEnv(p, n_envs=B, seed=3)
    assert e1.spec.integrator == "rodas4"
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
    p = copy.deepcopy(SC.scenarios()["four_tank_canonical"]["env_params"])
    p.update(integrator="rodas4")
    env = VecEnv(p, n_envs=64)
    env.reset()
    with pytest.raises(Exception):
        env.rollout(torch.zeros((3, 2, 64), device="cuda", dtype=torch.float64))  # PCG_E_UNSUPPORTED
    env.close()


def test_guarded_rk4_vs_oracle_on_the_ignition_box():
    """PCG_INT_RK4G (the cstr default): the same envs are accepted / escalated on both sides (nsteps == (0,0) marks an
    accepted env), accepted envs agree like fixed-step RK4, escalated ones like the adaptive pair"""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
    p.pop("noise", None), p.pop("noise_percentage", None)
    p.update(x0=np.array([0.85, 330.0, 0.85]), uncertainty_percentages={"x0": [0.15 / 0.85, 20.0 / 330.0]})
    for per_env_t in (False, True):
        B = 6000
        env = VecEnv(p, n_envs=B, seed=9, per_env_t=per_env_t)
        assert env.spec.integrator == "tsit5g"
        orc = O.OracleEnv(env.spec, B, seed=9, per_env_t=per_env_t)
        env.reset(), orc.reset()
        rng = np.random.default_rng(1)
        seen_esc = 0
        for i in range(6):
            a = rng.uniform(-1, 1, (1, B))
            o, r, d, _, _ = env.step(torch.tensor(a, device=env.device))
            orc.step(a)
            ng, no = env.nsteps.cpu().numpy(), orc.nsteps
            esc_g, esc_o = ng.sum(axis=0) > 0, no.sum(axis=0) > 0
            assert (esc_g != esc_o).sum() <= 2, (i, (esc_g != esc_o).sum())  # a guard value within round-off of 0 may flip
            same = esc_g == esc_o
            seen_esc += int(esc_g.sum())
            xs = np.maximum(np.abs(orc.x), 1e-9)
            ex = np.max(np.abs(env.x.cpu().numpy() - orc.x) / xs, axis=0)
            assert ex[same & ~esc_g].max() <= 1e-11, (i, ex[same & ~esc_g].max())
            if (same & esc_g).any():
                assert ex[same & esc_g].max() <= 1e-8, (i, ex[same & esc_g].max())  # the ignition front amplifies round-off
                assert np.mean(np.all(ng[:, same & esc_g] == no[:, same & esc_g], axis=0)) >= 0.99
            assert ex.max() <= 1e-6 and not env.status.any()
            env.x.copy_(torch.tensor(orc.x, device=env.device))  # one-step comparisons
        assert seen_esc > B // 10
        env.close()


@pytest.mark.parametrize("scenario", ["heat_exchanger_sp", "biofilm_sp", "me_reactive"])
@pytest.mark.parametrize("integ", ["rodas3", "rodas4", "rodas5"])
def test_step_kernels_of_the_large_models_dense_path(scenario, integ):
    """the GENERAL STEP KERNEL (not pcg_integrate) of the 16- / 20- / 24-state models under the three Rosenbrock integrators,
    both counter modes, one-step comparisons with the oracle from a common state.  Round 5: the 24-state model's lock-stepped
    PCG_INT_RODAS4 / PCG_INT_RODAS5 kernels (all 512 vector registers in use, ~1400 spilled scalar registers) came back with a
    garbage x[2] for most envs under the compiler's scalar-spills-into-vector-lanes -- the fuzz had only covered
    PCG_INT_RODAS3, and the integration hook is another kernel.  Models with more than 16 states now run the attempt with
    loops that stay loops (pcg_integrators.hpp: ros_try_rolled: the same arithmetic, a third of the vector spills);
    tools/integrator_sweep.py runs this check over every registry model."""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    S = SC.scenarios()
    name = scenario if scenario in S else [k for k in S if k.startswith(scenario.split("_")[0])][0]
    for per_env_t in (False, True):
        p = copy.deepcopy(S[name]["env_params"])
        p.update(integrator=integ, rtol=1e-6, atol=1e-8)
        B = 130
        env = VecEnv(copy.deepcopy(p), n_envs=B, seed=3, per_env_t=per_env_t)
        spec = env.spec
        orc = O.OracleEnv(spec, B, seed=3, per_env_t=per_env_t)
        env.reset(), orc.reset()
        rng = np.random.default_rng(1)
        for i in range(3):
            a = rng.uniform(-1, 1, (spec.na, B))
            if not spec.normalise_a:
                a = (a + 1) * (spec.a_high - spec.a_low)[:, None] / 2 + spec.a_low[:, None]
            env.step(torch.tensor(a, device=env.device))
            orc.step(a)
            xg = env.x.cpu().numpy()
            xs = np.maximum(np.abs(orc.x), 1e-6 * np.max(np.abs(orc.x), axis=1, keepdims=True))
            ex = np.max(np.abs(xg - orc.x) / xs, axis=0)
            same = np.all(env.nsteps.cpu().numpy() == orc.nsteps, axis=0)
            assert same.mean() >= 0.95 and ex.max() <= 1e-4 and np.quantile(ex, 0.99) <= 5e-7, (name, integ, per_env_t, i, same.mean(), ex.max())
            assert not env.status.any()
            env.x.copy_(torch.tensor(orc.x, device=env.device))
        env.close()
