"""GPU: the fixed-step plans added late in round 3 through the C ABI against their oracle twins -- Cooper & Verner's order-8
method (PCG_INT_CV8: four_tank's default; lean pipelined kernel, general kernel, fused rollout, integration hook, run-time
compiled user models) and the guarded fixed-step Tsit5 plan of the cstr (PCG_INT_T5G, the model's default: general
kernel)."""
import copy

import numpy as np
import pytest

import helpers as H
import scenarios as SC

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    return torch


CASES = [("cstr", "cstr"), ("four_tank", "four_tank"), ("multistage_extraction", "multistage_extraction"),
         ("crystallization", "crystallization"), ("heat_exchanger", "heat_exchanger"), ("first_order_system", "first_order_system")]


@pytest.mark.parametrize("fix,model", CASES)
def test_cv8_integrate_vs_oracle(fix, model):
    torch = _torch()
    from oracle import oracle as O
    from test_gpu_parity import _plan_for
    from test_oracle_golden import _spec_for_integration

    g = H.gold("tight_" + fix)
    spec = _spec_for_integration(model, float(g["dt"]), g["u"].shape[1], integrator="cv8", substeps=512)
    lib, plan = _plan_for(spec, torch)
    xs, us = g["x"].T.copy(), g["u"].T.copy()
    x, u = torch.tensor(xs, device="cuda"), torch.tensor(us, device="cuda")
    assert lib.pcg_integrate(plan, x.shape[1], x.data_ptr(), u.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    lib.pcg_plan_destroy(plan)
    want, _ = O.integrate(spec, xs, us)
    ok = np.isfinite(want).all(axis=0)  # (igniting cstr samples overflow a fixed step on both sides)
    got = x.cpu().numpy()
    assert ok.sum() >= 15 and np.array_equal(np.isfinite(got).all(axis=0), ok)
    got, want = got[:, ok], want[:, ok]
    scale = np.maximum(np.abs(want), 1e-6 * np.max(np.abs(want), axis=1, keepdims=True))
    assert np.max(np.abs(got - want) / scale) <= 1e-10  # 512 steps of round-off, exp / sqrt implementations
    if model != "cstr":
        t = g["xf"].T[:, ok]
        assert np.all(np.abs(got - t) <= 1e-8 * np.abs(t) + 1e-9)


@pytest.mark.parametrize("name", ["four_tank_canonical", "four_tank_paper_reward", "cstr_cons_pen_norm", "cryst_adelta", "heat_exchanger_sp"])
@pytest.mark.parametrize("per_env_t", [False, True])
def test_cv8_steps_vs_oracle(name, per_env_t):
    """lean pipelined kernel (four_tank_canonical, lock-stepped), general kernel (constraints, a_delta, per-env counters)"""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()[name]["env_params"])
    if not name.startswith("four_tank"):
        p.update(integrator="cv8", substeps=4 if "cstr" in name else 24)
    B = 1000
    env = VecEnv(p, n_envs=B, seed=5, per_env_t=per_env_t)
    assert env.spec.integrator == "cv8"
    orc = O.OracleEnv(env.spec, B, seed=5, per_env_t=per_env_t)
    env.reset(), orc.reset()
    rng = np.random.default_rng(2)
    for i in range(10):
        a = rng.uniform(-0.5 if name.startswith("four_tank") else -1, 1, (env.spec.na, B))  # (four_tank: levels stay positive)
        if not env.spec.normalise_a:
            a = (a + 1) * (env.spec.a_high - env.spec.a_low)[:, None] / 2 + env.spec.a_low[:, None]
        o, r, d, _, info = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
        xs = np.maximum(np.abs(orc.x), 1e-9)
        assert np.max(np.abs(env.x.cpu().numpy() - orc.x) / xs) <= 1e-11, (name, i)
        assert np.allclose(env.obs_soa.cpu().numpy(), oc, rtol=1e-10, atol=1e-11)
        assert np.allclose(r.cpu().numpy(), rc, rtol=1e-9, atol=1e-10) and np.array_equal(d.cpu().numpy().astype(np.uint8), dc)
        assert not env.status.any()
    env.close()


def test_cv8_odd_batch_and_one_env_per_lane_agree_with_two():
    """B odd -> the one-env-per-lane instantiation of the lean kernel; same bits as the two-envs-per-lane one"""
    torch = _torch()
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()["four_tank_canonical"]["env_params"])
    outs = []
    for B in (2048, 2047):
        env = VecEnv(p, n_envs=B, seed=3)
        env.reset()
        gen = torch.Generator(device="cuda").manual_seed(0)
        a = 1.5 * torch.rand((2, 2048), generator=gen, device="cuda", dtype=torch.float64) - 0.5  # (upper 3/4 of the box: levels stay positive)
        for i in range(5):
            env.step(a[:, :B].contiguous())
        outs.append(env.x[:, :2047].clone())
        env.close()
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])


def test_cv8_fused_rollout_and_autoreset_match_stepping():
    torch = _torch()
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()["four_tank_canonical"]["env_params"])
    B = 512
    env = VecEnv(p, n_envs=B, seed=11)
    N = env.spec.N
    gen = torch.Generator(device="cuda").manual_seed(4)
    acts = 1.5 * torch.rand((N, 2, B), generator=gen, device="cuda", dtype=torch.float64) - 0.5
    env.reset()
    obs_seq, rew_seq = env.rollout(acts[:N - 1], collect_obs=True)
    x_roll = env.x.clone()
    env.reset()
    for t in range(N - 1):
        o, r, d, _, _ = env.step(acts[t])
        assert torch.allclose(env.obs_soa, obs_seq[t], rtol=1e-12, atol=1e-13), t
        assert torch.allclose(r, rew_seq[t], rtol=1e-11, atol=1e-13), t
    assert torch.allclose(env.x, x_roll, rtol=1e-12, atol=1e-14)
    env.close()
    # same-launch auto-reset on the episode's last step (the AR instantiation of the lean kernel) == step, then reset
    e1, e2 = VecEnv(p, n_envs=B, seed=11, auto_reset=True), VecEnv(p, n_envs=B, seed=11)
    e1.reset(), e2.reset()
    for t in range(N - 1):
        o1, r1, d1, _, _ = e1.step(acts[t])
        o2, r2, d2, _, _ = e2.step(acts[t])
        assert torch.equal(r1, r2) and torch.equal(d1, d2)
    assert bool(d1.all()) and int(e1.t) == 0
    e2.reset()
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.obs_soa, e2.obs_soa)
    e1.close(), e2.close()


def test_cv8_on_a_run_time_compiled_user_model():
    """integrator id reaches the hipRTC instantiation: a pendulum as C expressions, CV8 x 4 against the oracle's user RHS"""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    cm = {"states": ["th", "om"], "inputs": ["tau"], "parameters": {"g_l": 9.81, "c": 0.3},
          "rhs": ["om", "-g_l*sin(th) - c*om + tau"]}
    p = dict(N=30, tsim=3.0, SP={"th": [0.5] * 30}, r_scale={"th": 1.0},
             o_space={"low": np.array([-4.0, -10.0, -4.0]), "high": np.array([4.0, 10.0, 4.0])},
             a_space={"low": np.array([-2.0]), "high": np.array([2.0])}, x0=np.array([0.3, 0.0, 0.5]), custom_model=cm,
             normalise_a=True, normalise_o=True, integrator="cv8", substeps=4)
    env = VecEnv(p, n_envs=512, seed=1)
    assert env.spec.integrator == "cv8"
    O.register_user_rhs(env.spec)
    orc = O.OracleEnv(env.spec, 512, seed=1)
    env.reset(), orc.reset()
    rng = np.random.default_rng(0)
    for i in range(6):
        a = rng.uniform(-1, 1, (1, 512))
        env.step(torch.tensor(a, device="cuda"))
        orc.step(a)
        assert np.max(np.abs(env.x.cpu().numpy() - orc.x)) <= 1e-11
    env.close()


@pytest.mark.parametrize("integrator", ["tsit5g", "rk4g"])
@pytest.mark.parametrize("per_env_t", [False, True])
def test_guarded_plans_vs_oracle_on_the_ignition_box(integrator, per_env_t):
    """the same envs are accepted / escalated on both sides (nsteps == (0,0) marks an accepted env), accepted envs agree like a
    fixed step, escalated ones like the adaptive pair"""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
    p.pop("noise", None), p.pop("noise_percentage", None)
    p.update(x0=np.array([0.85, 330.0, 0.85]), uncertainty_percentages={"x0": [0.15 / 0.85, 20.0 / 330.0]}, integrator=integrator)
    B = 6000
    env = VecEnv(p, n_envs=B, seed=9, per_env_t=per_env_t)
    orc = O.OracleEnv(env.spec, B, seed=9, per_env_t=per_env_t)
    env.reset(), orc.reset()
    rng = np.random.default_rng(1)
    seen_esc = 0
    for i in range(6):
        a = rng.uniform(-1, 1, (1, B))
        o, r, d, _, _ = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
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
        assert np.allclose(r.cpu().numpy()[same], rc[same], rtol=1e-6, atol=1e-9)
        env.x.copy_(torch.tensor(orc.x, device=env.device))  # one-step comparisons
    assert seen_esc > B // 10
    env.close()


@pytest.mark.parametrize("integrator,nsub", [("rk4g", 5), ("tsit5g", 2)])
def test_guarded_plans_closed_loop_is_never_escalated_and_matches_tight(integrator, nsub):
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd.config import EnvSpec

    p = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
    p.pop("noise", None), p.pop("noise_percentage", None)
    p["integrator"] = integrator
    B = 1 << 16
    env = VecEnv(p, n_envs=B, seed=2)
    assert env.spec.substeps == nsub
    pt = dict(p)
    pt.update(integrator="dopri5", rtol=1e-13, atol=1e-13)
    orc = O.OracleEnv(EnvSpec(pt), 2048, seed=2)
    env.reset(), orc.reset()
    gen = torch.Generator(device="cuda").manual_seed(0)
    worst = 0.0
    for i in range(env.spec.N - 1):
        a = 2 * torch.rand((1, B), generator=gen, device="cuda", dtype=torch.float64) - 1
        env.step(a)
        assert int(env.nsteps.sum()) == 0, i
        orc.step(a[:, :2048].cpu().numpy())
        worst = max(worst, float(np.max(np.abs(env.x[:, :2048].cpu().numpy() - orc.x) / np.abs(orc.x))))
        orc.x[:] = env.x[:, :2048].cpu().numpy()  # one-step errors
    assert worst <= 1e-6, worst
    env.close()


def test_work_queue_option_on_a_model_without_cost_key():
    """PCG_OPT_VARIANT 5: the adaptive pair of ANY model through the in-workgroup work queue (default only for the extraction
    models) -- same step sequences and states as the classic one-env-per-lane kernel, on a batch with a heavy tail (cstr
    over the ignition box)"""
    torch = _torch()
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
    p.pop("noise", None), p.pop("noise_percentage", None)
    p.update(x0=np.array([0.85, 330.0, 0.85]), uncertainty_percentages={"x0": [0.15 / 0.85, 20.0 / 330.0]}, integrator="dopri5")
    B = 20000
    q, c = VecEnv(p, n_envs=B, seed=4, variant=5), VecEnv(p, n_envs=B, seed=4, variant=1)
    q.reset(), c.reset()
    gen = torch.Generator(device="cuda").manual_seed(0)
    tail = 0
    for i in range(12):
        a = 2 * torch.rand((1, B), generator=gen, device="cuda", dtype=torch.float64) - 1
        oq, rq, dq, _, _ = q.step(a)
        oc, rc, dc, _, _ = c.step(a)
        assert torch.equal(q.nsteps, c.nsteps) and torch.equal(q.x, c.x) and torch.equal(rq, rc), i
        tail = max(tail, int(q.nsteps.sum(dim=0).max()))
    assert tail > 3 * int(q.nsteps.sum(dim=0).median())  # the tail is really there (ignition fronts)
    q.close(), c.close()


def test_full_size_default_plans_of_four_tank_and_cryst():
    """BASELINE sizes under the round-3 default plans: four_tank B = 2^20 (one order-8 step, lean pipelined kernel) and
    crystallization B = 2^18 (four order-8 steps, general kernel): lane independence under a permutation (bitwise), an
    oracle slice to round-off, a tight solve of the slice inside the plan's accuracy class, physical state boxes."""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd.config import EnvSpec

    gen = torch.Generator(device="cuda").manual_seed(7)
    for name, B, steps, tol_tight in (("four_tank_canonical", 1 << 20, 12, 1e-6), ("cryst_adelta", 1 << 18, 8, 1e-6)):
        p = copy.deepcopy(SC.scenarios()[name]["env_params"])
        env, env2 = VecEnv(p, n_envs=B, seed=3), VecEnv(p, n_envs=B, seed=3)
        assert env.spec.integrator == "cv8"
        env.reset(), env2.reset()
        x0 = env.x * (1 + 0.02 * (2 * torch.rand(env.x.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
        perm = torch.randperm(B, generator=gen, device="cuda")
        env.x.copy_(x0), env2.x.copy_(x0[:, perm])
        n_or = 2048
        orc = O.OracleEnv(env.spec, n_or, seed=3)
        pt = copy.deepcopy(p)
        pt.update(integrator="dopri5", rtol=1e-13, atol=1e-13)
        tru = O.OracleEnv(EnvSpec(pt), n_or, seed=3)
        orc.reset(), tru.reset()
        orc.x[:] = x0[:, :n_or].cpu().numpy()
        tru.x[:] = orc.x
        na = env.spec.na
        for i in range(steps):
            a = 1.5 * torch.rand((na, B), generator=gen, device="cuda", dtype=torch.float64) - 0.5  # (four_tank: levels stay positive)
            o1, r1, d1, _, _ = env.step(a)
            o2, r2, d2, _, _ = env2.step(a[:, perm].contiguous())
            orc.step(a[:, :n_or].cpu().numpy()), tru.step(a[:, :n_or].cpu().numpy())
            assert torch.equal(env.x[:, perm], env2.x) and torch.equal(r1[perm], r2), (name, i)
        xg = env.x[:, :n_or].cpu().numpy()
        sc = np.maximum(np.abs(orc.x), 1e-9 * np.abs(orc.x).max(axis=1, keepdims=True))
        assert np.max(np.abs(xg - orc.x) / sc) <= 1e-11, name
        st = np.maximum(np.abs(tru.x), 1e-9 * np.abs(tru.x).max(axis=1, keepdims=True))
        assert np.max(np.abs(xg - tru.x) / st) <= tol_tight, (name, np.max(np.abs(xg - tru.x) / st))
        assert bool(torch.isfinite(env.x).all()) and not env.status.any()
        if name.startswith("four_tank"):
            assert bool((env.x > 0).all()) and bool((env.x < 5.0).all())
        env.close(), env2.close()


@pytest.mark.parametrize("integrator", ["tsit5g", "rk4g"])
@pytest.mark.parametrize("kw", [{}, dict(per_env_t=True, auto_reset=True), dict(auto_reset=True)])
def test_guarded_plans_two_launch_form_is_the_single_launch_bit_for_bit(integrator, kw, monkeypatch):
    """Batches that fill the chip run a guarded plan as two launches -- the general kernel marks the envs its guard does not
    trust, the work-queue kernel of the adaptive pair integrates exactly those (pcg_abi.hip) -- where smaller ones keep the
    fallback inside the first kernel (the form the oracle tests above pin).  The arithmetic of an env is the same either way:
    on the ignition box, with observation noise and a constraint, both forms give the same bits in every output, every step,
    through episode ends (PCG_NO_FIXUP=1 keeps the single launch at any size)."""
    torch = _torch()
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()["cstr_cons_pen_norm"]["env_params"])
    p.update(x0=np.array([0.85, 330.0, 0.85]), uncertainty_percentages={"x0": [0.15 / 0.85, 20.0 / 330.0]}, integrator=integrator,
             noise=True, noise_percentage=0.002)
    B = (1 << 17) + 777
    two = VecEnv(copy.deepcopy(p), n_envs=B, seed=4, **kw)
    one = VecEnv(copy.deepcopy(p), n_envs=B, seed=4, **kw)
    two.reset(), one.reset()
    assert torch.equal(two.x, one.x)
    gen = torch.Generator(device="cuda").manual_seed(3)
    escalated = 0
    for i in range(two.spec.N + 3 if kw else 8):
        a = 2 * torch.rand((two.spec.na, B), generator=gen, device="cuda", dtype=torch.float64) - 1
        monkeypatch.delenv("PCG_NO_FIXUP", raising=False)
        o2, r2, d2, _, _ = two.step(a)
        monkeypatch.setenv("PCG_NO_FIXUP", "1")
        o1, r1, d1, _, _ = one.step(a)
        monkeypatch.delenv("PCG_NO_FIXUP", raising=False)
        escalated += int((two.nsteps.sum(dim=0) > 0).sum())
        for name, u, v in (("x", two.x, one.x), ("obs", o2, o1), ("rew", r2, r1), ("done", d2, d1), ("nsteps", two.nsteps, one.nsteps),
                           ("status", two.status, one.status), ("viol", two.viol, one.viol)):
            if u is not None:
                assert torch.equal(u, v), (i, name)
        assert int(d2.max()) <= 1  # no pending mark survives the second launch
        if two.t_env is not None:
            assert torch.equal(two.t_env, one.t_env)
        if not kw and i == 3:
            break
    assert escalated > B // 20
    two.close(), one.close()
