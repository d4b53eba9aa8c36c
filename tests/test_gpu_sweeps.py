"""Registry-wide sweeps of the kernel instantiations, inside the gated suite (round 6; rounds 3-5 ran them as builder tools:
tools/integrator_sweep.py, tools/shape_sweep.py -- which is how a silently wrong step_kernel<heat_exchanger, RODAS4|5,
lock-stepped> survived two green rounds).

  test_integrator_sweep        every registry model x every integrator x both counter modes x dispatch: env steps through
                               the step kernels against the oracle, EVERY lane, one-step comparisons from a common state
  test_ros_single_attempt      the Rosenbrock attempt on its own, through the product kernels: a configuration under which
                               the controller takes exactly ONE attempt of size dt and accepts it (huge tolerance, small
                               dt: no decisions) -- every lane at <= 1e-12 against the oracle, for every model x
                               {rodas3, rodas4, rodas5} x both counter modes x {pcg_step, pcg_integrate}
  test_shape_sweep             fused rollout / HIP graph / same-launch auto-reset of a plan == its own step launches
  test_uncertainty_sweep       per-env model parameters (step_kernel<..., UNC> / rollout_kernel<..., UNC>) against the oracle

Together with tests/test_zz_kernel_coverage.py (which fails on a shipped kernel that no passing oracle / fixture test
launched) this is the guard of integrator.py:90-107 for every model of pcgym.py:128-148.
"""
import copy

import numpy as np
import pytest

import scenarios as SC

pytestmark = pytest.mark.gpu

ROS = ("rodas3", "rodas4", "rodas5")
FIXED = ("rk4", "cv8")
ADAPT = ("dopri5", "tsit5") + ROS
GUARDED = ("rk4g", "tsit5g")  # models with a guard hook only (pcg_models.hpp: has_guard)


def _models():
    """first scenario of every registry model (the tools' rule)"""
    out, seen = [], set()
    for name, sc in SC.scenarios().items():
        p0 = sc["env_params"]
        m = p0.get("model")
        if m is None or m in seen or p0.get("custom_model") is not None:
            continue
        seen.add(m)
        out.append((m, name))
    return out


MODELS = _models()
MODEL_NAMES = [m for m, _ in MODELS]
SCEN = dict(MODELS)


def _params(model, integ, **over):
    p = copy.deepcopy(SC.scenarios()[SCEN[model]]["env_params"])
    p.update(integrator=integ, rtol=1e-6, atol=1e-8)
    if integ in FIXED + GUARDED:
        p.pop("rtol"), p.pop("atol")
    for k in ("uncertainty_percentages", "uncertainty_bounds", "distribution", "empirical_distribution"):
        p.pop(k, None)
    p.update(over)
    return p


def _actions(spec, rng, B):
    a = rng.uniform(-1, 1, (spec.na, B))
    if not spec.normalise_a:
        a = (a + 1) * (spec.a_high - spec.a_low)[:, None] / 2 + spec.a_low[:, None]
    return a


def _worst(xg, xo):
    """largest difference over every lane, relative to max(|x|, 1e-6 of the component's range over the batch)"""
    ok = np.isfinite(xo).all(axis=0)
    assert np.array_equal(np.isfinite(xg).all(axis=0), ok), "failure pattern differs from the oracle's"
    if not ok.any():
        return 0.0
    xs = np.maximum(np.abs(xo[:, ok]), 1e-6 * np.max(np.abs(xo[:, ok]), axis=1, keepdims=True))
    xs = np.maximum(xs, 1e-300)
    return float(np.max(np.abs(xg[:, ok] - xo[:, ok]) / xs))


def _make(p, B, **kw):
    from pcgym_amd import VecEnv

    try:
        return VecEnv(copy.deepcopy(p), n_envs=B, **kw)
    except ValueError as e:  # a combination the plan refuses by design (e.g. a guarded scheme on a model without a guard)
        pytest.skip(f"refused at plan creation: {str(e)[:100]}")


# default dispatch, the classic one-env-per-lane kernels (PCG_OPT_VARIANT 1), the two persistent streaming shapes (2, 3:
# RK4 / DOPRI5 of the five full models only -- anything else falls back to the default inside the library)
DISPATCH = {"auto": None, "classic": 1, "stream1": 2, "stream2": 3}


@pytest.mark.parametrize("dispatch", list(DISPATCH))
@pytest.mark.parametrize("pe", [False, True], ids=["lockstep", "per_env_t"])
@pytest.mark.parametrize("integ", FIXED + ADAPT + GUARDED)
@pytest.mark.parametrize("model", MODEL_NAMES)
def test_integrator_sweep(model, integ, pe, dispatch):
    import torch
    from oracle import oracle as O

    if dispatch.startswith("stream") and integ not in ("rk4", "dopri5"):
        pytest.skip("no streaming kernel for this integrator")
    if integ in GUARDED and model != "cstr":
        pytest.skip("guarded schemes exist for models with a guard hook")
    B = 130 if not dispatch.startswith("stream") else 700
    kw = {"variant": DISPATCH[dispatch]} if DISPATCH[dispatch] else {}
    env = _make(_params(model, integ), B, seed=3, per_env_t=pe, **kw)
    spec = env.spec
    orc = O.OracleEnv(spec, B, seed=3, per_env_t=pe)
    env.reset(), orc.reset()
    rng = np.random.default_rng(1)
    worst, same = 0.0, 1.0
    for _ in range(3):
        a = _actions(spec, rng, B)
        og, rg, dg, _, _ = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
        worst = max(worst, _worst(env.x.cpu().numpy(), orc.x))
        fin = np.isfinite(rc)
        assert np.allclose(rg.cpu().numpy()[fin], rc[fin], rtol=1e-6, atol=1e-9 * (1 + np.max(np.abs(rc[fin]), initial=0)))
        assert np.array_equal(dg.cpu().numpy().astype(bool), dc.astype(bool))
        if env.nsteps is not None and orc.nsteps is not None:
            same = min(same, float(np.mean(np.all(env.nsteps.cpu().numpy() == orc.nsteps, axis=0))))
        env.x.copy_(torch.tensor(orc.x, device=env.device))  # the next step starts from a common state
    env.close()
    # measured over all 266 + 228 combinations on the round-5 build: worst 4.8e-8 (a difference-quotient Jacobian through
    # three adaptive steps), identical step sequences >= 0.992
    assert worst <= 1e-6, f"worst relative difference over every lane {worst:.2e}"
    assert same >= 0.98, f"identical step sequences on {same:.3f} of the lanes"


def _single_attempt_params(model, integ, dt):
    p = _params(model, integ)
    N = int(p["N"])
    p.update(tsim=dt * N, rtol=1.0, atol=1.0)
    for k in ("SP", "disturbances"):
        if p.get(k):
            p[k] = {kk: list(np.asarray(v, dtype=float)) for kk, v in p[k].items()}
    return p


# env-step lengths (in the model's own time unit) at which every lane's first attempt IS the env step: h = min(c h0, dt) =
# dt with h0 = 0.01 |x| / |f| in the tolerance-scaled norm, and at tolerance 1 any finite attempt is accepted.  Two sizes
# per model: one where W = I / (gamma h) - J is dominated by the diagonal, one as large as the first-step rule allows
# (h |J| up to ~1 for the stiff models), where the factorisation, the pivoting and every stage's solve carry weight.
@pytest.mark.parametrize("entry", ["step", "integrate"])
@pytest.mark.parametrize("pe", [False, True], ids=["lockstep", "per_env_t"])
@pytest.mark.parametrize("integ", ROS)
@pytest.mark.parametrize("model", MODEL_NAMES)
def test_ros_single_attempt(model, integ, pe, entry):
    import torch
    from oracle import oracle as O

    if entry == "integrate" and pe:
        pytest.skip("pcg_integrate has no counter mode")
    B = 192
    base = SC.scenarios()[SCEN[model]]["env_params"]
    dt0 = float(base["tsim"]) / int(base["N"])
    tested = 0
    for frac in (1e-6, 1e-4, 1e-3, 1e-2):
        dt = dt0 * frac
        p = _single_attempt_params(model, integ, dt)
        env = _make(p, B, seed=11, per_env_t=pe)
        spec = env.spec
        orc = O.OracleEnv(spec, B, seed=11, per_env_t=pe)
        env.reset(), orc.reset()
        rng = np.random.default_rng(5)
        # spread the start states over the observation box (the scenarios start every env at one x0)
        x0 = orc.x.copy()
        x0 *= 1.0 + 0.2 * rng.uniform(-1, 1, x0.shape)
        a = _actions(spec, rng, B)
        if entry == "step":
            env.x.copy_(torch.tensor(x0, device=env.device))
            orc.x[:] = x0
            env.step(torch.tensor(a, device=env.device))
            orc.step(a)
            xg, xo, ng, no = env.x.cpu().numpy(), orc.x, env.nsteps.cpu().numpy(), orc.nsteps
        else:
            u = np.zeros((spec.nu, B))  # the held input [actions | model disturbance inputs at their parameter values]
            u[:spec.na] = a if not spec.normalise_a else (a + 1) * (spec.a_high - spec.a_low)[:, None] / 2 + spec.a_low[:, None]
            for j, name in enumerate(spec.model.disturbances[: spec.nu - spec.na]):
                u[spec.na + j] = float(spec.model.parameters[name])
            xo, no = O.integrate(spec, x0, u)
            xt, ut = torch.tensor(x0, device=env.device), torch.tensor(u, device=env.device)
            nt = torch.zeros((2, B), dtype=torch.int32, device=env.device)
            assert env._lib.pcg_integrate(env._plan, B, xt.data_ptr(), ut.data_ptr(), nt.data_ptr(), None) == 0
            torch.cuda.synchronize()
            xg, ng = xt.cpu().numpy(), nt.cpu().numpy()
        env.close()
        if not (np.all(no[0] == 1) and np.all(no[1] == 0)):
            continue  # at this dt some lane's first-step rule asks for less than dt: not a single-attempt configuration
        tested += 1
        assert np.array_equal(ng, no), "the kernels took another step sequence than the oracle"
        w = _worst(xg, xo)
        assert w <= 1e-12, f"dt = {frac:g} x the scenario's: one attempt differs by {w:.2e} on some lane"
        # ... and the attempt moved the state: the comparison is not x == x
        assert np.max(np.abs(xo - x0)) > 0
    assert tested >= 1, "no single-attempt configuration found for this model"


def _close(a, b, tol=1e-9):
    import torch

    a, b = a.double(), b.double()
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    if not torch.equal(fa, fb):
        return float("inf")
    if not fa.any():
        return 0.0
    return ((a[fa] - b[fa]).abs() / b[fa].abs().clamp_min(1e-9)).max().item()


@pytest.mark.parametrize("integ", FIXED + ADAPT)
@pytest.mark.parametrize("model", MODEL_NAMES)
def test_shape_sweep(model, integ):
    """rollout (T steps, state in registers) == T step launches; a HIP graph of the T steps == the launches, bitwise;
    same-launch auto-reset through an episode end == step + reset.  (The step launches themselves are held against the
    oracle by test_integrator_sweep.)"""
    import torch

    B, T = 200, 5
    base = SC.scenarios()[SCEN[model]]["env_params"]
    p = _params(model, integ, N=T + 2, tsim=float(base["tsim"]) * (T + 2) / base["N"])
    for k in ("SP", "disturbances"):
        if p.get(k):
            p[k] = {kk: list(np.asarray(v, dtype=float)[: T + 2]) for kk, v in p[k].items()}
    e_step, e_roll, e_graph = (_make(p, B, seed=5) for _ in range(3))
    ar = _make(p, B, seed=5, auto_reset=True)
    ref = _make(p, B, seed=5)
    spec = e_step.spec
    gen = torch.Generator(device="cuda").manual_seed(7)
    acts = 2 * torch.rand((T + 4, spec.na, B), generator=gen, device="cuda", dtype=torch.float64) - 1
    if not spec.normalise_a:
        lo = torch.tensor(spec.a_low, device="cuda")[None, :, None]
        hi = torch.tensor(spec.a_high, device="cuda")[None, :, None]
        acts = (acts + 1) * (hi - lo) / 2 + lo
    for e in (e_step, e_roll, e_graph, ar, ref):
        e.reset()
    obs_s, rew_s = [], []
    for i in range(T):
        o, r, d, _, _ = e_step.step(acts[i])
        obs_s.append(e_step.obs_soa.clone()), rew_s.append(r.clone())
    try:
        oq, rq = e_roll.rollout(acts[:T], collect_obs=True, collect_rew=True)
        rolled = True
    except Exception as e:  # noqa: BLE001  (no fused rollout for this integrator: PCG_E_UNSUPPORTED)
        from pcgym_amd._lib import PcgError

        assert isinstance(e, (PcgError, ValueError)), e
        rolled = False
    if rolled:
        # bitwise for most shapes; the fused rollout of some models contracts differently
        ds = [_close(e_roll.x, e_step.x)] + [_close(rq[i], rew_s[i]) for i in range(T)] + [_close(oq[i], obs_s[i]) for i in range(T)]
        assert max(ds) <= 1e-9, f"fused rollout differs from its step launches by {max(ds):.2e}"
        assert torch.equal(e_roll.status, e_step.status)
    g = e_graph.capture_steps([acts[i] for i in range(T)])
    g.replay()
    torch.cuda.synchronize()
    assert _close(e_graph.x, e_step.x) == 0.0 and _close(e_graph.rew, rew_s[-1]) == 0.0, "graph replay != launches"
    assert torch.equal(e_graph.status, e_step.status)
    for i in range(T + 2):  # auto-reset through the episode end (N - 1 = T + 1 steps), then one step of the next episode
        o, r, d, _, _ = ar.step(acts[i])
        if ref.t == ref.N - 1:
            ref.reset()
        o2, r2, d2, _, _ = ref.step(acts[i])
        assert _close(r, r2) == 0.0 and torch.equal(d, d2), f"auto-reset step {i}: reward / done differ"
        if ref.t != ref.N - 1:  # (at the episode end `ar` already holds the NEW x0)
            assert _close(ar.x, ref.x) == 0.0, f"auto-reset step {i}: state differs"
    for e in (e_step, e_roll, e_graph, ar, ref):
        e.close()


def _unc_params(model, integ):
    from pcgym_amd.models import get_model

    mi = get_model(model)
    if mi.affine_builder is not None:
        pytest.skip("affine registry models have no per-env parameter kernel")
    names = [k for k, v in mi.parameters.items() if float(v) != 0.0 and k not in ("N", "eq_exponent")]
    if not names:
        pytest.skip("no parameter to perturb")
    pick = names[:2]
    p = _params(model, integ)
    p.update(uncertainty_percentages={k: 0.03 for k in pick}, distribution="uniform",
             uncertainty_bounds={"low": np.array([min(0.9 * mi.parameters[k], 1.1 * mi.parameters[k]) for k in pick]),
                                 "high": np.array([max(0.9 * mi.parameters[k], 1.1 * mi.parameters[k]) for k in pick])})
    return p


@pytest.mark.parametrize("pe", [False, True], ids=["lockstep", "per_env_t"])
@pytest.mark.parametrize("integ", ["rk4", "dopri5"])
@pytest.mark.parametrize("model", MODEL_NAMES)
def test_uncertainty_sweep(model, integ, pe):
    """row f-3 for every model: per-env parameters sampled at reset (pcgym.py:212-253, 301-316), the dynamics use each
    env's own values: step kernels and the fused rollout against the oracle"""
    import torch
    from oracle import oracle as O

    B, T = 160, 3
    p = _unc_params(model, integ)
    env = _make(p, B, seed=21, per_env_t=pe, env_offset=1000)
    spec = env.spec
    orc = O.OracleEnv(spec, B, seed=21, per_env_t=pe, env_offset=1000)
    env.reset(), orc.reset()
    assert np.allclose(env.p_unc.cpu().numpy(), orc.p_unc, rtol=1e-14)
    assert np.std(orc.p_unc, axis=1).min() > 0
    rng = np.random.default_rng(2)
    acts = [_actions(spec, rng, B) for _ in range(T)]
    x_start = env.x.clone()
    for a in acts:
        env.step(torch.tensor(a, device=env.device)), orc.step(a)
        assert _worst(env.x.cpu().numpy(), orc.x) <= 1e-8
        assert np.allclose(env.obs_soa.cpu().numpy(), orc.obs, rtol=1e-8, atol=1e-9)
        env.x.copy_(torch.tensor(orc.x, device=env.device))
    if not pe:  # the fused rollout with per-env parameters (lock-stepped by construction)
        env2 = _make(p, B, seed=21, env_offset=1000)
        env2.reset()
        assert torch.equal(env2.x, x_start)
        orc2 = O.OracleEnv(spec, B, seed=21, env_offset=1000)
        orc2.reset()
        env2.rollout(torch.tensor(np.stack(acts), device=env2.device), collect_rew=True)
        for a in acts:
            orc2.step(a)
        assert _worst(env2.x.cpu().numpy(), orc2.x) <= 1e-7
        env2.close()
    env.close()
