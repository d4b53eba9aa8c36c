"""Registry-wide sweeps of the kernel instantiations, inside the gated suite (round 6; rounds 3-5 ran them as builder tools:
tools/integrator_sweep.py, tools/shape_sweep.py -- which is how a silently wrong step_kernel<heat_exchanger, RODAS4|5,
lock-stepped> survived two green rounds).

  test_integrator_sweep     every model x integrator x counter mode x feature set x dispatch: env steps through the step
                            kernels against the oracle, EVERY lane, one-step comparisons from a common state
  test_ros_single_attempt   the Rosenbrock attempt on its own, through the product kernels: a configuration under which the
                            controller takes exactly ONE attempt of size dt and accepts it (huge tolerance, small dt: no
                            decisions) -- every lane at <= 1e-12 (+ 5e-9 of its own increment) against the oracle, for every
                            model x {rodas3, rodas4, rodas5} x both counter modes x {pcg_step, pcg_integrate}
  test_integrate_sweep      pcg_integrate (the boundary of reference_engine.hip_integration_engine) for every model x
                            integrator against the oracle
  test_shape_sweep          fused rollout / HIP graph / same-launch auto-reset of a plan == its own step launches, which are
                            held against the oracle in the same test
  test_uncertainty_sweep    per-env model parameters (step_kernel<..., UNC> / rollout_kernel<..., UNC>) against the oracle
  test_feature_kernels      the feature-masked pipelined kernels of the two small models, mask by mask
  test_queue_shapes         the work-queue launch shapes that only large batches take (512-thread workgroups, one workgroup
                            per CU) against oracle slices

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
GUARDED = ("rk4g", "tsit5g")  # models with a guard hook only (pcg_models.hpp: has_guard -- the cstr)
FULL = ("cstr", "four_tank", "multistage_extraction", "multistage_extraction_reactive", "crystallization",
        "first_order_system", "hydraulic_tank", "nonsmooth_control")  # Model::FULL: streaming / pipelined / LDS-stage kernels


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
SCEN = dict(MODELS)
# the extraction models carry two instantiations each: eq_exponent == 2 (the reference's default: multiply-only kernels,
# PCG_KID_ME_SQ / _REACTIVE_SQ) and the pow() form (Model<PCG_MODEL_ME>, <PCG_MODEL_ME_REACTIVE>): "^1.5" selects the latter
MODEL_KEYS = [m for m, _ in MODELS] + ["multistage_extraction^1.5", "multistage_extraction_reactive^1.5"]


def _registry_object(model, **params):
    """an object the way the reference's registry classes look to make_env (pcgym.py:150-153): class name, info()"""
    from pcgym_amd.models import get_model

    mi = get_model(model)
    info = {"parameters": {**mi.parameters, **params}, "states": list(mi.states), "inputs": list(mi.inputs),
            "disturbances": list(mi.disturbances)}
    return type(model, (), {"info": lambda self: info, "int_method": "hip"})()


def _params(key, integ, feat="scen", **over):
    """env_params of the model's first scenario under `integ`, in one of three feature sets:
      scen  as the scenario has it
      lean  nothing beyond the set-point reward (the kernels' lean forms: EXTRAS = false, pipelined / streaming paths)
      cons  lean + one constraint row with the penalty on (EXTRAS = true, the feature-masked kernels of the small models)"""
    model, _, expo = key.partition("^")
    p = copy.deepcopy(SC.scenarios()[SCEN[model]]["env_params"])
    if expo:
        p["custom_model"] = _registry_object(model, eq_exponent=float(expo))
    p.update(integrator=integ, rtol=1e-6, atol=1e-8)
    if integ in FIXED + GUARDED:
        p.pop("rtol"), p.pop("atol")
    if integ == "cv8" and model.startswith("multistage"):
        p["substeps"] = 256  # (the model's default plan is implicit: the order-8 scheme's own default step is unstable here)
    for k in ("uncertainty_percentages", "uncertainty_bounds", "distribution", "empirical_distribution"):
        p.pop(k, None)
    if feat != "scen":
        for k in ("a_delta", "a_0", "a_space_act", "noise", "noise_percentage", "constraints", "done_on_cons_vio",
                  "r_penalty", "custom_reward"):
            p.pop(k, None)
        if not p.get("SP"):  # terminal-reward scenarios: a set point on the first state instead
            from pcgym_amd.models import get_model

            mi = get_model(model)
            nx = len(mi.states)
            x0 = np.asarray(p["x0"], dtype=float)[:nx]
            for k in ("reward_states", "maximise_reward"):
                p.pop(k, None)
            sp = float(x0[0]) if x0[0] != 0 else 0.5
            p["SP"] = {mi.states[0]: [sp] * int(p["N"])}
            p["x0"] = np.concatenate([x0, [sp]])
            lo, hi = np.asarray(p["o_space"]["low"], dtype=float)[:nx], np.asarray(p["o_space"]["high"], dtype=float)[:nx]
            p["o_space"] = {"low": np.concatenate([lo, [min(0.0, 2 * sp)]]), "high": np.concatenate([hi, [max(1.0, 2 * sp)]])}
            p["r_scale"] = {mi.states[0]: 1.0}
    if feat == "cons":
        c0 = float(np.asarray(p["x0"], dtype=float)[0])
        p.update(constraints=lambda x, u, c0=c0: np.array([x[0] - c0]).reshape(-1,), done_on_cons_vio=False, r_penalty=True)
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


def _bars(key, integ):
    """(largest difference over every lane, share of lanes with the oracle's step sequence) one env step may show.
    The pow() form of the extraction cascades under an EXPLICIT adaptive pair runs at its stability limit, where the
    embedded error estimate is round-off amplified ~1e8 x: pow() of libm here and of OCML there differ in the last bit, a
    few steps later the sequences do, and the results agree to the plan's tolerance (1e-6), not to round-off -- the
    multiply-only form (eq_exponent == 2, the reference's default) has a bit-identical twin and is held to round-off
    like every other model (tests/helpers.py "adaptive parity")."""
    if "^" in key and integ in ("dopri5", "tsit5"):
        return 5e-6, 0.5
    if key == "crystallization" and integ in ROS:
        # moments from 1e-1 to 1e9 in one state vector: the difference-quotient Jacobian's last-bit noise (dJ/J ~ 1e-8)
        # passes through an LU of that conditioning; measured 1.1e-6 on single lanes of the full action box, identical
        # step sequences (the plan's tolerance is 1e-6; every other model stays below 5e-8)
        return 5e-6, 0.98
    return 1e-6, 0.98


def _make(p, B, **kw):
    from pcgym_amd import VecEnv

    try:
        return VecEnv(copy.deepcopy(p), n_envs=B, **kw)
    except ValueError as e:  # a combination the plan refuses by design
        pytest.skip(f"refused at plan creation: {str(e)[:100]}")


# ---- step kernels ---------------------------------------------------------------------------------------------------------
# dispatch:  auto     the library's own choice at this batch size
#            odd      the same with an odd batch (one env per lane in the lean kernels: EPL = 1)
#            classic  PCG_OPT_VARIANT 1: the one-env-per-lane general kernels
#            queue    PCG_OPT_VARIANT 5: the in-workgroup work queue whatever the model and batch (adaptive pairs)
#            lds      PCG_OPT_LDS_STAGES: DOPRI5 with the stage vectors in LDS (Model::FULL)
#            stream1/2  PCG_OPT_VARIANT 2 / 3: the persistent streaming kernels, one / two envs per lane (RK4, Model::FULL)
#            nostatus  the library's own choice for a caller that keeps no per-env status byte (the lean RK4 launches of the
#                      larger full models then take the streaming kernel)
DISPATCH = {"auto": {}, "odd": {}, "classic": {"variant": 1}, "queue": {"variant": 5}, "lds": {"lds_stages": True},
            "stream1": {"variant": 2, "track_status": False}, "stream2": {"variant": 3, "track_status": False},
            "nostatus": {"track_status": False}}


def _sweep_cases():
    out = []
    for key in MODEL_KEYS:
        model = key.partition("^")[0]
        integs = FIXED + ADAPT + (GUARDED if model == "cstr" else ())
        for integ in integs:
            for pe in (False, True):
                for feat in ("lean", "cons"):
                    ds = ["auto", "classic"]
                    if not pe and feat == "lean" and integ in FIXED and model in FULL:
                        ds.append("odd")
                    if integ in ("dopri5", "rodas4", "rodas5"):
                        ds.append("queue")
                    if integ == "dopri5" and model in FULL:
                        ds.append("lds")
                    if integ == "rk4" and model in FULL and not pe and feat == "lean":
                        ds += ["stream1", "nostatus"] + (["stream2"] if model in ("cstr", "four_tank") else [])
                    for d in ds:
                        out.append(pytest.param(key, integ, pe, feat, d, id=f"{key}-{integ}-{'per_env_t' if pe else 'lockstep'}-{feat}-{d}"))
    return out


@pytest.mark.parametrize("key,integ,pe,feat,dispatch", _sweep_cases())
def test_integrator_sweep(key, integ, pe, feat, dispatch):
    import torch
    from oracle import oracle as O
    from pcgym_amd._lib import PcgError

    B = 131 if dispatch == "odd" else 700 if dispatch in ("stream1", "stream2", "nostatus") else 260
    env = _make(_params(key, integ, feat), B, seed=3, per_env_t=pe, **DISPATCH[dispatch])
    spec = env.spec
    orc = O.OracleEnv(spec, B, seed=3, per_env_t=pe)
    env.reset(), orc.reset()
    rng = np.random.default_rng(1)
    worst, same = 0.0, 1.0
    for i in range(3):
        a = _actions(spec, rng, B)
        try:
            og, rg, dg, _, _ = env.step(torch.tensor(a, device=env.device))
        except PcgError as e:
            if i == 0 and e.status == -6 and dispatch in ("queue", "stream1", "stream2", "lds"):
                pytest.skip("this launch shape does not exist for the combination (PCG_E_UNSUPPORTED)")
            raise
        oc, rc, dc = orc.step(a)
        worst = max(worst, _worst(env.x.cpu().numpy(), orc.x))
        fin = np.isfinite(rc)
        assert fin.mean() >= 0.5, "most envs fail on both sides: the comparison says nothing"
        assert np.allclose(rg.cpu().numpy()[fin], rc[fin], rtol=1e-6, atol=1e-9 * (1 + np.max(np.abs(rc[fin]), initial=0)))
        assert np.array_equal(dg.cpu().numpy().astype(bool), dc.astype(bool))
        assert np.allclose(og.cpu().numpy().T[:, fin], oc[:, fin], rtol=1e-6, atol=1e-6 * max(1.0, np.max(np.abs(oc[:, fin]))))
        if spec.ncon:
            assert np.array_equal(env.viol.cpu().numpy().astype(bool), orc.viol.astype(bool))
        if env.nsteps is not None and orc.nsteps is not None:
            same = min(same, float(np.mean(np.all(env.nsteps.cpu().numpy() == orc.nsteps, axis=0))))
        env.x.copy_(torch.tensor(orc.x, device=env.device))  # the next step starts from a common state
    env.close()
    # measured over all 266 + 228 combinations on the round-5 build: worst 4.8e-8 (a difference-quotient Jacobian through
    # three adaptive steps), identical step sequences >= 0.992
    bar, seq = _bars(key, integ)
    assert worst <= bar, f"worst relative difference over every lane {worst:.2e}"
    assert same >= seq, f"identical step sequences on {same:.3f} of the lanes"


# ---- the Rosenbrock attempt on its own -----------------------------------------------------------------------------------
def _single_attempt_params(key, integ, dt):
    p = _params(key, integ)
    N = int(p["N"])
    p.update(tsim=dt * N, rtol=1.0, atol=1.0)
    return p


# env-step lengths (fractions of the scenario's dt) at which every lane's first attempt IS the env step: h = min(c h0, dt) =
# dt with h0 = 0.01 |x| / |f| in the tolerance-scaled norm, and at tolerance 1 any finite attempt is accepted.  Small ones,
# where W = I / (gamma h) - J is dominated by its diagonal, up to as large as the first-step rule allows (h |J| ~ 0.1 - 1 for
# the stiff models), where the factorisation, the pivoting and every stage's solve carry weight.
@pytest.mark.parametrize("entry", ["step", "integrate"])
@pytest.mark.parametrize("pe", [False, True], ids=["lockstep", "per_env_t"])
@pytest.mark.parametrize("integ", ROS)
@pytest.mark.parametrize("key", MODEL_KEYS)
def test_ros_single_attempt(key, integ, pe, entry):
    import torch
    from oracle import oracle as O

    if entry == "integrate" and pe:
        pytest.skip("pcg_integrate has no counter mode")
    B = 192
    base = SC.scenarios()[SCEN[key.partition("^")[0]]]["env_params"]
    dt0 = float(base["tsim"]) / int(base["N"])
    tested = 0
    for frac in (1e-6, 1e-4, 1e-3, 1e-2):
        dt = dt0 * frac
        p = _single_attempt_params(key, integ, dt)
        env = _make(p, B, seed=11, per_env_t=pe)
        spec = env.spec
        orc = O.OracleEnv(spec, B, seed=11, per_env_t=pe)
        env.reset(), orc.reset()
        rng = np.random.default_rng(5)
        # spread the start states (the scenarios start every env at one x0)
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
        # EVERY lane, every component.  The bar: 1e-12 of the component's scale, plus 5e-9 of the lane's own increment -- what
        # a difference-quotient Jacobian may legitimately turn a last-bit difference of exp / pow / sqrt into (libm here,
        # OCML there): dJ/J ~ ulp / sqrt(eps) ~ 1e-8, and the attempt passes dJ on as (gamma h |J|) |x' - x| dJ/J.  At the
        # small step sizes the increment is ~1e-6 of the state and the bar IS 1e-12; at the large ones a wrong stage
        # coefficient, a mis-restored spill or a stale pivot changes the increment by O(1), not by 1e-8 of itself.
        scale = np.maximum(np.abs(xo), 1e-6 * np.max(np.abs(xo), axis=1, keepdims=True))
        scale = np.maximum(scale, 1e-300)
        assert np.isfinite(xo).all() and np.isfinite(xg).all()
        err = np.abs(xg - xo) / scale
        inc = np.max(np.abs(xo - x0) / scale, axis=0, keepdims=True)
        bar = 1e-12 + 5e-9 * inc
        worst = float(np.max(err / bar))
        assert worst <= 1.0, (f"dt = {frac:g} x the scenario's: one attempt differs by {err.max():.2e} on some lane "
                              f"({worst:.2f} x the bar; the lanes' increments are {inc.min():.1e} .. {inc.max():.1e})")
        # ... and the attempt moved the state: the comparison is not x == x
        assert np.max(np.abs(xo - x0)) > 0
    assert tested >= 1, "no single-attempt configuration for this model"


# ---- pcg_integrate ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lds", [False, True], ids=["regs", "lds_stages"])
@pytest.mark.parametrize("integ", FIXED + ADAPT + GUARDED)
@pytest.mark.parametrize("key", MODEL_KEYS)
def test_integrate_sweep(key, integ, lds):
    import torch
    from oracle import oracle as O
    from pcgym_amd import _abi as abi

    model = key.partition("^")[0]
    if integ in GUARDED and model != "cstr":
        pytest.skip("guarded schemes exist for models with a guard hook")
    if lds and not (integ == "dopri5" and model in FULL):
        pytest.skip("stage store only exists for dopri5 on the full models")
    B = 200
    env = _make(_params(key, integ, "lean"), B, seed=13)
    spec = env.spec
    if lds:
        assert env._lib.pcg_plan_set_option(env._plan, abi.PCG_OPT_LDS_STAGES, 1) == 0
    orc = O.OracleEnv(spec, B, seed=13)
    orc.reset()
    rng = np.random.default_rng(9)
    x0 = orc.x * (1.0 + 0.1 * rng.uniform(-1, 1, orc.x.shape))
    a = _actions(spec, rng, B)
    u = np.zeros((spec.nu, B))
    u[:spec.na] = a if not spec.normalise_a else (a + 1) * (spec.a_high - spec.a_low)[:, None] / 2 + spec.a_low[:, None]
    for j, name in enumerate(spec.model.disturbances[: spec.nu - spec.na]):
        u[spec.na + j] = float(spec.model.parameters[name])
    xo, no = O.integrate(spec, x0, u)
    xt, ut = torch.tensor(x0, device=env.device), torch.tensor(u, device=env.device)
    nt = torch.zeros((2, B), dtype=torch.int32, device=env.device)
    assert env._lib.pcg_integrate(env._plan, B, xt.data_ptr(), ut.data_ptr(), nt.data_ptr(), None) == 0
    torch.cuda.synchronize()
    env.close()
    assert _worst(xt.cpu().numpy(), xo) <= 1e-6
    if integ not in FIXED:
        assert np.mean(np.all(nt.cpu().numpy() == no, axis=0)) >= 0.98


# ---- the other entry points of a plan ----------------------------------------------------------------------------------------
def _close(a, b):
    import torch

    a, b = a.double(), b.double()
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    if not torch.equal(fa, fb):
        return float("inf")
    if not fa.any():
        return 0.0
    return ((a[fa] - b[fa]).abs() / b[fa].abs().clamp_min(1e-9)).max().item()


def _shape_cases():
    out = []
    for key in MODEL_KEYS:
        model = key.partition("^")[0]
        for integ in FIXED + ADAPT + (GUARDED if model == "cstr" else ()):
            for feat in ("lean", "cons"):
                ds = ["auto"]
                if feat == "lean":
                    ds.append("classic")
                if feat == "lean" and integ in FIXED and model in FULL:
                    ds.append("odd")
                if integ == "dopri5" and model in FULL and feat == "lean":
                    ds.append("lds")
                for d in ds:
                    out.append(pytest.param(key, integ, feat, d, id=f"{key}-{integ}-{feat}-{d}"))
    return out


@pytest.mark.parametrize("key,integ,feat,dispatch", _shape_cases())
def test_shape_sweep(key, integ, feat, dispatch):
    """rollout (T steps, state in registers) == T step launches; a HIP graph of the T steps == the launches, bitwise;
    same-launch auto-reset through an episode end == step + reset -- and the step launches against the oracle over the
    same T steps."""
    import torch
    from oracle import oracle as O
    from pcgym_amd._lib import PcgError

    B, T = (201 if dispatch == "odd" else 200), 5
    model = key.partition("^")[0]
    base = SC.scenarios()[SCEN[model]]["env_params"]
    p = _params(key, integ, feat)
    p.update(N=T + 2, tsim=float(base["tsim"]) * (T + 2) / base["N"])
    for k in ("SP", "disturbances"):
        if p.get(k):
            p[k] = {kk: list(np.asarray(v, dtype=float)[: T + 2]) for kk, v in p[k].items()}
    kw = DISPATCH[dispatch]
    e_step, e_roll, e_graph = (_make(p, B, seed=5, **kw) for _ in range(3))
    ar = _make(p, B, seed=5, auto_reset=True, **kw)
    ref = _make(p, B, seed=5, **kw)
    spec = e_step.spec
    orc = O.OracleEnv(spec, B, seed=5)
    gen = torch.Generator(device="cuda").manual_seed(7)
    acts = 2 * torch.rand((T + 4, spec.na, B), generator=gen, device="cuda", dtype=torch.float64) - 1
    if not spec.normalise_a:
        lo = torch.tensor(spec.a_low, device="cuda")[None, :, None]
        hi = torch.tensor(spec.a_high, device="cuda")[None, :, None]
        acts = (acts + 1) * (hi - lo) / 2 + lo
    for e in (e_step, e_roll, e_graph, ar, ref, orc):
        e.reset()
    obs_s, rew_s = [], []
    for i in range(T):
        o, r, d, _, _ = e_step.step(acts[i])
        obs_s.append(e_step.obs_soa.clone()), rew_s.append(r.clone())
        orc.step(acts[i].cpu().numpy())
        assert _worst(e_step.x.cpu().numpy(), orc.x) <= 10 * _bars(key, integ)[0], f"step {i}: the step launches leave the oracle"
        orc.x[:] = e_step.x.cpu().numpy()  # one-step comparisons: unstable models amplify round-off from step to step
    try:
        oq, rq = e_roll.rollout(acts[:T], collect_obs=True, collect_rew=True)
        rolled = True
    except PcgError as e:  # no fused rollout for this integrator
        assert e.status == -6, e
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
    orc2 = O.OracleEnv(spec, B, seed=5)
    orc2.reset()
    for i in range(T + 2):  # auto-reset through the episode end (N - 1 = T + 1 steps), then one step of the next episode
        o, r, d, _, _ = ar.step(acts[i])
        if ref.t == ref.N - 1:
            ref.reset(), orc2.reset()
        o2, r2, d2, _, _ = ref.step(acts[i])
        _, r3, _ = orc2.step(acts[i].cpu().numpy())
        assert _close(r, r2) == 0.0 and torch.equal(d, d2), f"auto-reset step {i}: reward / done differ"
        fin = np.isfinite(r3)
        assert np.allclose(r.cpu().numpy()[fin], r3[fin], rtol=1e-4, atol=1e-6 * (1 + np.max(np.abs(r3[fin]), initial=0)))
        if ref.t != ref.N - 1:  # (at the episode end `ar` already holds the NEW x0)
            assert _close(ar.x, ref.x) == 0.0, f"auto-reset step {i}: state differs"
            assert _worst(ar.x.cpu().numpy(), orc2.x) <= 10 * _bars(key, integ)[0]
            orc2.x[:] = ar.x.cpu().numpy()
    for e in (e_step, e_roll, e_graph, ar, ref):
        e.close()


# ---- per-env parameters ------------------------------------------------------------------------------------------------------
def _unc_params(key, integ):
    from pcgym_amd.models import get_model

    model = key.partition("^")[0]
    mi = get_model(model)
    if mi.affine_builder is not None:
        pytest.skip("affine registry models have no per-env parameter kernel")
    names = [k for k, v in mi.parameters.items() if float(v) != 0.0 and k not in ("N", "eq_exponent")]
    pick = names[:2]
    p = _params(key, integ, "lean")
    p.update(uncertainty_percentages={k: 0.03 for k in pick}, distribution="uniform",
             uncertainty_bounds={"low": np.array([min(0.9 * mi.parameters[k], 1.1 * mi.parameters[k]) for k in pick]),
                                 "high": np.array([max(0.9 * mi.parameters[k], 1.1 * mi.parameters[k]) for k in pick])})
    return p


@pytest.mark.parametrize("pe", [False, True], ids=["lockstep", "per_env_t"])
@pytest.mark.parametrize("integ", ["rk4", "dopri5"])
@pytest.mark.parametrize("key", MODEL_KEYS)
def test_uncertainty_sweep(key, integ, pe):
    """row f-3 for every model: per-env parameters sampled at reset (pcgym.py:212-253, 301-316), the dynamics use each
    env's own values: step kernels and the fused rollout against the oracle"""
    import torch
    from oracle import oracle as O

    B, T = 160, 3
    p = _unc_params(key, integ)
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
        assert _worst(env.x.cpu().numpy(), orc.x) <= (1e-8 if "^" not in key else 5e-6)
        assert np.allclose(env.obs_soa.cpu().numpy(), orc.obs, rtol=1e-8 if "^" not in key else 1e-5, atol=1e-9 if "^" not in key else 1e-5)
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
        assert _worst(env2.x.cpu().numpy(), orc2.x) <= (1e-7 if "^" not in key else 2e-5)
        env2.close()
    env.close()


# ---- feature-masked pipelined kernels (pcg_step_feat.hpp: RK4 plans of the two small models) -----------------------------------
# name -> (env_params changes, VecEnv arguments, pass the `viol` buffer although no constraint is configured)
def _feat_sets(model):
    p0 = SC.scenarios()[SCEN[model]]["env_params"]
    c0 = float(np.asarray(p0["x0"], dtype=float)[0])
    cons = dict(constraints=lambda x, u, c0=c0: np.array([x[0] - c0]).reshape(-1,), done_on_cons_vio=False, r_penalty=True)
    track = dict(custom_reward={"kind": "sp_track", "R": 0.05})
    return {
        "viol_only": ({}, {}, True),                    # mask 0: the lean step with the `viol` output
        "viol_autoreset": ({}, {"auto_reset": True}, True),  # FT_AR
        "cons": (cons, {}, False),                      # FT_CONS
        "track": (track, {}, False),                    # FT_TRACK
        "cons_track": ({**cons, **track}, {}, False),   # FT_CONS | FT_TRACK (the constraint-showcase configuration)
        "a_delta": ({}, {}, False),                     # FT_ALL (the only mask with FT_ADELTA)
    }


@pytest.mark.parametrize("fs", ["viol_only", "viol_autoreset", "cons", "track", "cons_track", "a_delta"])
@pytest.mark.parametrize("model", ["cstr", "four_tank"])
def test_feature_kernels(model, fs):
    import torch
    from oracle import oracle as O

    B = 512
    over, kw, viol = _feat_sets(model)[fs]
    p = _params(model, "rk4", "lean")
    p.update(over)
    if fs == "a_delta":
        a_lo, a_hi = np.asarray(p["a_space"]["low"], dtype=float), np.asarray(p["a_space"]["high"], dtype=float)
        p.update(a_delta=True, a_0=(a_lo + a_hi) / 2, a_space_act={"low": a_lo, "high": a_hi},
                 a_space={"low": -(a_hi - a_lo) / 20, "high": (a_hi - a_lo) / 20}, normalise_a=True)
    env = _make(p, B, seed=17, **kw)
    spec = env.spec
    if viol:
        env._buf.viol = env.viol.data_ptr()
    orc = O.OracleEnv(spec, B, seed=17)
    env.reset(), orc.reset()
    rng = np.random.default_rng(4)
    for i in range(spec.N + 1 if kw.get("auto_reset") else 4):
        a = _actions(spec, rng, B)
        if kw.get("auto_reset") and orc.t == spec.N - 1:
            orc.reset()
        og, rg, dg, _, _ = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
        assert np.allclose(rg.cpu().numpy(), rc, rtol=1e-9, atol=1e-9 * (1 + np.abs(rc).max()))
        assert np.array_equal(dg.cpu().numpy().astype(bool), dc.astype(bool))
        if not (kw.get("auto_reset") and orc.t == spec.N - 1):  # (at the episode end the env already holds the new x0)
            assert _worst(env.x.cpu().numpy(), orc.x) <= 1e-11
        if spec.ncon:
            assert np.array_equal(env.viol.cpu().numpy().astype(bool), orc.viol.astype(bool))
            assert np.allclose(env.g.cpu().numpy(), orc.g, rtol=1e-10, atol=1e-10)
    env.close()


# ---- launch shapes of the work queue that only large batches take -------------------------------------------------------------
@pytest.mark.parametrize("pe", [False, True], ids=["lockstep", "per_env_t"])
@pytest.mark.parametrize("integ,B", [("dopri5", 1 << 18), ("rodas4", 70_000), ("rodas5", 70_000), ("rodas4", 1 << 18),
                                     ("rodas5", 1 << 18)])
@pytest.mark.parametrize("key", ["multistage_extraction", "multistage_extraction^1.5"])
def test_queue_shapes(key, integ, B, pe):
    """the extraction cascade at the batch sizes that select: 512-thread workgroups (DOPRI5 from 229,376 envs), ONE
    workgroup per CU on the register-only instantiation (the Rosenbrock pairs between 65,536 and 300,000 envs) and two
    workgroups per CU beyond -- one env step of the full batch, three windows of it against the oracle"""
    import torch
    from oracle import oracle as O

    p = _params(key, integ, "lean")
    env = _make(p, B, seed=31, per_env_t=pe)
    spec = env.spec
    env.reset()
    gen = torch.Generator(device="cuda").manual_seed(3)
    a = 2 * torch.rand((spec.na, B), generator=gen, device="cuda", dtype=torch.float64) - 1
    x0 = env.x.clone()
    env.step(a)
    assert int(env.status.sum().item()) == 0
    W = 192
    for lo in (0, B // 2 - 77, B - W):
        orc = O.OracleEnv(spec, W, seed=31, per_env_t=pe, env_offset=lo)
        orc.reset()
        assert np.allclose(orc.x, x0[:, lo:lo + W].cpu().numpy(), rtol=1e-14)
        orc.step(a[:, lo:lo + W].cpu().numpy())
        bar, seq = _bars(key, integ)
        assert _worst(env.x[:, lo:lo + W].cpu().numpy(), orc.x) <= (bar if bar > 1e-6 else 1e-7)
        assert np.mean(np.all(env.nsteps[:, lo:lo + W].cpu().numpy() == orc.nsteps, axis=0)) >= seq
    env.close()


@pytest.mark.parametrize("pe", [False, True], ids=["lockstep", "per_env_t"])
@pytest.mark.parametrize("integ", GUARDED)
def test_guarded_fixup_shapes(integ, pe):
    """the guarded plans of the cstr in their two-launch form (from 65,536 envs: the general kernel marks the envs it does
    not trust, the work-queue kernel of the adaptive pair finishes exactly those) on the full x0 box of SURVEY.md section
    8(d), a third of which ignites: two env steps of the full batch, windows of it against the oracle"""
    import torch
    from oracle import oracle as O

    B = 1 << 17
    p = _params("cstr", integ, "lean")
    p.update(x0=np.array([0.85, 330.0, 0.85]), uncertainty_percentages={"x0": [0.15 / 0.85, 20.0 / 330.0]},
             distribution="uniform")
    env = _make(p, B, seed=41, per_env_t=pe)
    spec = env.spec
    env.reset()
    gen = torch.Generator(device="cuda").manual_seed(5)
    W = 256
    for _ in range(2):
        a = 2 * torch.rand((spec.na, B), generator=gen, device="cuda", dtype=torch.float64) - 1
        x0 = env.x.clone()
        env.step(a)
        assert int(env.status.sum().item()) == 0
        assert float((env.nsteps.sum(dim=0) > 0).double().mean().item()) > 0.05  # the fix-up launch had envs to finish
        for lo in (0, B // 2 - 77, B - W):
            orc = O.OracleEnv(spec, W, seed=41, per_env_t=pe, env_offset=lo)
            orc.reset()
            orc.x[:] = x0[:, lo:lo + W].cpu().numpy()
            orc.step(a[:, lo:lo + W].cpu().numpy())
            assert _worst(env.x[:, lo:lo + W].cpu().numpy(), orc.x) <= 1e-9
            assert np.array_equal(env.nsteps[:, lo:lo + W].cpu().numpy(), orc.nsteps)
    env.close()
