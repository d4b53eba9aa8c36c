"""GPU: the reference-side binding (pcgym_amd.reference_engine.hip_integration_engine) at the reference's own plug
point -- ``integration_engine(make_env, env_params).casadi_step(state, uk)["xf"].full()`` / ``.jax_step(state, uk)``
(integrator.py:19-107, pcgym.py:281,423-429) -- checked against what the reference itself produced: the do-mpc / CVODES
trajectories it ships, and the ``make_env`` recordings in tests/golden (state before, action -> state after)."""
import copy

import numpy as np
import pytest

import helpers as H
import scenarios as SC
from pcgym_amd import models as M

pytestmark = pytest.mark.gpu

PAPER = [("cstr", "cstr"), ("four_tank", "four_tank"), ("multistage_extraction", "multistage_extraction"),
         ("crystallization", "crystallization"), ("cstr_constraint", "cstr")]


def _params(model, dt, nu, **kw):
    mi = M.get_model(model)
    nx = len(mi.states)
    p = {"model": model, "N": 10, "tsim": 10 * dt, "x0": np.ones(nx), "normalise_a": False, "normalise_o": False,
         "a_space": {"low": -np.ones(len(mi.inputs)), "high": np.ones(len(mi.inputs))},
         "o_space": {"low": -np.ones(nx), "high": np.ones(nx)}, "reward_states": [], "maximise_reward": True,
         "integration_method": "hip"}
    if nu > len(mi.inputs):
        p["disturbances"] = {k: np.zeros(10) for k in mi.disturbances}
        p["disturbance_bounds"] = {"low": -np.ones(len(mi.disturbances)), "high": np.ones(len(mi.disturbances))}
    p.update(kw)
    return p


@pytest.mark.parametrize("fix,model", PAPER)
def test_engine_replays_the_trajectories_the_reference_ships(fix, model):
    """x[:, i] = F(x[:, i-1], u[:, i]; dt) for i >= 2, one env per call like make_env.step"""
    from pcgym_amd import hip_integration_engine

    g = H.gold("paper_" + fix)
    x, u, dt = g["x"], g["u"], float(g["dt"])
    nx = len(M.get_model(model).states)
    eng = hip_integration_engine(None, _params(model, dt, u.shape[0], **H.TIGHT[model]))
    assert eng.nx == nx and eng.nu == u.shape[0]
    want = x[:nx, 2:]
    scale = np.maximum(np.abs(want), 1e-6 * np.max(np.abs(want), axis=1, keepdims=True))
    for i in range(2, x.shape[1]):
        state = np.concatenate([x[:nx, i - 1], [0.123]])  # make_env passes its whole state vector [x | SP | d]
        xf = eng.casadi_step(state, u[:, i])["xf"].full()
        assert xf.shape == (nx, 1)
        assert np.max(np.abs(xf[:, 0] - x[:nx, i]) / scale[:, i - 2]) <= 5e-8, (fix, i)
        if i % 7 == 0:
            xj = eng.jax_step(state, u[:, i])
            assert xj.shape == (nx,) and np.array_equal(xj, xf[:, 0])


def _plain_scenarios():
    """recorded scenarios whose uk is the mapped action alone (no disturbance inputs, no action increments) and whose
    model has inputs -- the cases where uk can be rebuilt from the recording without re-implementing make_env.step"""
    out = []
    for k, v in SC.scenarios().items():
        p = v["env_params"]
        if p.get("model") is None or p.get("disturbances") is not None or p.get("a_delta") or "ref_env_params" in v:
            continue
        if not M.get_model(p["model"]).inputs:
            continue
        out.append(k)
    return sorted(out)


@pytest.mark.parametrize("name", _plain_scenarios())
def test_engine_reproduces_make_env_recordings(name):
    """the recorded reference run: state[i] --(action map, pcgym.py:371-379)--> uk --engine--> state[i+1][:nx]"""
    from pcgym_amd import hip_integration_engine
    from pcgym_amd.config import EnvSpec

    sc = SC.scenarios()[name]
    g = H.gold("step_" + name)
    p = copy.deepcopy(sc["env_params"])
    p.update(H.tight_for(p))
    spec = EnvSpec(p)
    assert spec.nu == spec.na and not spec.a_delta  # scenarios without disturbance inputs / increments
    eng = hip_integration_engine(None, p)
    st, acts = g["state"], g["actions"]
    for i in range(acts.shape[0]):
        a = acts[i]
        uk = (a + 1) * (spec.a_high - spec.a_low) / 2 + spec.a_low if spec.normalise_a else a
        xf = eng.casadi_step(st[i], uk)["xf"].full()[:, 0]
        want = st[i + 1][: spec.nx]
        assert np.all(np.abs(xf - want) <= 2e-9 * np.maximum(np.abs(want), 1e-3) + 5e-10), (name, i)


def test_engine_argument_and_failure_behaviour():
    from pcgym_amd import hip_integration_engine

    p = _params("multistage_extraction", 1.0, 2, integrator="dopri5", rtol=1e-8, atol=1e-8, max_steps=5)
    eng = hip_integration_engine(None, p)
    x = np.full(10, 0.3)
    with pytest.raises(ValueError, match="uk"):
        eng.casadi_step(x, np.array([5.0]))
    with pytest.raises(ValueError, match="state"):
        eng.casadi_step(x[:4], np.array([5.0, 10.0]))
    with pytest.raises(RuntimeError, match="integration failed"):  # 5 steps are not enough at |lambda| dt ~ 240
        eng.casadi_step(x, np.array([500.0, 1000.0]))
    p["max_steps"] = 100000
    eng2 = hip_integration_engine(None, p)  # a different plan (keyed by the numeric configuration)
    assert np.isfinite(eng2.jax_step(x, np.array([500.0, 1000.0]))).all()
    with pytest.raises(ValueError):
        hip_integration_engine(None, None)


def test_engine_with_a_python_custom_model():
    """the reference hands its engine the env_params it was given, custom_model object included (integrator.py:19-31): a
    non-affine Python model is traced and compiled, and the engine integrates it like the reference's CVODES would"""
    from scipy.integrate import solve_ivp

    from pcgym_amd import hip_integration_engine
    from test_gpu_user_model import _ChemostatObject, _chemostat_params

    p = _chemostat_params(integrator="dopri5", rtol=1e-10, atol=1e-12)
    p["custom_model"] = _ChemostatObject()
    eng = hip_integration_engine(None, p)
    rng = np.random.default_rng(3)
    for _ in range(5):
        x = np.array([rng.uniform(0.2, 2.0), rng.uniform(0.05, 3.0)])
        u = np.array([rng.uniform(0.0, 0.45)])
        xf = eng.casadi_step(np.concatenate([x, [1.4]]), u)["xf"].full()[:, 0]
        r = solve_ivp(lambda t, y: _ChemostatObject()(y, u), (0.0, eng.spec.dt), x, method="LSODA", rtol=1e-12, atol=1e-14)
        assert np.all(np.abs(xf - r.y[:, -1]) <= 1e-8 * np.maximum(np.abs(r.y[:, -1]), 1e-3))
