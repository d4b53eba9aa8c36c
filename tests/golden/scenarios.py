"""Scenario catalogue shared by the golden-vector generator and the parity tests.

Each scenario is an ``env_params`` dict in the reference's own vocabulary
(/root/reference/src/pcgym/pcgym.py:32-253 consumes exactly these keys) plus a
scripted action sequence.  ``gen_golden.py`` feeds the dict to the *reference*
``make_env`` (run in the build container only); the tests feed the very same
dict to this repo's ``make_env`` and compare against the committed fixtures.

Configs are the reference's canonical ones:
  cstr        README.md:16-55, pc-gym_paper/train_policies/cstr/cstr_train.py:12-47
  four_tank   pc-gym_paper/train_policies/four_tank/4tank_train.py:54-84
  ME          pc-gym_paper/train_policies/multistage_extraction/me_train.py:55-91
  cryst       pc-gym_paper/train_policies/crystalisation/cryst_train.py:50-99
  custom      tests/environment/test_make_env_custom_model.py:66-86 (the only KAT)

This file is this repo's own code: it contains no reference source.
"""
from __future__ import annotations

import numpy as np


def _thirds(n, a, b, c):
    k = n // 3
    return [a] * k + [b] * k + [c] * (n - 2 * k)


def _halves(n, a, b):
    k = n // 2
    return [a] * k + [b] * (n - k)


# ----------------------------------------------------------------------------
# constraint callables in the reference's style g(x,u) <= 0
# (docs/guides/constraints.md:35-51)
# ----------------------------------------------------------------------------
def cons_cstr_T(x, u):
    return np.array([319 - x[1], x[1] - 331]).reshape(-1,)


def cons_cstr_T_u(x, u):
    # state and input rows mixed
    return np.array([x[1] - 336.0, 295.2 - u[0], 0.5 * x[0] + 0.001 * x[1] - 0.8]).reshape(-1,)


def cons_me(x, u):
    return np.array([x[8] - 0.45, 0.2 - x[0]]).reshape(-1,)


def cons_cstr_nonaffine(x, u):
    """a quadratic band on T and a curved Ca floor: NOT affine in (x, u) -- on our side the same two rows are C
    expressions compiled into the step kernel (env_params['constraints'] = {'expr': [...]})"""
    return np.array([0.02 * (x[1] - 325.0) ** 2 - 0.5, 0.84 - x[0] - 5e-4 * (u[0] - 298.0) ** 2]).reshape(-1,)


def cons_cstr_nonaffine_q3(x, u):
    """with normalise_o / normalise_a the reference hands the callable a re-"de-normalised" state and input (quirk Q3,
    pcgym.py:597-608): T = 327 K arrives as 8508 -- the bounds are written for THOSE numbers"""
    return np.array([1e-6 * (x[1] - 8500.0) ** 2 - 0.004, 1.0e3 - 1.0e-2 * u[0] ** 2 + x[0]]).reshape(-1,)


def reward_cstr_exp(self, x, u, con):
    """custom_reward(self, obs, uk, violated) protocol (pcgym.py:470-471): tracking + an exponential temperature cost
    + a violation charge -- outside the declarative sp_track family; on our side a C expression"""
    return float(-1e3 * (x[0] - self.SP["Ca"][self.t]) ** 2 - 0.05 * np.exp(0.1 * (x[1] - 330.0)) - (5.0 if con else 0.0))


class LinearCustomModel:
    """Shape of tests/environment/test_make_env_custom_model.py:7-25 (written
    from the protocol description, pcgym.py:150-153: __call__(x,u), info(),
    attribute int_method)."""

    def __init__(self, p1=1.0, p2=2.0):
        self.int_method = "casadi"
        self.p1 = p1
        self.p2 = p2

    def __call__(self, x, u):
        return np.array([self.p1 * x[0] + u[0], self.p2 * x[1]])

    def info(self):
        return {
            "parameters": {"p1": self.p1, "p2": self.p2},
            "states": ["x1", "x2"],
            "inputs": ["u1"],
            "disturbances": [],
        }


def _cstr_base(N=60, tsim=26.0):
    return {
        "N": N,
        "tsim": tsim,
        "SP": {"Ca": _thirds(N, 0.85, 0.9, 0.87)},
        "o_space": {"low": np.array([0.7, 300, 0.8]), "high": np.array([1, 350, 0.9])},
        "a_space": {"low": np.array([295]), "high": np.array([302])},
        "x0": np.array([0.8, 330, 0.8]),
        "r_scale": {"Ca": 1e3},
        "model": "cstr",
        "normalise_a": True,
        "normalise_o": True,
    }


def _four_tank_base(N=60, tsim=1000.0):
    return {
        "N": N,
        "tsim": tsim,
        "SP": {"h3": _halves(N, 0.5, 0.1), "h4": _halves(N, 0.2, 0.3)},
        "o_space": {"low": np.array([0.0] * 6), "high": np.array([0.6] * 6)},
        "a_space": {"low": np.array([0.1, 0.1]), "high": np.array([10.0, 10.0])},
        "x0": np.array([0.141, 0.112, 0.072, 0.42, 0.5, 0.2]),
        "r_scale": {"h3": 1e3, "h4": 1e3},
        "model": "four_tank",
        "normalise_a": True,
        "normalise_o": True,
    }


def _me_base(N=60, tsim=60.0):
    return {
        "N": N,
        "tsim": tsim,
        "SP": {"X5": _thirds(N, 0.3, 0.4, 0.3)},
        "o_space": {"low": np.array([0.0] * 10 + [0.3]), "high": np.array([1.0] * 10 + [0.4])},
        "a_space": {"low": np.array([5.0, 10.0]), "high": np.array([500.0, 1000.0])},
        "x0": np.array([0.55, 0.3, 0.45, 0.25, 0.4, 0.20, 0.35, 0.15, 0.25, 0.1, 0.3]),
        "r_scale": {"X5": 1e2},
        "model": "multistage_extraction",
        "normalise_a": True,
        "normalise_o": True,
    }


def _cryst_base(N=30, tsim=30.0):
    mu = [1478.00986666666, 22995.8230590611, 1800863.24079725, 248516167.940593]
    cv0 = float(np.sqrt(mu[2] * mu[0] / (mu[1] ** 2) - 1))
    ln0 = mu[1] / (mu[0] + 1e-6)
    return {
        "N": N,
        "tsim": tsim,
        "SP": {"CV": [1.0] * N, "Ln": [15.0] * N},
        "o_space": {
            "low": np.array([0, 0, 0, 0, 0, 0, 0, 0.9, 14.0]),
            "high": np.array([1e20, 1e20, 1e20, 1e20, 0.5, 2, 20, 1.1, 16.0]),
        },
        "a_space": {"low": np.array([-1.0]), "high": np.array([1.0])},
        "a_space_act": {"low": np.array([10.0]), "high": np.array([40.0])},
        "x0": np.array(mu + [0.15861523304, cv0, ln0, 1.0, 15.0]),
        "model": "crystallization",
        "normalise_a": True,
        "normalise_o": True,
        "a_0": 39.0,
        "a_delta": True,
    }


def _me_reactive_base(N=40, tsim=40.0):
    x0 = []
    for s in range(5):
        x0 += [1.6 - 0.25 * s, 0.05 + 0.01 * s, 1.2 + 0.1 * s, 0.02 * (s + 1)]
    return {
        "N": N,
        "tsim": tsim,
        "SP": {"XA5": _halves(N, 0.6, 0.8)},
        "o_space": {"low": np.array([0.0] * 20 + [0.0]), "high": np.array([3.0] * 20 + [2.0])},
        "a_space": {"low": np.array([5.0, 10.0]), "high": np.array([50.0, 100.0])},
        "x0": np.array(x0 + [0.6]),
        "model": "multistage_extraction_reactive",
        "normalise_a": True,
        "normalise_o": True,
    }


def scenarios():
    """name -> dict(env_params=..., steps=T, action_seed=..., notes=...)."""
    S = {}

    # 1. BASELINE.json configs[0]: README quick-start, 99-step rollout
    S["cstr_quickstart"] = dict(
        env_params={
            "N": 100,
            "tsim": 25,
            "SP": {"Ca": _halves(100, 0.85, 0.9)},
            "o_space": {"low": np.array([0.7, 300, 0.8]), "high": np.array([1, 350, 0.9])},
            "a_space": {"low": np.array([295]), "high": np.array([302])},
            "x0": np.array([0.8, 330, 0.8]),
            "model": "cstr",
        },
        steps=99,
        action_seed=0,
    )
    S["cstr_canonical"] = dict(env_params=_cstr_base(), steps=59, action_seed=1)

    p = _cstr_base()
    p.update(normalise_a=False, normalise_o=False)
    S["cstr_raw"] = dict(env_params=p, steps=59, action_seed=2, raw_actions=True)

    # constraints: quirk Q3 (normalised state fed to g) + Q4 penalty per SP key
    p = _cstr_base()
    p.update(constraints=cons_cstr_T, done_on_cons_vio=False, r_penalty=True)
    S["cstr_cons_pen_norm"] = dict(env_params=p, steps=59, action_seed=3)

    p = _cstr_base()
    p.update(normalise_a=False, normalise_o=False, constraints=cons_cstr_T,
             done_on_cons_vio=False, r_penalty=True)
    S["cstr_cons_pen_raw"] = dict(env_params=p, steps=59, action_seed=4, raw_actions=True)

    p = _cstr_base()
    p.update(normalise_a=False, normalise_o=False, constraints=cons_cstr_T_u,
             done_on_cons_vio=True, r_penalty=False)
    S["cstr_cons_done_raw"] = dict(env_params=p, steps=59, action_seed=5, raw_actions=True)

    # disturbances (docs/guides/disturbances.md:5-17)
    p = _cstr_base()
    N = p["N"]
    p.update(
        disturbances={"Ti": np.repeat([350.0, 345.0, 350.0], [N // 4, N // 2, N - N // 4 - N // 2])},
        disturbance_bounds={"low": np.array([320.0]), "high": np.array([360.0])},
    )
    S["cstr_dist_Ti"] = dict(env_params=p, steps=58, action_seed=6)

    p = _cstr_base()
    p.update(
        normalise_o=False,
        disturbances={
            "Caf": np.linspace(0.95, 1.05, N),
            "Ti": 350.0 + 3.0 * np.sin(np.arange(N) / 5.0),
        },
        disturbance_bounds={"low": np.array([320.0, 0.9]), "high": np.array([360.0, 1.1])},
    )
    S["cstr_dist_both"] = dict(env_params=p, steps=58, action_seed=7)

    S["four_tank_canonical"] = dict(env_params=_four_tank_base(), steps=59, action_seed=8)
    S["me_canonical"] = dict(env_params=_me_base(), steps=59, action_seed=9,
                             action_scale=0.25, action_shift=-0.7)

    p = _me_base()
    N = p["N"]
    p.update(
        disturbances={"X0": np.repeat([0.6, 0.7, 0.6], [N // 3, N // 3, N - 2 * (N // 3)])},
        disturbance_bounds={"low": np.array([0.5]), "high": np.array([0.8])},
        constraints=cons_me, done_on_cons_vio=False, r_penalty=False,
        normalise_o=False, normalise_a=False,  # normalise_a + disturbances + constraints
        # raises in the reference itself (pcgym.py:597-600 broadcasts Nu against na)
    )
    S["me_dist_cons"] = dict(env_params=p, steps=58, action_seed=10,
                             action_scale=0.25, action_shift=-0.7, raw_actions=True)

    S["cryst_adelta"] = dict(env_params=_cryst_base(), steps=29, action_seed=11,
                             action_scale=0.3, action_shift=-0.2)
    S["me_reactive"] = dict(env_params=_me_reactive_base(), steps=39, action_seed=12)

    # terminal ("batch") reward path, pcgym.py:502-532: no SP, reward_states
    p = _cstr_base()
    del p["SP"]
    del p["r_scale"]
    p.update(
        o_space={"low": np.array([0.7, 300]), "high": np.array([1, 350])},
        x0=np.array([0.8, 330.0]),
        reward_states=["Ca", "T"],
        maximise_reward=False,
        r_scale={"T": 0.01},
        N=20,
        tsim=20 * 26.0 / 60.0,
    )
    S["cstr_batch_reward"] = dict(env_params=p, steps=19, action_seed=13)

    # partial observation (pcgym.py:344-347, 495-498)
    p = _cstr_base()
    p.update(partial_observation=["Ca"])
    S["cstr_partial_obs"] = dict(env_params=p, steps=20, action_seed=14)

    # ---- "next" row f-2 models ------------------------------------------------------------------
    S["complex_cstr_sp"] = dict(env_params={
        "N": 40, "tsim": 40 * 26.0 / 60.0, "SP": {"Cb": _halves(40, 0.3, 0.4)},
        "o_space": {"low": np.array([0.0, 0.0, 0.0, 300.0, 0.0]), "high": np.array([1.0, 1.0, 1.0, 350.0, 1.0])},
        "a_space": {"low": np.array([295.0]), "high": np.array([302.0])},
        "x0": np.array([0.8, 0.1, 0.05, 325.0, 0.3]), "r_scale": {"Cb": 1e2}, "model": "complex_cstr"},
        steps=39, action_seed=21)
    S["photo_batch_reward"] = dict(env_params={   # terminal ("batch") reward, maximise the product c_q
        "N": 12, "tsim": 240, "reward_states": ["c_q"], "maximise_reward": True,
        "o_space": {"low": np.array([0.0, 0.0, 0.0]), "high": np.array([10.0, 1000.0, 0.05])},
        "a_space": {"low": np.array([120.0, 0.0]), "high": np.array([400.0, 40.0])},
        "x0": np.array([1.0, 150.0, 0.0]), "model": "photobioreactor"},
        steps=11, action_seed=22)
    S["distillation_sp"] = dict(env_params={
        "N": 30, "tsim": 30.0, "SP": {"X0": _halves(30, 0.95, 0.9)},
        "o_space": {"low": np.array([0.0] * 9 + [0.8]), "high": np.array([1.0] * 9 + [1.0])},
        "a_space": {"low": np.array([1.0, 150.0]), "high": np.array([5.0, 400.0])},
        "x0": np.array([0.95, 0.9, 0.8, 0.65, 0.5, 0.35, 0.2, 0.1, 0.05, 0.95]), "r_scale": {"X0": 1e2},
        "model": "distillation_column"},
        steps=29, action_seed=23)
    S["first_order_sp"] = dict(env_params={
        "N": 30, "tsim": 6.0, "SP": {"x": _halves(30, 0.5, -0.25)},
        "o_space": {"low": np.array([-2.0, -2.0]), "high": np.array([2.0, 2.0])},
        "a_space": {"low": np.array([-1.5]), "high": np.array([1.5])},
        "x0": np.array([0.0, 0.5]), "model": "first_order_system"},
        steps=29, action_seed=24)

    # biofilm_reactor: the layout of pc-gym_paper/train_policies/Biofilm/biofilm_train.py:45-72, but with x0 / action
    # ranges that keep S2 > -K2 (with the paper's own x0 the reference RHS reaches its Monod pole K2 + S2 = 0 in
    # the second step and returns NaN -- checked with the reference RHS + LSODA)
    S["biofilm_sp"] = dict(env_params={
        "N": 40, "tsim": 40, "SP": {"S2_A": _halves(40, 1.5, 2.0)},
        "o_space": {"low": np.array([-10, 0, -10, 0] * 4 + [0.9], dtype=float),
                    "high": np.array([10, 10, 10, 700] * 4 + [2.1], dtype=float)},
        "a_space": {"low": np.array([5, 10, 0.05, 0.5, 0.05]), "high": np.array([10, 30, 0.2, 1, 1], dtype=float)},
        "x0": np.array([0.3, 1.0, 5, 5] * 4 + [1.5], dtype=float), "model": "biofilm_reactor"},
        steps=39, action_seed=25)
    S["heat_exchanger_sp"] = dict(env_params={
        "N": 30, "tsim": 15.0, "SP": {"Tt8": _halves(30, 330.0, 340.0)},
        "o_space": {"low": np.array([250.0] * 24 + [300.0]), "high": np.array([420.0] * 24 + [360.0])},
        "a_space": {"low": np.array([0.1, 0.1, 350.0, 290.0]), "high": np.array([5.0, 5.0, 400.0, 310.0])},
        "x0": np.array([340.0, 320.0, 300.0] * 8 + [330.0]), "r_scale": {"Tt8": 1e-2}, "model": "heat_exchanger"},
        steps=29, action_seed=26)

    # ---- the remaining registry models (pcgym.py:128-148), one recorded make_env episode each --------------
    S["disease_sp"] = dict(env_params={
        "N": 30, "tsim": 30.0, "SP": {"I": _halves(30, 0.1, 0.05)},
        "o_space": {"low": np.array([0.0, 0.0, 0.0, 0.0]), "high": np.array([1.0, 1.0, 1.0, 0.5])},
        "a_space": {"low": np.array([0.0]), "high": np.array([0.1])},
        "x0": np.array([0.9, 0.1, 0.0, 0.1]), "r_scale": {"I": 10.0}, "model": "disease"},
        steps=29, action_seed=41)
    S["batch_reward"] = dict(env_params={   # terminal reward on the intermediate product
        "N": 20, "tsim": 10.0, "reward_states": ["Cb"], "maximise_reward": True,
        "o_space": {"low": np.array([0.0, 0.0, 0.0, 280.0]), "high": np.array([1.0, 1.0, 1.0, 420.0])},
        "a_space": {"low": np.array([290.0]), "high": np.array([350.0])},
        "x0": np.array([1.0, 0.0, 0.0, 320.0]), "model": "batch"},
        steps=19, action_seed=42)
    S["cstr_series_sp"] = dict(env_params={
        "N": 20, "tsim": 100.0, "SP": {"T2": _halves(20, 320.0, 325.0)},
        "o_space": {"low": np.array([0.0, 280.0, 0.0, 280.0, 300.0]), "high": np.array([100.0, 360.0, 100.0, 360.0, 340.0])},
        "a_space": {"low": np.array([1e-5, 1e-5, 290.0, 290.0]), "high": np.array([5e-5, 5e-5, 310.0, 310.0])},
        "x0": np.array([50.0, 320.0, 40.0, 320.0, 320.0]), "r_scale": {"T2": 1e-2}, "model": "cstr_series_recycle"},
        steps=19, action_seed=43)
    S["polymer_sp"] = dict(env_params={   # short horizon: the reactor runs away thermally within ~10 time units
        "N": 20, "tsim": 4.0, "SP": {"T": _halves(20, 306.0, 308.0)},
        "o_space": {"low": np.array([280.0, 0.0, 0.0, 300.0]), "high": np.array([380.0, 10.0, 1.0, 340.0])},
        "a_space": {"low": np.array([0.005, 290.0, 4.0, 0.3]), "high": np.array([0.01, 310.0, 8.0, 0.6])},
        "x0": np.array([305.0, 5.0, 0.3, 306.0]), "r_scale": {"T": 1e-2}, "model": "polymerisation_reactor"},
        steps=19, action_seed=44)
    S["hydraulic_sp"] = dict(env_params={
        "N": 30, "tsim": 15.0, "SP": {"q2": _halves(30, 1.0, 0.5)},
        "o_space": {"low": np.array([-1.0, -1.0, 0.0]), "high": np.array([3.0, 3.0, 2.0])},
        "a_space": {"low": np.array([-1.0]), "high": np.array([1.0])},
        "x0": np.array([1.0, 1.0, 1.0]), "model": "hydraulic_tank"},
        steps=29, action_seed=45)
    S["nonsmooth_sp"] = dict(env_params={
        "N": 30, "tsim": 15.0, "SP": {"X1": _halves(30, 0.5, -0.25)},
        "o_space": {"low": np.array([-2.0, -2.0, -1.0]), "high": np.array([2.0, 2.0, 1.0])},
        "a_space": {"low": np.array([-1.0]), "high": np.array([1.0])},
        "x0": np.array([0.0, 0.0, 0.5]), "model": "nonsmooth_control"},
        steps=29, action_seed=46)
    # models without inputs: a 1-entry placeholder a_space is what runs through the reference unchanged
    S["invariant_batch_reward"] = dict(env_params={
        "N": 20, "tsim": 2.0, "reward_states": ["xC"], "maximise_reward": True,
        "o_space": {"low": np.zeros(4), "high": np.ones(4)},
        "a_space": {"low": np.array([0.0]), "high": np.array([1.0])},
        "x0": np.array([1.0, 0.8, 0.0, 0.0]), "model": "invariant_batch"},
        steps=19, action_seed=47)
    S["oscillator_sp"] = dict(env_params={
        "N": 20, "tsim": 10.0, "SP": {"x1": _halves(20, 0.0, 0.0)},
        "o_space": {"low": np.array([-3.0] * 20 + [-1.0]), "high": np.array([3.0] * 20 + [1.0])},
        "a_space": {"low": np.array([0.0]), "high": np.array([1.0])},
        "x0": np.array([1.0, 0.5, 0.0, -0.5, -1.0, -0.5, 0.0, 0.5, 1.0, 0.5] + [0.0] * 10 + [0.0]),
        "model": "coupled_oscillator"},
        steps=19, action_seed=48)

    # ---- the custom_reward family of the paper scripts, declarative on our side ------------------------
    # (the reference run uses the callable named in ref_custom_reward, loaded by gen_golden.py from the
    # reference tree; the fixture only holds the recorded tuples)
    p = _cstr_base()
    p.update(custom_reward={"kind": "sp_track", "R": 0.1})
    S["cstr_paper_reward"] = dict(env_params=p, steps=59, action_seed=31,
                                  ref_custom_reward=("pc-gym_paper/train_policies/cstr/custom_reward.py",
                                                     "sp_track_reward"))
    p = _four_tank_base()
    p.update(custom_reward={"kind": "sp_track", "R": 0.1})
    S["four_tank_paper_reward"] = dict(env_params=p, steps=59, action_seed=32,
                                       ref_custom_reward=("pc-gym_paper/train_policies/cstr/custom_reward.py",
                                                          "sp_track_reward"))
    # constraint showcase: box term while a constraint row is violated; the bounds are given in the order the
    # reference's con_reward unpacks them ("lower_bound, upper_bound = [327, 321]", custom_reward.py:4-5,45)
    p = _cstr_base()
    p.update(normalise_a=False, normalise_o=False, constraints=cons_cstr_T, done_on_cons_vio=False, r_penalty=False,
             custom_reward={"kind": "sp_track", "R": 0.01, "box": {"T": [327, 321]}})
    S["cstr_con_reward"] = dict(env_params=p, steps=59, action_seed=33, raw_actions=True,
                                ref_custom_reward=("pc-gym_paper/constraint_showcase/custom_reward.py", "con_reward"))

    # crystallisation: CV and Ln recomputed from the observed moments (cryst_train.py:17-48; the def is compiled out of
    # the training script by gen_golden.py -- the script itself trains on import)
    p = _cryst_base()
    p.update(custom_reward={"kind": "cryst_moments"})
    S["cryst_paper_reward"] = dict(env_params=p, steps=29, action_seed=34,
                                   ref_custom_reward=("pc-gym_paper/train_policies/crystalisation/cryst_train.py",
                                                      "oracle_reward", "extract"))

    # the reference's own known-answer test (custom linear model)
    # ---- non-affine callables: Python on the reference side, C expressions (run-time compiled) on ours ----------
    p = _cstr_base()
    p.update(model="cstr", normalise_a=False, normalise_o=False, done_on_cons_vio=False, r_penalty=True)
    pr = dict(p, constraints=cons_cstr_nonaffine)
    pe = dict(p, constraints={"expr": ["0.02*(T-325.0)*(T-325.0) - 0.5", "0.84 - Ca - 5e-4*pow(Tc-298.0, 2)"]})
    S["cstr_expr_cons_raw"] = dict(env_params=pe, ref_env_params=pr, steps=59, action_seed=41, raw_actions=True)
    p = _cstr_base()
    p.update(model="cstr", normalise_a=True, normalise_o=True, done_on_cons_vio=True, r_penalty=False)
    pr = dict(p, constraints=cons_cstr_nonaffine_q3, custom_reward=reward_cstr_exp)
    pe = dict(p, constraints={"expr": ["1e-6*(x[1]-8500.0)*(x[1]-8500.0) - 0.004", "1.0e3 - 1.0e-2*u[0]*u[0] + x[0]"]},
              custom_reward={"expr": "-1e3*(Ca-SP_Ca)*(Ca-SP_Ca) - 0.05*exp(0.1*(T-330.0)) - (violated ? 5.0 : 0.0)"})
    S["cstr_expr_reward_q3"] = dict(env_params=pe, ref_env_params=pr, steps=59, action_seed=42)

    S["custom_linear_kat"] = dict(
        env_params={
            "custom_model": LinearCustomModel(1.5, 2.5),
            "a_space": {"low": np.array([-1]), "high": np.array([1])},
            "o_space": {"low": np.array([-1, -1]), "high": np.array([1, 1])},
            "SP": {"x2": [2] * 100},
            "N": 100,
            "tsim": 10,
            "x0": np.array([1.0, 1.0]),
        },
        steps=5,
        action_seed=None,
        fixed_action=0.5,
    )
    return S


def actions_for(name, sc):
    """Scripted policy: i.i.d. U(-1,1) (or U(low,high) when normalise_a is off)."""
    p = sc["env_params"]
    na = len(p["a_space"]["low"])
    T = sc["steps"]
    if sc.get("fixed_action") is not None:
        return np.full((T, na), sc["fixed_action"], dtype=np.float64)
    rng = np.random.default_rng(sc["action_seed"])
    a = rng.uniform(-1.0, 1.0, size=(T, na))
    a = a * sc.get("action_scale", 1.0) + sc.get("action_shift", 0.0)
    a = np.clip(a, -1.0, 1.0)
    if sc.get("raw_actions"):
        lo = np.asarray(p["a_space"]["low"], dtype=np.float64)
        hi = np.asarray(p["a_space"]["high"], dtype=np.float64)
        a = (a + 1.0) * (hi - lo) / 2.0 + lo
    return a
