#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

It imports the *reference's own Python* (pc-gym v0.1.8 under /root/reference/src)
behind inert stub modules for the third-party packages that are not installed
here (gymnasium, casadi, diffrax, jax, do_mpc, matplotlib), and records

  rhs_*.npz        reference model RHS  (model_classes.py)  on random (x,u)
  tight_*.npz      (x,u,dt)->x' with the reference RHS + scipy LSODA rtol 1e-13
                   -- "the true solution" the reference's CVODES approximates
  paper_*.npz      rep 0 of the MPC-oracle trajectories the reference ships in
                   pc-gym_paper/**/data.npy  (authored by its own CVODES run)
  step_<scn>.npz   full reference make_env.reset()/step() tuples (pcgym.py:263-500)
                   for the scenarios of scenarios.py, with the CVODES call replaced
                   by LSODA(rtol 1e-12) on the env's own model (zero-order hold on uk,
                   interval [0,dt] -- integrator.py:163-182)

  metrics_ref.npz  outputs of the reference's reproducibility_metric (evaluation_metrics.py:81-327, numpy only) on
                   small random rollout dictionaries, incl. the shapes on which its MAD raises or broadcasts (:127-130)

Only data leaves this script (small .npz under tests/golden/); no reference source
or bytecode is copied.  The script refuses to run when /root/reference is absent.
"""
from __future__ import annotations

import os
import pickletools
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
REF_SRC = os.path.join(REF, "src")


def _install_stubs():
    """Inert stand-ins so the reference modules import.  None of them computes
    anything on the recorded path (numpy does, through the reference's own code)."""
    jax = types.ModuleType("jax")
    jax.numpy = np
    sys.modules["jax"] = jax
    sys.modules["jax.numpy"] = np

    gym = types.ModuleType("gymnasium")

    class Env:  # gym.Env base: the reference only subclasses it
        pass

    class Box:
        def __init__(self, low, high, dtype=np.float64, **kw):
            self.low = np.asarray(low, dtype=np.float64)
            self.high = np.asarray(high, dtype=np.float64)
            self.shape = self.low.shape
            self.dtype = dtype

        def sample(self):
            return np.random.uniform(self.low, self.high)

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box = Box
    gym.Env = Env
    gym.spaces = spaces
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces

    casadi = types.ModuleType("casadi")
    for n in ("SX", "vertcat", "Function", "integrator", "sum1", "reshape", "DM", "mtimes"):
        setattr(casadi, n, None)
    sys.modules["casadi"] = casadi

    diffrax = types.ModuleType("diffrax")
    for n in ("diffeqsolve", "ODETerm", "Tsit5", "PIDController"):
        setattr(diffrax, n, None)
    sys.modules["diffrax"] = diffrax

    sys.modules["do_mpc"] = types.ModuleType("do_mpc")
    mpl = types.ModuleType("matplotlib")
    plt = types.ModuleType("matplotlib.pyplot")
    mpl.pyplot = plt
    sys.modules["matplotlib"] = mpl
    sys.modules["matplotlib.pyplot"] = plt


def _import_reference():
    if not os.path.isdir(REF_SRC):
        raise SystemExit("gen_golden.py: /root/reference is absent -- this script only runs in the build container")
    _install_stubs()
    sys.path.insert(0, REF_SRC)
    import pcgym.pcgym as P  # noqa: E402
    import pcgym.model_classes as M  # noqa: E402
    return P, M


from scipy.integrate import solve_ivp  # noqa: E402


class _TightEngine:
    """Replacement for the reference's integration_engine (integrator.py:7-107):
    same call contract (casadi_step(state, uk) -> {"xf": obj with .full()}), the
    ODE being the env's own reference model, solved to 1e-12."""

    def __init__(self, make_env, env_params):
        self.env = make_env(env_params)

    def casadi_step(self, state, uk):
        nx = self.env.Nx_oracle
        model = self.env.model
        uk = np.asarray(uk, dtype=np.float64).reshape(-1)
        sol = solve_ivp(lambda t, x: np.asarray(model(x, uk), dtype=np.float64),
                        (0.0, self.env.dt), np.asarray(state[:nx], dtype=np.float64),
                        method="LSODA", rtol=1e-12, atol=1e-14)
        xf = sol.y[:, -1]

        class _DM:
            def full(self_inner):
                return xf.reshape(nx, 1)

        return {"xf": _DM()}


def tight_step(model, x, u, dt):
    sol = solve_ivp(lambda t, y: np.asarray(model(y, u), dtype=np.float64), (0.0, dt),
                    np.asarray(x, dtype=np.float64), method="LSODA", rtol=1e-13, atol=1e-15)
    return sol.y[:, -1]


# ---------------------------------------------------------------------------
def gen_rhs_and_tight(M, out, only_new=False):
    rng = np.random.default_rng(20260928)

    def cryst_x(n):
        mu = np.array([1478.00986666666, 22995.8230590611, 1800863.24079725, 248516167.940593])
        X = np.empty((n, 7))
        for i in range(n):
            m = mu * (1 + 0.3 * rng.uniform(-1, 1, 4))
            # keep mu2*mu0/mu1^2 > 1 (CV real)
            m[2] = max(m[2], 1.05 * m[1] ** 2 / m[0])
            c = rng.uniform(0.14, 0.17)
            X[i] = [*m, c, np.sqrt(m[2] * m[0] / m[1] ** 2 - 1), m[1] / m[0]]
        return X

    specs = {
        # name: (ctor, x sampler, u sampler, dt, n_tight)
        "cstr": (lambda: M.cstr(int_method="casadi"),
                 lambda n: np.c_[rng.uniform(0.7, 1.0, n), rng.uniform(310, 350, n)],
                 lambda n: rng.uniform(295, 302, (n, 1)), 26.0 / 60.0),
        "cstr_d": (lambda: M.cstr(int_method="casadi"),
                   lambda n: np.c_[rng.uniform(0.7, 1.0, n), rng.uniform(310, 350, n)],
                   lambda n: np.c_[rng.uniform(295, 302, n), rng.uniform(320, 360, n), rng.uniform(0.9, 1.1, n)],
                   1.0),
        "four_tank": (lambda: M.four_tank(int_method="casadi"),
                      lambda n: rng.uniform(0.05, 0.6, (n, 4)),
                      lambda n: rng.uniform(0.1, 10, (n, 2)), 1000.0 / 60.0),
        "multistage_extraction": (lambda: M.multistage_extraction(int_method="casadi"),
                                  lambda n: rng.uniform(0.05, 0.7, (n, 10)),
                                  lambda n: np.c_[rng.uniform(5, 500, n), rng.uniform(10, 1000, n)], 1.0),
        "multistage_extraction_d": (lambda: M.multistage_extraction(int_method="casadi"),
                                    lambda n: rng.uniform(0.05, 0.7, (n, 10)),
                                    lambda n: np.c_[rng.uniform(5, 60, n), rng.uniform(10, 120, n),
                                                    rng.uniform(0.5, 0.8, n), rng.uniform(0.0, 0.1, n)], 1.0),
        "multistage_extraction_reactive": (lambda: M.multistage_extraction_reactive(int_method="casadi"),
                                           lambda n: rng.uniform(0.0, 2.0, (n, 20)),
                                           lambda n: np.c_[rng.uniform(5, 50, n), rng.uniform(10, 100, n)], 1.0),
        "crystallization": (lambda: M.crystallization(int_method="casadi"),
                            cryst_x, lambda n: rng.uniform(10, 40, (n, 1)), 1.0),
    }
    # ---- "next" row f-2 models (appended: the draws of the entries above are unchanged) ----
    def col(*ranges):
        return lambda n: np.column_stack([rng.uniform(a, b, n) for a, b in ranges])

    def distill_x(n):
        base = np.array([0.95, 0.9, 0.8, 0.65, 0.5, 0.35, 0.2, 0.1, 0.05])
        return np.clip(base * (1 + 0.2 * rng.uniform(-1, 1, (n, 9))), 0.01, 0.99)

    specs.update({
        "complex_cstr": (lambda: M.complex_cstr(int_method="casadi"),
                         col((0.5, 1.0), (0.0, 0.5), (0.0, 0.3), (310, 340)), col((295, 302)), 26.0 / 60.0),
        "complex_cstr_d": (lambda: M.complex_cstr(int_method="casadi"),
                           col((0.5, 1.0), (0.0, 0.5), (0.0, 0.3), (310, 335)),
                           col((295, 302), (330, 355), (0.9, 1.1)), 26.0 / 60.0),
        "disease": (lambda: M.disease_model(int_method="casadi"),
                    col((0.3, 0.9), (0.01, 0.3), (0.0, 0.3)), col((0.0, 0.1)), 1.0),
        "batch": (lambda: M.batch(int_method="casadi"),
                  col((0.3, 1.0), (0.0, 0.5), (0.0, 0.3), (300, 400)), col((290, 350)), 0.5),
        "photobioreactor": (lambda: M.photo_production(int_method="casadi"),
                            col((0.5, 3.0), (50, 800), (0.0, 0.02)), col((120, 400), (0, 40)), 20.0),
        "cstr_series_recycle": (lambda: M.cstr_series_recycle(int_method="casadi"),
                                col((20, 90), (300, 340), (20, 90), (300, 340)),
                                col((1e-5, 5e-5), (1e-5, 5e-5), (290, 310), (290, 310)), 5.0),
        "distillation_column": (lambda: M.distillation_column(int_method="casadi"),
                                distill_x, col((1.0, 5.0), (150, 400)), 1.0),
        "polymerisation_reactor": (lambda: M.polymerisation_reactor(int_method="casadi"),
                                   col((300, 345), (1.0, 8.0), (0.01, 0.5)),
                                   col((0.001, 0.01), (290, 330), (4.0, 8.0), (0.2, 0.6)), 1.0),
        "hydraulic_tank": (lambda: M.hydraulic_tank(int_method="casadi"), col((0, 2), (0, 2)), col((-1, 1)), 0.5),
        "first_order_system": (lambda: M.first_order_system(int_method="casadi"), col((-2, 2)), col((-1, 1)), 0.5),
        "nonsmooth_control": (lambda: M.nonsmooth_control(int_method="casadi"), col((-2, 2), (-2, 2)), col((-1, 1)), 0.5),
    })
    # ---- the last four registry models (appended; earlier draws unchanged) ----
    specs.update({
        "biofilm_reactor": (lambda: M.biofilm_reactor(int_method="casadi"),
                            col(*([(0.5, 5.0), (0.05, 3.0), (1.0, 12.0), (0.05, 20.0)] * 4)),
                            col((0.0, 10.0), (1.0, 30.0), (0.05, 1.0), (0.05, 1.0), (0.05, 1.0)), 1.0),
        "heat_exchanger": (lambda: M.heat_exchanger(int_method="casadi"),
                           col(*([(280.0, 360.0)] * 24)),
                           col((0.1, 5.0), (0.1, 5.0), (300.0, 400.0), (280.0, 320.0)), 1.0),
        # no inputs: one dummy column (ignored by the reference RHS, u=None there)
        "invariant_batch": (lambda: M.invariant_batch(int_method="casadi"),
                            col(*([(0.0, 1.0)] * 4)), col((0.0, 0.0)), 0.1),
        "coupled_oscillator": (lambda: M.coupled_oscillators(int_method="casadi"),
                               col(*([(-1.0, 1.0)] * 20)), col((0.0, 0.0)), 0.5),
    })
    n_rhs, n_tight = 64, 24
    for name, (ctor, xs, us, dt) in specs.items():
        if only_new and os.path.exists(os.path.join(out, f"rhs_{name}.npz")):
            rng_state = xs(n_rhs), us(n_rhs)  # keep the generator stream identical to a full run
            continue
        X = xs(n_rhs)
        U = us(n_rhs)
        DX = np.stack([np.asarray(ctor()(X[i].copy(), U[i].copy()), dtype=np.float64) for i in range(n_rhs)])
        np.savez(os.path.join(out, f"rhs_{name}.npz"), x=X, u=U, dx=DX)
        XT = X[:n_tight]
        UT = U[:n_tight]
        XF = np.stack([tight_step(ctor(), XT[i].copy(), UT[i].copy(), dt) for i in range(n_tight)])
        np.savez(os.path.join(out, f"tight_{name}.npz"), x=XT, u=UT, dt=np.float64(dt), xf=XF)
        print(f"  rhs/tight {name}: {n_rhs}/{n_tight}")


def _safe_load_pickled_npy(path):
    """The paper .npy files are object pickles.  Refuse anything whose pickle
    stream references globals outside numpy."""
    with open(path, "rb") as f:
        magic = np.lib.format.read_magic(f)
        if magic == (1, 0):
            np.lib.format.read_array_header_1_0(f)
        else:
            np.lib.format.read_array_header_2_0(f)
        payload = f.read()
    prev_strs = []
    for op, arg, _ in pickletools.genops(payload):
        if op.name == "GLOBAL":
            mod = arg.split(" ")[0]
            if not (mod.startswith("numpy") or mod == "_codecs"):
                raise RuntimeError(f"{path}: unexpected pickle global {arg!r}")
        elif op.name == "STACK_GLOBAL":
            mod = prev_strs[-2] if len(prev_strs) >= 2 else "?"
            if not (mod.startswith("numpy") or mod == "_codecs"):
                raise RuntimeError(f"{path}: unexpected pickle global {mod!r}")
        if isinstance(arg, str):
            prev_strs.append(arg)
    return np.load(path, allow_pickle=True).item()


def gen_paper(out):
    files = {
        "cstr": ("pc-gym_paper/train_policies/cstr/data.npy", 26.0 / 60.0),
        "four_tank": ("pc-gym_paper/train_policies/four_tank/visualisation/data.npy", 1000.0 / 60.0),
        "multistage_extraction": ("pc-gym_paper/train_policies/multistage_extraction/data.npy", 1.0),
        "crystallization": ("pc-gym_paper/train_policies/crystalisation/data.npy", 1.0),
        "cstr_constraint": ("pc-gym_paper/constraint_showcase/constraint_rollout_data.npy", 26.0 / 60.0),
    }
    for name, (rel, dt) in files.items():
        d = _safe_load_pickled_npy(os.path.join(REF, rel))
        o = d["oracle"]
        x = np.asarray(o["x"], dtype=np.float64)[:, :, 0]  # (nx_obs, N)
        u = np.asarray(o["u"], dtype=np.float64)[:, :, 0]  # (nu, N)
        np.savez(os.path.join(out, f"paper_{name}.npz"), x=x, u=u, dt=np.float64(dt))
        print(f"  paper {name}: x{x.shape} u{u.shape}")


def gen_steps(P, out, only_new=False):
    sys.path.insert(0, HERE)
    import scenarios as SC

    P.integration_engine = _TightEngine
    for name, sc in SC.scenarios().items():
        if only_new and os.path.exists(os.path.join(out, f"step_{name}.npz")):
            continue
        p = sc.get("ref_env_params", sc["env_params"])  # (Python callables where our side takes C expressions)
        if sc.get("ref_custom_reward"):  # the reference side runs its own callable (loaded from the reference tree)
            import copy
            import importlib.util

            rel, fname = sc["ref_custom_reward"][:2]
            p = copy.deepcopy(p)
            if "extract" in sc["ref_custom_reward"][2:]:
                # the function sits in a training script that cannot be imported (it trains on import): compile just
                # that def from the reference file, at generation time only
                import ast

                src = open(os.path.join(REF, rel)).read()
                fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == fname)
                ns = {"np": np}
                exec(compile(ast.Module(body=[fn], type_ignores=[]), rel, "exec"), ns)
                p["custom_reward"] = ns[fname]
            else:
                spec = importlib.util.spec_from_file_location("ref_custom_reward_" + name, os.path.join(REF, rel))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                p["custom_reward"] = getattr(mod, fname)
        A = SC.actions_for(name, sc)
        np.random.seed(12345)  # only matters for action_space.sample() in _setup_constraints
        env = P.make_env(p)
        obs0, info = env.reset()
        T = sc["steps"]
        nobs = obs0.shape[0]
        obs = np.zeros((T + 1, nobs))
        state = np.zeros((T + 1, np.asarray(env.state).shape[0]))
        rew = np.zeros(T)
        done = np.zeros(T, dtype=np.uint8)
        obs[0] = obs0
        state[0] = env.state
        ncon = getattr(env, "n_con", 0) if env.constraint_active else 0
        for i in range(T):
            o, r, d, tr, info = env.step(A[i].copy())
            obs[i + 1] = o
            state[i + 1] = env.state
            rew[i] = r
            done[i] = d
        rec = dict(actions=A, obs=obs, state=state, rew=rew, done=done,
                   dt=np.float64(env.dt), Nx=np.int64(env.Nx), nx=np.int64(env.Nx_oracle))
        if ncon:
            rec["cons_info"] = np.asarray(info["cons_info"][:, :, 0], dtype=np.float64)  # (ncon, N)
        np.savez(os.path.join(out, f"step_{name}.npz"), **rec)
        print(f"  step {name}: T={T} nobs={nobs} ncon={ncon} sum_r={rew.sum():.6g} done_at={int(np.argmax(done)) if done.any() else -1}")


METRIC_CASES = [  # (name, {component: shape}); the last axis is the reps axis
    ("n7_reps5", {"r": (1, 7, 5), "x": (3, 7, 5), "u": (1, 7, 5), "g": (2, 7, 1, 5)}),
    ("n6_reps6", {"r": (1, 6, 6), "x": (2, 6, 6), "u": (2, 6, 6), "g": (3, 6, 1, 6)}),   # N == reps: MAD broadcasts
    ("n1_reps4", {"r": (1, 1, 4), "x": (3, 1, 4), "u": (1, 1, 4)}),                      # N == 1
    ("flat", {"r": (9,), "x": (4, 8)}),                                                  # 1-D (reshaped) and 2-D data
    ("reps1", {"r": (1, 5, 1), "x": (2, 5, 1), "g": (2, 5, 1, 1)}),
]


def gen_metrics(out):
    """The reference's own metric classes (evaluation_metrics.py, imported by path: numpy is its only dependency) on
    seeded random data; a combination on which the reference raises is recorded as such (`<key>_raises`)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_evaluation_metrics",
                                                  os.path.join(REF_SRC, "pcgym", "evaluation_metrics.py"))
    EM = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(EM)
    rng = np.random.default_rng(20260929)
    rec = {}
    for name, shapes in METRIC_CASES:
        data = {"pi": {c: rng.normal(size=sh) * (1 + 3 * rng.random()) for c, sh in shapes.items()}}
        for c, v in data["pi"].items():
            rec[f"{name}__in__{c}"] = v
        for disp in ("std", "mad"):
            for perf in ("mean", "median"):
                m = EM.reproducibility_metric(disp, perf, -1.25)
                for kind, fn in (("perf", m.policy_performance_metric), ("disp", m.policy_dispersion_metric),
                                 ("scal", m.scalarised_performance)):
                    for c in shapes:
                        key = f"{name}__{disp}__{perf}__{kind}__{c}"
                        try:
                            rec[key] = np.asarray(fn(data, c)["pi"][c])
                        except ValueError:
                            rec[key + "_raises"] = np.int64(1)
    np.savez(os.path.join(out, "metrics_ref.npz"), **rec)
    n_raise = sum(1 for k in rec if k.endswith("_raises"))
    print(f"  metrics_ref.npz: {len(rec)} arrays, {n_raise} combinations on which the reference raises")


def main():
    if "--metrics-only" in sys.argv:
        gen_metrics(HERE)
        return
    P, M = _import_reference()
    out = HERE
    only_new = "--only-new" in sys.argv  # keep committed fixtures untouched, add the missing ones
    print("RHS + tight steps"); gen_rhs_and_tight(M, out, only_new)
    if not only_new:
        print("paper trajectories"); gen_paper(out)
    print("full-step tuples"); gen_steps(P, out, only_new)
    print("evaluation metrics"); gen_metrics(out)
    tot = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out) if f.endswith(".npz"))
    print(f"total fixture bytes: {tot}")


if __name__ == "__main__":
    main()
