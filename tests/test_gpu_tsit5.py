"""GPU: Tsit5 (integration_method 'jax' -> the reference's own method, integrator.py:56-61) through the C ABI against the
oracle twin: identical step sequences (quantised controller), states to round-off; full step tuples in both counter modes."""
import copy

import numpy as np
import pytest

import helpers as H
import scenarios as SC

pytestmark = pytest.mark.gpu

CASES = [("cstr", "cstr"), ("cstr_d", "cstr"), ("four_tank", "four_tank"), ("multistage_extraction", "multistage_extraction"),
         ("multistage_extraction_reactive", "multistage_extraction_reactive"), ("crystallization", "crystallization"),
         ("heat_exchanger", "heat_exchanger"), ("biofilm_reactor", "biofilm_reactor"), ("first_order_system", "first_order_system")]


def _torch():
    import torch

    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    return torch


@pytest.mark.parametrize("fix,model", CASES)
def test_integrate_vs_oracle(fix, model):
    torch = _torch()
    from oracle import oracle as O
    from test_gpu_parity import ADAPTIVE_TOL, _plan_for
    from test_oracle_golden import _spec_for_integration

    g = H.gold("tight_" + fix)
    spec = _spec_for_integration(model, float(g["dt"]), g["u"].shape[1], integrator="tsit5")
    lib, plan = _plan_for(spec, torch)
    xs, us = g["x"].T.copy(), g["u"].T.copy()
    x, u = torch.tensor(xs, device="cuda"), torch.tensor(us, device="cuda")
    ns = torch.zeros((2, x.shape[1]), dtype=torch.int32, device="cuda")
    assert lib.pcg_integrate(plan, x.shape[1], x.data_ptr(), u.data_ptr(), ns.data_ptr(), None) == 0
    torch.cuda.synchronize()
    lib.pcg_plan_destroy(plan)
    want, ns_o = O.integrate(spec, xs, us)
    H.adaptive_check(model, x.cpu().numpy(), want, ns.cpu().numpy(), ns_o, fix, tol=ADAPTIVE_TOL.get(fix, 1e-11))
    t = g["xf"].T
    assert np.all(np.abs(x.cpu().numpy() - t) <= 1e-5 * np.abs(t) + 1e-7)


@pytest.mark.parametrize("name", ["cstr_canonical", "me_dist_cons", "cryst_adelta"])
@pytest.mark.parametrize("per_env_t", [False, True])
def test_jax_method_steps_vs_oracle(name, per_env_t):
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()[name]["env_params"])
    p["integration_method"] = "jax"
    B = 600
    env = VecEnv(p, n_envs=B, seed=5, per_env_t=per_env_t)
    assert env.spec.integrator == "tsit5" and env.spec.rtol == 1e-8
    orc = O.OracleEnv(env.spec, B, seed=5, per_env_t=per_env_t)
    env.reset(), orc.reset()
    rng = np.random.default_rng(2)
    for i in range(8):
        a = rng.uniform(-1, 1, (env.spec.na, B))
        if not env.spec.normalise_a:
            a = (a + 1) * (env.spec.a_high - env.spec.a_low)[:, None] / 2 + env.spec.a_low[:, None]
        o, r, d, _, _ = env.step(torch.tensor(a, device=env.device))
        oc, rc, dc = orc.step(a)
        # (8 steps without re-synchronisation: the near-neutral cstr dynamics at 330 K carry round-off along, 1.3e-10 seen)
        H.adaptive_check(env.spec.model.name, env.x.cpu().numpy(), orc.x, env.nsteps.cpu().numpy(), orc.nsteps, (name, i),
                         tol=1e-9)
        assert np.allclose(r.cpu().numpy(), rc, rtol=1e-8, atol=1e-9) and np.array_equal(d.cpu().numpy().astype(np.uint8), dc)
        assert not env.status.any()
    env.close()
