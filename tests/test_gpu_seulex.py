"""GPU: the cooperative rule of PCG_INT_RODAS4 plans (cfg.coop_thr) and SEULEX-8 through the C ABI against the oracle's twin.

The arithmetic is an exactly specified operation sequence on both sides and the rule is exact arithmetic: the kernels and the
oracle pick the SAME envs, take identical big-step sequences and agree to round-off -- whether one lane runs the eight rows
of the tableau (classic kernel, pcg_integrate, fused rollout) or eight lanes share the env (the cooperative phase of the
work-queue kernel, tiles with more heavy envs than a workgroup has groups included).  tests/test_gpu_rodas4.py runs the
default plan (rule on) through every kernel shape as well; here the batches are drawn so that the heavy envs dominate."""
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


def _heavy_actions(rng, na, B, frac):
    """normalised actions; a fraction `frac` of the envs from the corner the pair finds heavy (low liquid, high gas flow)"""
    a = rng.uniform(-1, 1, (na, B))
    k = int(frac * B)
    a[0, :k] = rng.uniform(-1.0, -0.93, k)
    a[1, :k] = rng.uniform(0.2, 1.0, k)
    perm = rng.permutation(B)
    return a[:, perm]


def test_integrate_serial_vs_oracle():
    """pcg_integrate (one lane per env, all eight rows): the (state, action) pairs of the action box incl. its corners"""
    torch = _torch()
    from oracle import oracle as O
    from test_gpu_parity import _plan_for
    from test_rodas4 import _me_box
    from test_seulex import _keys

    spec, cases, refs = _me_box(3000, 5)
    s4 = spec(integrator="rodas4", cooperative={"thr": 48})
    lib, plan = _plan_for(s4, torch)
    for (xx, uu), ref in zip(cases, refs):
        heavy = _keys(s4, xx, uu) >= s4.coop_thr
        assert heavy.sum() >= 100
        x = torch.tensor(xx, device="cuda")
        u = torch.tensor(uu, device="cuda")
        ns = torch.zeros((2, x.shape[1]), dtype=torch.int32, device="cuda")
        assert lib.pcg_integrate(plan, x.shape[1], x.data_ptr(), u.data_ptr(), ns.data_ptr(), None) == 0
        torch.cuda.synchronize()
        want, ns_o = O.integrate(s4, xx, uu)
        H.adaptive_check("multistage_extraction", x.cpu().numpy(), want, ns.cpu().numpy(), ns_o, "integrate", tol=1e-11)
        assert ns_o[:, heavy].sum(0).max() <= 15 and np.max(np.abs(want - ref) / np.abs(ref)) <= 1e-6
    lib.pcg_plan_destroy(plan)


@pytest.mark.parametrize("per_env_t", [False, True])
@pytest.mark.parametrize("frac", [0.08, 0.6])
def test_cooperative_queue_vs_classic_vs_oracle(per_env_t, frac, monkeypatch):
    """the cooperative phase (eight lanes per env) against the classic kernel (one lane, bitwise) and the oracle, 8 steps
    without re-synchronisation; frac = 0.6: far more heavy envs in a tile than its workgroup has groups -- the groups refill"""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    monkeypatch.setenv("PCG_Q_FORCE", "1")
    p = copy.deepcopy(SC.scenarios()["me_dist_cons"]["env_params"])
    p.update(integrator="rodas4", cooperative={"thr": 48})
    B = 2600
    q = VecEnv(p, n_envs=B, seed=4, per_env_t=per_env_t)
    cl = VecEnv(p, n_envs=B, seed=4, per_env_t=per_env_t, variant=1)
    orc = O.OracleEnv(q.spec, B, seed=4, per_env_t=per_env_t)
    q.reset(), cl.reset(), orc.reset()
    rng = np.random.default_rng(8)
    seen_heavy = 0
    for i in range(8):
        a = _heavy_actions(rng, 2, B, frac)
        if not q.spec.normalise_a:
            a = (a + 1) * (q.spec.a_high - q.spec.a_low)[:, None] / 2 + q.spec.a_low[:, None]
        at = torch.tensor(a, device=q.device)
        o, r, d, _, _ = q.step(at)
        cl.step(at)
        oc, rc, dc = orc.step(a)
        assert torch.equal(q.x, cl.x) and torch.equal(q.nsteps, cl.nsteps) and torch.equal(q.rew, cl.rew), i
        H.adaptive_check("multistage_extraction", q.x.cpu().numpy(), orc.x, q.nsteps.cpu().numpy(), orc.nsteps,
                         ("coop", per_env_t, frac, i), tol=1e-11)
        assert np.allclose(r.cpu().numpy(), rc, rtol=1e-9, atol=1e-10)
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), dc) and not q.status.any()
        seen_heavy += int((orc.nsteps.sum(0) <= 15).sum())  # (cheap envs of the pair take > 6 attempts; a loose count)
    assert seen_heavy > 0
    q.close(), cl.close()


def test_rule_on_and_off_are_both_in_class_and_the_rule_shortens_the_chain():
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd.config import EnvSpec

    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p.update(integrator="rodas4")
    B = 1 << 14
    on, off = VecEnv(p, n_envs=B, seed=2), VecEnv(dict(p, cooperative=False), n_envs=B, seed=2)
    assert on.spec.coop_thr == 60.0 and off.spec.coop_thr == 0.0
    on.reset(), off.reset()
    pt = copy.deepcopy(p)
    pt.update(integrator="dopri5", rtol=1e-13, atol=1e-13)
    n_or = 4096
    tru = O.OracleEnv(EnvSpec(pt), n_or, n_threads=8)
    tru.reset()
    gen = torch.Generator(device="cuda").manual_seed(5)
    for i in range(3):
        a = 2 * torch.rand((2, B), generator=gen, device="cuda", dtype=torch.float64) - 1
        tru.x[:] = on.x[:, :n_or].cpu().numpy()
        tru.t = on.t
        off.x.copy_(on.x)
        on.step(a), off.step(a)
        tru.step(a[:, :n_or].cpu().numpy())
        for e in (on, off):
            err = np.max(np.abs(e.x[:, :n_or].cpu().numpy() - tru.x) / np.abs(tru.x))
            assert err <= 1e-6 and not e.status.any(), (i, err)
        att_on, att_off = on.nsteps.sum(dim=0), off.nsteps.sum(dim=0)
        assert att_on.max().item() < att_off.max().item() and att_on.max().item() <= 80, (att_on.max().item(), att_off.max().item())
    on.close(), off.close()


@pytest.mark.parametrize("integrator", ["rodas4", "dopri5"])
@pytest.mark.parametrize("per_env_t", [False, True])
def test_lean_tile_layout_is_the_full_one_bit_for_bit(integrator, per_env_t, monkeypatch):
    """The work-queue kernel's LEAN tile layout (round 5: no first-step and step-count arrays in LDS, only the configured
    disturbance values of the held input -- what lets the tile's state live in LDS where two workgroups share a CU, e.g. the
    extraction segment of BASELINE configs[4]) against the full layout and the classic kernel: every output bitwise equal,
    step counts included (written from phase 2 in the lean layout), with a configured disturbance and a constraint, and
    against the oracle."""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    monkeypatch.setenv("PCG_Q_FORCE", "1")
    p = copy.deepcopy(SC.scenarios()["me_dist_cons"]["env_params"])
    p.update(integrator=integrator)
    if integrator == "rodas4":
        p.update(cooperative={"thr": 48})
    B = 2100
    full = VecEnv(p, n_envs=B, seed=6, per_env_t=per_env_t)
    cl = VecEnv(p, n_envs=B, seed=6, per_env_t=per_env_t, variant=1)
    monkeypatch.setenv("PCG_Q_FORCE_LEAN", "1")
    lean = VecEnv(p, n_envs=B, seed=6, per_env_t=per_env_t)
    orc = O.OracleEnv(full.spec, B, seed=6, per_env_t=per_env_t)
    for e in (full, cl, lean, orc):
        e.reset()
    rng = np.random.default_rng(12)
    for i in range(6):
        a = _heavy_actions(rng, 2, B, 0.2)
        if not full.spec.normalise_a:
            a = (a + 1) * (full.spec.a_high - full.spec.a_low)[:, None] / 2 + full.spec.a_low[:, None]
        at = torch.tensor(a, device=full.device)
        monkeypatch.delenv("PCG_Q_FORCE_LEAN")
        full.step(at), cl.step(at)
        monkeypatch.setenv("PCG_Q_FORCE_LEAN", "1")
        lean.step(at)
        orc.step(a)
        for other in (full, cl):
            assert torch.equal(lean.x, other.x) and torch.equal(lean.nsteps, other.nsteps) and torch.equal(lean.rew, other.rew), i
            assert torch.equal(lean.obs_soa, other.obs_soa) and torch.equal(lean.viol, other.viol) and torch.equal(lean.done, other.done), i
        tol = 1e-11 if integrator == "rodas4" else 1e-10
        H.adaptive_check("multistage_extraction", lean.x.cpu().numpy(), orc.x, lean.nsteps.cpu().numpy(), orc.nsteps,
                         ("lean", integrator, per_env_t, i), tol=tol)
    for e in (full, cl, lean):
        e.close()
