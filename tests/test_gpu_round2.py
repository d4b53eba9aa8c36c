"""GPU: round-2 parity cases -- BASELINE configs[4] (mixed batch with Gaussian disturbances) against the oracle, the
per-env status byte, the default cstr path on the ignition branch, checkpoint round trips, the feature-masked
pipelined kernels against the classic one-env-per-lane kernel."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H  # noqa: F401
import scenarios as SC

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torch():
    import torch

    assert torch.cuda.is_available()
    return torch


def _cmp(env, orc, tol_x, tag):
    """state / observation / reward / done / status of one step against the oracle; adaptive plans through
    helpers.adaptive_check (identical step sequences)"""
    name = env.spec.model.name
    if env.nsteps is not None:
        H.adaptive_check(name, env.x.cpu().numpy(), orc.x, env.nsteps.cpu().numpy(), orc.nsteps, tag, tol=tol_x)
    xs = np.maximum(np.abs(orc.x), 1e-6 * np.max(np.abs(orc.x), axis=1, keepdims=True))
    ex = np.max(np.abs(env.x.cpu().numpy() - orc.x) / xs)
    eo = np.max(np.abs(env.obs_soa.cpu().numpy() - orc.obs) / np.maximum(np.abs(orc.obs), 1e-3))
    er = np.max(np.abs(env.rew.cpu().numpy() - orc.rew) / np.maximum(np.abs(orc.rew), 1.0))
    assert ex <= tol_x and eo <= tol_x * 10 and er <= max(tol_x * 1e3, 1e-9), (tag, ex, eo, er)
    assert np.array_equal(env.done.cpu().numpy(), orc.done), tag
    assert np.array_equal(env.status.cpu().numpy(), orc.status), tag


def test_mixed_batch_with_gaussian_disturbances_vs_oracle():
    """BASELINE configs[4] at test size: the bench's own segment configurations (bench.mixed_segments: cstr +
    Ti ~ N(350,2), four_tank, multistage_extraction + X0 ~ N(0.6,0.02), set-point step changes), three plans on three
    streams, stepped over an episode boundary with same-launch auto-reset; one OracleEnv per segment with the same
    GLOBAL env offsets draws the same Philox streams.  Actions over the full box (ME: |lambda| dt up to ~240)."""
    torch = _torch()
    import bench as BN
    from oracle import oracle as O
    from pcgym_amd import MixedVecEnv

    segs = BN.mixed_segments(3 * 1400)
    for p, _ in segs:  # short episodes so that the run crosses two resets
        N = 7
        p["tsim"] = N * float(p["tsim"]) / p["N"]
        p["N"] = N
        p["SP"] = {k: list(np.asarray(v, dtype=float)[::10][:N]) for k, v in p["SP"].items()}
        if p.get("disturbances"):
            p["disturbances"] = {k: np.asarray(v)[:N] for k, v in p["disturbances"].items()}
    mixed = MixedVecEnv(segs, seed=21, env_offset=5 * 10**9, auto_reset=True)
    orcs = [O.OracleEnv(e.spec, e.B, seed=21, env_offset=off) for e, off in zip(mixed.envs, mixed.offsets)]
    assert [e.spec.model.name for e in mixed.envs] == ["cstr", "four_tank", "multistage_extraction"]
    assert mixed.envs[0].spec.gauss and mixed.envs[2].spec.gauss and mixed.envs[2].spec.integrator == "rodas5"
    mixed.reset()
    for o in orcs:
        o.reset()
    rng = np.random.default_rng(4)
    for i in range(15):
        acts = [rng.uniform(-1, 1, (e.spec.na, e.B)) for e in mixed.envs]
        acts[1] = 0.75 * acts[1] + 0.25  # four_tank: keep tank 3 from draining (sqrt of a negative level, both sides)
        mixed.step([torch.tensor(a, device=mixed.device) for a in acts])
        torch.cuda.synchronize()
        for e, o, a in zip(mixed.envs, orcs, acts):
            o.step(a)
            rew, done = o.rew.copy(), o.done.copy()
            if o.t == e.N - 1:
                o.reset()  # what the fused launch did: next episode, next RNG key
            name = e.spec.model.name
            tx = 1e-11  # the ME segment included: identical step sequences, bit-exact right-hand side (helpers.py)
            xs = np.maximum(np.abs(o.x), 1e-6 * np.max(np.abs(o.x), axis=1, keepdims=True))
            ex = np.max(np.abs(e.x.cpu().numpy() - o.x) / xs, axis=0)
            eo = np.max(np.abs(e.obs_soa.cpu().numpy() - o.obs) / np.maximum(np.abs(o.obs), 1e-3))
            assert ex.max() <= tx and eo <= tx * 10, (name, i, ex.max(), eo)
            assert np.allclose(e.rew.cpu().numpy(), rew, rtol=1e-9, atol=1e-10), (name, i)
            assert np.array_equal(e.done.cpu().numpy(), done), (name, i)
            assert e.t == o.t and not e.status.any()
            if e.nsteps is not None and o.t != 0:  # (after a fused reset the counts of the finished step are kept on both sides)
                same = np.all(e.nsteps.cpu().numpy() == o.nsteps, axis=0)
                assert same.all(), (name, i, same.mean())
    # the Gaussian disturbance really is per env and inside its clip box (observation slot, un-normalised)
    me = mixed.envs[2]
    lo, hi = me.spec.o_low[-1], me.spec.o_high[-1]
    x0_obs = (me.obs_soa[-1].cpu().numpy() + 1) / 2 * (hi - lo) + lo
    assert 0.015 < x0_obs.std() < 0.025 and abs(x0_obs.mean() - 0.6) < 0.003 and x0_obs.min() >= 0.5
    mixed.close()


def test_mixed_full_shard_properties():
    """BASELINE configs[4] at full shard size (1,048,572 envs): size-independent properties -- lane independence of
    every segment under a permutation of its envs' actions/state (bitwise), an oracle-checked window at the END of
    every segment (global offsets), finiteness and the status byte."""
    torch = _torch()
    import bench as BN
    from oracle import oracle as O
    from pcgym_amd import MixedVecEnv

    segs = BN.mixed_segments(1 << 20)
    m1 = MixedVecEnv(segs, seed=3)
    m2 = MixedVecEnv(segs, seed=3)
    assert m1.B == 3 * 349524
    m1.reset()
    m2.reset()
    W = 1024
    gen = torch.Generator(device=m1.device).manual_seed(5)
    orcs = [O.OracleEnv(e.spec, W, seed=3, env_offset=off + e.B - W) for e, off in zip(m1.envs, m1.offsets)]
    for o in orcs:
        o.reset()
    for i in range(2):
        acts = []
        for e in m1.envs:
            a = 2 * torch.rand((e.spec.na, e.B), generator=gen, device=e.device, dtype=torch.float64) - 1
            acts.append(0.75 * a + 0.25 if e.spec.model.name == "four_tank" else a)
        m1.step(acts)
        m2.step(acts)
        torch.cuda.synchronize()
        for e, e2, o, a in zip(m1.envs, m2.envs, orcs, acts):
            assert torch.equal(e.x, e2.x) and torch.equal(e.obs_soa, e2.obs_soa)  # run-to-run determinism
            o.step(a[:, e.B - W:].cpu().numpy())
            xs = np.maximum(np.abs(o.x), 1e-6 * np.max(np.abs(o.x), axis=1, keepdims=True))
            assert np.max(np.abs(e.x[:, e.B - W:].cpu().numpy() - o.x) / xs) <= 1e-11, (e.spec.model.name, i)
            assert np.max(np.abs(e.obs_soa[:, e.B - W:].cpu().numpy() - o.obs)) <= 1e-10
            if e.nsteps is not None:  # the strict slice check of configs[2], inside the mixed shard: identical step
                # sequences for every env of the window (bit-exact right-hand side for the extraction segment)
                H.adaptive_check(e.spec.model.name, e.x[:, e.B - W:].cpu().numpy(), o.x,
                                 e.nsteps[:, e.B - W:].cpu().numpy(), o.nsteps, ("mixed full shard", i), tol=1e-11)
            assert bool(torch.isfinite(e.x).all()) and not bool(e.status.any())
    m1.close()
    m2.close()


def test_me_gaussian_inlet_disturbance_vs_oracle():
    """the ME segment of configs[4] on its own: X0 ~ N(0.6, 0.02) per env and step, DOPRI5, full action box, per-env
    counters; B odd -> classic kernel, B even -> same kernel (adaptive plans have no two-env form)."""
    torch = _torch()
    import bench as BN
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    p = BN.mixed_segments(6)[2][0]
    for B, pet in ((1001, False), (2048, True)):
        env = VecEnv(p, n_envs=B, seed=9, per_env_t=pet, env_offset=123456789)
        orc = O.OracleEnv(env.spec, B, seed=9, per_env_t=pet, env_offset=123456789)
        env.reset()
        orc.reset()
        rng = np.random.default_rng(B)
        for i in range(6):
            a = rng.uniform(-1, 1, (2, B))
            env.step(torch.tensor(a, device=env.device))
            orc.step(a)
            torch.cuda.synchronize()
            _cmp(env, orc, 1e-11, (B, i))
            env.x.copy_(torch.tensor(orc.x, device=env.device))  # one-step comparisons (stability-limited model)
        env.close()


def test_status_reports_failed_and_nonfinite_steps():
    """PCG_ST_*: a DOPRI5 lane that exhausts its step budget is flagged and its state poisoned (never a partially
    integrated state); a non-finite state is flagged; healthy envs are untouched.  Oracle agrees env by env."""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd import _abi as abi

    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p.update(integrator="dopri5", max_steps=70)
    B = 1500
    env = VecEnv(p, n_envs=B, seed=1)
    orc = O.OracleEnv(env.spec, B, seed=1)
    env.reset()
    orc.reset()
    a = np.random.default_rng(0).uniform(-1, 1, (2, B))
    _, _, _, _, info = env.step(torch.tensor(a, device=env.device))
    orc.step(a)
    st = info["status"].cpu().numpy()
    assert np.mean(st == orc.status) >= 0.97  # envs right at the budget may differ by a step (stability-limited model)
    bad = st == abi.PCG_ST_MAX_STEPS
    assert 0.1 < bad.mean() < 0.9                      # the stiff (high-flow) envs run out of budget, the mild ones do not
    x = env.x.cpu().numpy()
    assert np.isnan(x[:, bad]).all() and np.isfinite(x[:, ~bad]).all()
    assert np.isnan(env.rew.cpu().numpy()[bad]).all()
    both = ~bad & (orc.status == 0)
    assert np.allclose(x[:, both], orc.x[:, both], rtol=2e-6)
    env.close()
    # fixed-step kernels (lean / feature-masked and classic): a NaN state in -> PCG_ST_NONFINITE out
    for B in (1024, 1023):
        p = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
        p.update(integrator="rk4")
        env = VecEnv(p, n_envs=B, seed=1)
        env.reset()
        env.x[1, 5] = float("nan")
        env.x[0, B - 1] = float("inf")
        env.step(torch.zeros((1, B), dtype=torch.float64, device=env.device))
        st = env.status.cpu().numpy()
        assert st[5] == abi.PCG_ST_NONFINITE and st[B - 1] == abi.PCG_ST_NONFINITE and st.sum() == 2 * abi.PCG_ST_NONFINITE
        env.close()


def test_default_cstr_integrator_survives_the_ignition_branch():
    """SURVEY.md section 8(d) config 2's own x0 box U(0.7,1.0) x U(310,350) K at B = 2^20 with the DEFAULT integrator
    (adaptive: config.py): a quarter of these envs ignite (T -> 440..480 K), where fixed-step RK4 returned finite garbage
    in round 1.  Everything stays finite and physical, status is clean, a 4096-env slice agrees with a 1e-13 adaptive
    solve and a 256-env slice with SciPy LSODA(1e-13) on the oracle's RHS to the reference's accuracy class."""
    torch = _torch()
    import bench as BN
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd.config import EnvSpec
    from scipy.integrate import solve_ivp

    p = BN.workload_params()
    del p["integrator"], p["substeps"]
    p.update(tsim=26.0, x0=np.array([0.85, 330.0, 0.85]), uncertainty_percentages={"x0": [0.15 / 0.85, 20.0 / 330.0]})
    B, W = 1 << 20, 4096
    env = VecEnv(p, n_envs=B, seed=77)
    assert env.spec.integrator == "tsit5g"  # the guarded plan: igniting envs fall back to the adaptive pair at 1e-10
    env.reset()
    x0 = env.x.clone()
    assert float(x0[1].max()) > 349.0 and float(x0[1].min()) < 311.0
    gen = torch.Generator(device=env.device).manual_seed(1)
    a = 2 * torch.rand((1, B), generator=gen, device=env.device, dtype=torch.float64) - 1
    env.step(a)
    x1 = env.x.cpu().numpy()
    assert np.isfinite(x1).all() and not env.status.any()
    assert (x1[0] >= 0).all() and (x1[0] <= 1.0 + 1e-9).all() and (x1[1] > 300).all() and (x1[1] < 600).all()
    assert (x1[1] > 400).mean() > 0.05  # the ignition branch is really in the batch
    # tight adaptive solve of the same step on a slice
    pt = dict(p)
    pt.update(integrator="dopri5", rtol=1e-13, atol=1e-13)
    orc = O.OracleEnv(EnvSpec(pt), W, seed=77)
    orc.reset()
    assert np.allclose(orc.x, x0[:, :W].cpu().numpy(), rtol=1e-15)
    orc.step(a[:, :W].cpu().numpy())
    rel = np.abs(x1[:, :W] - orc.x) / np.abs(orc.x)
    assert rel.max() <= 1e-6, rel.max()
    # LSODA on the oracle's RHS (the generator of the golden fixtures used the same solver on the reference's RHS)
    par = np.array(env.spec.param_vector())
    Tc = (a[0, :256].cpu().numpy() + 1) * 3.5 + 295.0
    worst = 0.0
    for k in range(256):
        u = np.array([[Tc[k]], [350.0], [1.0]])
        f = lambda t, y: O.rhs(0, par, y.reshape(2, 1), u)[:, 0]  # noqa: E731
        sol = solve_ivp(f, (0.0, env.dt), x0[:, k].cpu().numpy(), method="LSODA", rtol=1e-13, atol=1e-13)
        worst = max(worst, float(np.max(np.abs(x1[:, k] - sol.y[:, -1]) / np.abs(sol.y[:, -1]))))
    assert worst <= 1e-6, worst
    env.close()


@pytest.mark.parametrize("case", ["unc_adelta", "track_per_t"])
def test_state_dict_round_trip(case):
    """step k, save, load into a NEW env, step: equal to the uninterrupted run -- including the per-env uncertain
    parameters sampled at reset, accumulators and the last outputs (env.obs is the policy's next input)."""
    torch = _torch()
    from pcgym_amd import VecEnv

    if case == "unc_adelta":
        p = copy.deepcopy(SC.scenarios()["cryst_adelta"]["env_params"])
        p.update(uncertainty_percentages={"kg": 0.05, "x0": [0.01] * 7}, distribution="uniform",
                 uncertainty_bounds={"low": np.array([40.0]), "high": np.array([56.0])})
        kw = {}
    else:
        p = copy.deepcopy(SC.scenarios()["cstr_paper_reward"]["env_params"])
        p.update(integrator="rk4", noise=True, noise_percentage=0.01)
        kw = dict(per_env_t=True, auto_reset=True)
    B = 514
    a = torch.tensor(np.random.default_rng(2).uniform(-1, 1, (10, len(p["a_space"]["low"]), B)), device="cuda")
    e1 = VecEnv(copy.deepcopy(p), n_envs=B, seed=4, **kw)
    e1.reset()
    for i in range(4):
        e1.step(a[i])
    sd = e1.state_dict()
    e2 = VecEnv(copy.deepcopy(p), n_envs=B, seed=999, **kw)  # fresh env, different seed: everything comes from the dict
    e2.load_state_dict(sd)
    assert torch.equal(e1.obs, e2.obs)
    for i in range(4, 10):
        o1, r1, d1, _, _ = e1.step(a[i])
        o2, r2, d2, _, _ = e2.step(a[i])
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(e1.x, e2.x), i
    if e1.p_unc is not None:
        assert torch.equal(e1.p_unc, e2.p_unc) and float(e1.p_unc.std()) > 0
    e1.close()
    e2.close()


FEAT_CASES = [
    ("cstr_canonical", dict(noise=True, noise_percentage=0.01), {}),
    ("cstr_dist_Ti", dict(gaussian_disturbances={"Ti": 2.0}), {}),
    ("cstr_cons_pen_norm", {}, {}),
    ("cstr_cons_done_raw", {}, dict(per_env_t=True, auto_reset=True)),
    ("cstr_paper_reward", dict(noise=True, noise_percentage=0.01), {}),
    ("cstr_con_reward", dict(noise=True, noise_percentage=0.005), {}),
    ("cstr_batch_reward", {}, {}),
    ("cstr_canonical", {}, dict(per_env_t=True)),
    ("cstr_canonical", dict(uncertainty_percentages={"x0": [0.05, 0.003]}, distribution="normal"), dict(auto_reset=True)),
    ("four_tank_canonical", dict(noise=True, noise_percentage=0.02), dict(per_env_t=True, auto_reset=True)),
    ("four_tank_paper_reward", {}, {}),
]


@pytest.mark.parametrize("name,extra,kw", FEAT_CASES)
def test_feature_masked_kernels_equal_the_classic_kernel(name, extra, kw):
    """every curated feature mask (pcg_step_feat.hpp: two envs per lane, compile-time features) against the classic
    one-env-per-lane kernel with run-time flags (PCG_OPT_VARIANT 1) AND against the oracle, over an episode boundary"""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()[name]["env_params"])
    p.update(extra)
    if p.get("model") == "cstr":
        p["integrator"] = "rk4"
    N = 8
    p["tsim"] = N * float(p["tsim"]) / p["N"]
    p["N"] = N
    if p.get("SP"):
        p["SP"] = {k: list(np.asarray(v, dtype=float)[::8][:N]) for k, v in p["SP"].items()}
    if p.get("disturbances"):
        p["disturbances"] = {k: np.asarray(v)[:N] for k, v in p["disturbances"].items()}
    B = 2050
    fe = VecEnv(copy.deepcopy(p), n_envs=B, seed=6, env_offset=10**10, **kw)
    cl = VecEnv(copy.deepcopy(p), n_envs=B, seed=6, env_offset=10**10, variant=1, **kw)
    orc = O.OracleEnv(fe.spec, B, seed=6, env_offset=10**10, per_env_t=kw.get("per_env_t", False))
    for e in (fe, cl, orc):
        e.reset()
    rng = np.random.default_rng(8)
    auto, pet = bool(kw.get("auto_reset")), bool(kw.get("per_env_t"))
    for i in range(2 * N if auto else N - 1):  # without auto-reset: one episode; with it: across two boundaries
        a = rng.uniform(-1, 1, (fe.spec.na, B))
        if not fe.spec.normalise_a:
            a = (a + 1) * (fe.spec.a_high - fe.spec.a_low)[:, None] / 2 + fe.spec.a_low[:, None]
        at = torch.tensor(a, device=fe.device)
        o1, r1, d1, _, _ = fe.step(at)
        o2, r2, d2, _, _ = cl.step(at)
        assert torch.allclose(o1, o2, rtol=1e-13, atol=1e-13) and torch.allclose(r1, r2, rtol=1e-12, atol=1e-12), (name, i)
        assert torch.equal(d1, d2) and torch.allclose(fe.x, cl.x, rtol=1e-13, atol=0), (name, i)
        assert torch.equal(fe.viol, cl.viol) and torch.equal(fe.status, cl.status)
        if fe.g is not None:
            assert torch.allclose(fe.g, cl.g, rtol=1e-12, atol=1e-12)
        if fe.t_env is not None:
            assert torch.equal(fe.t_env, cl.t_env)
        if fe.u_prev is not None:
            assert torch.equal(fe.u_prev, cl.u_prev)
        if fe.a_save_t is not None:
            assert torch.allclose(fe.a_save_t, cl.a_save_t, rtol=1e-14)
        # oracle: the step, then what the fused launch did (masked reset of the finished envs / of the whole batch)
        orc.step(a)
        rc, dc = orc.rew.copy(), orc.done.copy()
        if auto and pet:
            orc.reset(mask=dc)
        elif auto and orc.t == N - 1:
            orc.reset()
        assert np.array_equal(d1.cpu().numpy().astype(np.uint8), dc), (name, i)
        assert np.allclose(r1.cpu().numpy(), rc, rtol=1e-10, atol=1e-11), (name, i)
        assert np.allclose(fe.x.cpu().numpy(), orc.x, rtol=1e-12, atol=0), (name, i)
        assert np.allclose(fe.obs_soa.cpu().numpy(), orc.obs, rtol=1e-10, atol=1e-11), (name, i)
    fe.close()
    cl.close()


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["PCG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PCG_ROOT"], "tests", "golden"))
import time
import numpy as np, torch, torch.distributed as dist
import bench as BN
from pcgym_amd import MixedVecEnv, make_mixed_sharded_env, shard
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)                       # both ranks share this box's GPU: the sharding logic is what is tested
sizes = [2002, 1501, 1000]
segs = [(p, n) for (p, _), n in zip(BN.mixed_segments(6), sizes)]
env = make_mixed_sharded_env(segs, rank=rank, world=world, device=0, seed=31)
env.reset()
T = 5
rng = np.random.default_rng(12)
acts = [[rng.uniform(-1, 1, (e.spec.na, n)) for e, n in zip(env.envs, sizes)] for _ in range(T)]
for a in acts:
    a[1][:] = 0.75 * a[1] + 0.25
los = [shard.shard_range(n, rank, world) for n in sizes]
for i in range(T):
    env.step([torch.tensor(a[:, lo:hi], device="cuda") for a, (lo, hi) in zip(acts[i], los)])
torch.cuda.synchronize()
t0 = time.perf_counter()
full = [(shard.gather_to_rank0(e.x.cpu(), n, dim=1), shard.gather_to_rank0(e.obs_soa.cpu(), n, dim=1),
         shard.gather_to_rank0(e.rew.cpu(), n, dim=0)) for e, n in zip(env.envs, sizes)]
gather_s = time.perf_counter() - t0
if rank == 0:
    ref = MixedVecEnv(segs, device=0, seed=31)    # the un-sharded run
    ref.reset()
    for i in range(T):
        ref.step([torch.tensor(a, device="cuda") for a in acts[i]])
    torch.cuda.synchronize()
    for (x, o, r), e in zip(full, ref.envs):
        assert torch.equal(x, e.x.cpu()), "sharded state != single-device state (" + e.spec.model.name + ")"
        assert torch.equal(o, e.obs_soa.cpu()) and torch.equal(r, e.rew.cpu())
    print("MIXED_SHARD_OK gather_s=%.4f" % gather_s)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_step_a_sharded_mixed_batch_equal_to_the_unsharded_run(tmp_path):
    """make_mixed_sharded_env stepped by two gloo ranks (sharing this box's GPU), gathered to rank 0 == MixedVecEnv of
    the whole batch: same global env offsets -> same Gaussian-disturbance streams, bit for bit."""
    _torch()
    import socket

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PCG_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MIXED_SHARD_OK" in r.stdout


@pytest.mark.parametrize("name,B,kw", [("me_reactive", 3001, {}), ("cstr_canonical", 70000, {}), ("me_canonical", 2500, {}),
                                       ("cstr_cons_done_raw", 1500, dict(per_env_t=True, auto_reset=True)),
                                       ("heat_exchanger_sp", 700, {}), ("cryst_adelta", 900, {})])
def test_work_queue_kernel_equals_the_classic_adaptive_kernel(name, B, kw, monkeypatch):
    """DOPRI5 plans run on the LDS work-queue kernel (pcg_step_queue.hpp: lanes pull the next env of a cost-sorted tile
    when theirs is done); PCG_OPT_VARIANT 1 keeps the classic one-env-per-lane kernel.  Same per-env arithmetic ->
    identical step counts and round-off-level states for the accuracy-limited models; ragged batch sizes exercise
    partial tiles, uneven workgroup ranges and more than one sub-tile per workgroup."""
    torch = _torch()
    from pcgym_amd import VecEnv

    monkeypatch.setenv("PCG_Q_FORCE", "1")  # thinly filled tiles too (the host would route them to the classic kernel)
    p = copy.deepcopy(SC.scenarios()[name]["env_params"])
    p["integrator"] = "dopri5"
    q = VecEnv(copy.deepcopy(p), n_envs=B, seed=3, **kw)
    cl = VecEnv(copy.deepcopy(p), n_envs=B, seed=3, variant=1, **kw)
    q.reset()
    cl.reset()
    rng = np.random.default_rng(1)
    for i in range(5):
        a = rng.uniform(-1, 1, (q.spec.na, B))
        if not q.spec.normalise_a:
            a = (a + 1) * (q.spec.a_high - q.spec.a_low)[:, None] / 2 + q.spec.a_low[:, None]
        at = torch.tensor(a, device=q.device)
        o1, r1, d1, _, _ = q.step(at)
        o2, r2, d2, _, _ = cl.step(at)
        H.adaptive_check(q.spec.model.name, q.x.cpu().numpy(), cl.x.cpu().numpy(), q.nsteps.cpu().numpy(),
                         cl.nsteps.cpu().numpy(), (name, i), tol=1e-12)
        assert torch.equal(d1, d2) and torch.equal(q.status, cl.status), (name, i)
        tol = 1e-11
        assert torch.allclose(o1, o2, rtol=tol, atol=tol) and torch.allclose(r1, r2, rtol=tol * 100, atol=tol), (name, i)
        if q.t_env is not None:
            assert torch.equal(q.t_env, cl.t_env)
    q.close()
    cl.close()


def test_work_queue_results_do_not_depend_on_the_batch_order(monkeypatch):
    """bitwise lane independence: the same envs in a permuted batch land in other tiles, other sort positions and other
    lanes of the work-queue kernel -- and give the same bits (the chaotic extraction model included)."""
    torch = _torch()
    from pcgym_amd import VecEnv

    monkeypatch.setenv("PCG_Q_FORCE", "1")
    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    B = 40000
    e1, e2 = VecEnv(p, n_envs=B, seed=1), VecEnv(p, n_envs=B, seed=1)
    e1.reset()
    e2.reset()
    gen = torch.Generator(device="cuda").manual_seed(2)
    x0 = e1.x * (1 + 0.05 * (2 * torch.rand(e1.x.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
    perm = torch.randperm(B, generator=gen, device="cuda")
    e1.x.copy_(x0)
    e2.x.copy_(x0[:, perm])
    for i in range(3):
        a = 2 * torch.rand((2, B), generator=gen, device="cuda", dtype=torch.float64) - 1
        e1.step(a)
        e2.step(a[:, perm])
        assert torch.equal(e1.x[:, perm], e2.x) and torch.equal(e1.rew[perm], e2.rew), i
        assert torch.equal(e1.nsteps[:, perm], e2.nsteps) and torch.equal(e1.obs_soa[:, perm], e2.obs_soa)
    e1.close()
    e2.close()


# ---------------------------------------------------------------- run-time compiled user expressions (hipRTC) ----
@pytest.mark.parametrize("name", ["cstr_expr_cons_raw", "cstr_expr_reward_q3"])
def test_user_expressions_in_the_batched_kernel_match_the_reference_callables(name):
    """non-affine constraints(x,u) / custom_reward(self, obs, uk, violated): Python callables in the reference
    (tests/golden/scenarios.py: cons_cstr_nonaffine*, reward_cstr_exp), C expressions compiled into the general step
    kernel with hipRTC here.  (a) every env of a VecEnv batch replays the reference's recording; (b) per-env random
    actions against the oracle's integration + a NumPy evaluation of the same callables."""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd.config import EnvSpec

    sc = SC.scenarios()[name]
    g = H.gold("step_" + name)
    p = copy.deepcopy(sc["env_params"])
    p.update(H.tight_for(p))
    B = 130
    env = VecEnv(p, n_envs=B, seed=1)
    assert env.spec.user_cons_src is not None and env.spec.ncon == 2
    A = SC.actions_for(name, sc)
    obs, _ = env.reset()
    assert np.allclose(obs.cpu().numpy(), g["obs"][0][None, :], rtol=1e-12, atol=1e-12)
    ci = g["cons_info"]
    for i in range(sc["steps"]):
        a = torch.tensor(np.repeat(A[i].reshape(-1, 1), B, axis=1), device=env.device)
        o, r, d, _, info = env.step(a)
        want = g["obs"][i + 1]
        assert np.all(np.abs(o.cpu().numpy() - want[None, :]) <= 2e-9 * np.maximum(np.abs(want), 1.0)), (name, i)
        assert np.allclose(r.cpu().numpy(), g["rew"][i], rtol=1e-7, atol=1e-9), (name, i, r[0].item(), g["rew"][i])
        if i == 0:
            assert np.allclose(env.g_pre.cpu().numpy(), ci[:, 0:1], rtol=1e-9, atol=1e-9 * np.max(np.abs(ci)))
        assert np.allclose(info["g"].cpu().numpy(), ci[:, i + 1:i + 2], rtol=1e-8, atol=1e-9 * np.max(np.abs(ci))), (name, i)
        assert np.array_equal(info["viol"].cpu().numpy(), np.full(B, int((ci[:, i + 1] > 0).any()), dtype=np.uint8))
    env.close()

    # (b) random per-env actions: integration by the oracle (same plan without the expressions), epilogue in NumPy
    ref = sc["ref_env_params"]
    cons_py, rew_py = ref["constraints"], ref.get("custom_reward")
    pb = copy.deepcopy(p)
    pb.pop("constraints"), pb.pop("custom_reward", None)
    for k in ("done_on_cons_vio", "r_penalty"):
        pb.pop(k, None)
    B = 1001
    env = VecEnv(copy.deepcopy(p), n_envs=B, seed=3, per_env_t=True)
    s = env.spec
    orc = O.OracleEnv(EnvSpec(pb), B, seed=3, per_env_t=True)
    env.reset()
    orc.reset()
    rng = np.random.default_rng(5)
    lo, hi = s.a_low[0], s.a_high[0]
    olo, ohi = s.o_low, s.o_high
    for i in range(6):
        a = rng.uniform(-1, 1, (1, B)) if s.normalise_a else rng.uniform(lo, hi, (1, B))
        _, r, d, _, info = env.step(torch.tensor(a, device=env.device))
        orc.step(a)
        uk = (a[0] + 1) * (hi - lo) / 2 + lo if s.normalise_a else a[0]
        sp_old = s.sp[0, min(i, s.N - 1)]
        sp_new = s.sp[0, min(i + 1, s.N - 1)]
        gg = np.zeros((2, B))
        rr = np.zeros(B)
        for b in range(B):
            st = np.array([orc.x[0, b], orc.x[1, b], sp_old])
            uu = np.array([uk[b]])
            sq, uq = st, uu
            if s.reference_compat and s.normalise_o:  # quirk Q3: what the reference hands the callable
                sq = (st + 1) * (ohi - olo) / 2 + olo
            if s.reference_compat and s.normalise_a:
                uq = (uu + 1) * (hi - lo) / 2 + lo
            gg[:, b] = cons_py(sq, uq)
            viol = bool((gg[:, b] > 0).any())
            if rew_py is not None:
                class _E:  # the attributes the callable reads from `self`
                    SP = {"Ca": s.sp[0]}
                    t = i + 1
                rr[b] = rew_py(_E, st, uu, viol)
            else:
                rr[b] = -1e3 * (st[0] - sp_new) ** 2 - (1000.0 if (viol and s.r_penalty) else 0.0)
        assert np.allclose(env.x.cpu().numpy(), orc.x, rtol=1e-11), (name, i)
        assert np.allclose(info["g"].cpu().numpy(), gg, rtol=1e-9, atol=1e-9 * np.abs(gg).max()), (name, i)
        assert np.array_equal(info["viol"].cpu().numpy().astype(bool), (gg > 0).any(axis=0)), (name, i)
        assert np.allclose(r.cpu().numpy(), rr, rtol=1e-9, atol=1e-9), (name, i)
    env.close()


def test_user_expression_errors_are_reported_not_crashed():
    """a C expression that does not compile -> PCG_E_JIT with the compiler's message; names outside the whitelist are
    rejected on the host before anything is compiled"""
    _torch()
    from pcgym_amd import VecEnv, _lib

    p = copy.deepcopy(SC.scenarios()["cstr_expr_cons_raw"]["env_params"])
    p["constraints"] = {"expr": ["T - ", "Ca"]}  # syntactically broken C
    with pytest.raises(_lib.PcgError) as ei:
        VecEnv(p, n_envs=8)
    assert ei.value.status == -7 and b"error" in _lib.load().pcg_last_jit_log()
    p["constraints"] = {"expr": ["system(T)"]}
    with pytest.raises(ValueError):
        VecEnv(p, n_envs=8)


def test_host_gather_delivers_every_step_in_order():
    """pcgym_amd.HostGather: device-to-device snapshot + pinned D2H on a second stream, two slots in flight -- every
    step's obs / rew / done arrives on the host unchanged while the step loop runs ahead"""
    torch = _torch()
    import bench as BN
    from pcgym_amd import HostGather, VecEnv

    B = 1 << 16
    env = VecEnv(BN.workload_params(), n_envs=B, seed=2, auto_reset=True)
    ref = VecEnv(BN.workload_params(), n_envs=B, seed=2, auto_reset=True)
    env.reset()
    ref.reset()
    g = HostGather(env)
    assert g.bytes_per_step == B * (3 * 8 + 8 + 1)
    gen = torch.Generator(device=env.device).manual_seed(4)
    acts = 2 * torch.rand((70, 1, B), generator=gen, device=env.device, dtype=torch.float64) - 1
    pending = []
    for i in range(70):  # across an episode boundary (59 steps)
        env.step(acts[i])
        pending.append((i, g.push()))
        if len(pending) == 2:  # consume one step late: the loop stays one step ahead of the bus
            k, slot = pending.pop(0)
            out = g.wait(slot)
            ref.step(acts[k])
            assert torch.equal(out["obs"], ref.obs_soa.cpu()) and torch.equal(out["rew"], ref.rew.cpu()), k
            assert torch.equal(out["done"], ref.done.cpu()), k
    env.close()
    ref.close()


# ---- stiff-capable integrator (the reference integrates with CVODES BDF) ----------------
def test_rosenbrock_integrator_on_a_stiffened_column_through_the_hip_path():
    """The extraction column with hold-ups / 100 (|lambda| dt ~ 24,000): with a plan's usual step budget the explicit
    pair reports failure (status byte, NaN state) for most of the action box, the Rosenbrock pair integrates every env
    in ~250 steps, identically to its oracle twin, and agrees with the explicit pair given an unbounded budget."""
    torch = _torch()
    from oracle import oracle as O
    from pcgym_amd import VecEnv
    from pcgym_amd import _abi as abi

    B = 640
    sc = SC.scenarios()["me_canonical"]
    rng = np.random.default_rng(4)

    from pcgym_amd import models as M

    class multistage_extraction:  # the reference's custom_model hook (pcgym.py:150-153) with changed parameter values
        def info(self):
            d = M.get_model("multistage_extraction").info()
            d["parameters"].update(Vl=0.05, Vg=0.05)
            return d

    def build(integ, max_steps):
        p = copy.deepcopy(sc["env_params"])
        p.update(integrator=integ, rtol=1e-6, atol=1e-8, max_steps=max_steps, custom_model=multistage_extraction())
        return p

    acts = rng.uniform(-1, 1, (3, 2, B))
    res = {}
    for integ, ms in (("rodas3", 2000), ("dopri5", 2000), ("dopri5", 40000)):
        env = VecEnv(build(integ, ms), n_envs=B, seed=3)
        assert env.spec.model.parameters["Vl"] == 0.05 and env.spec.integrator == integ
        orc = O.OracleEnv(env.spec, B, seed=3)
        env.reset()
        orc.reset()
        for i in range(3):
            env.step(torch.tensor(acts[i], device=env.device))
            orc.step(acts[i])
            if integ == "rodas3":
                _cmp(env, orc, 5e-8, ("stiff column", i))
        torch.cuda.synchronize()
        res[(integ, ms)] = (env.x.cpu().numpy(), env.status.cpu().numpy(), env.nsteps.cpu().numpy())
        assert np.array_equal(env.status.cpu().numpy(), orc.status)
        env.close()
    xr, sr, nr = res[("rodas3", 2000)]
    xb, sb, _ = res[("dopri5", 2000)]
    xd, sd, nd = res[("dopri5", 40000)]
    assert not sr.any() and np.isfinite(xr).all() and nr.sum(axis=0).max() < 600
    # (a failed env stays failed: its NaN state makes the following steps fail too, and the sticky byte keeps the last code)
    assert (sb != abi.PCG_ST_OK).mean() > 0.5 and np.isnan(xb[:, sb != 0]).all() and np.isfinite(xb[:, sb == 0]).all()
    assert not sd.any() and nd.sum(axis=0).mean() > 2000
    assert np.max(np.abs(xr - xd)) <= 2e-5


def test_rosenbrock_plans_are_refused_where_no_kernel_exists():
    torch = _torch()
    from pcgym_amd import VecEnv, collect_rollouts

    p = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
    p.update(integrator="rodas3", rtol=1e-6, atol=1e-8)
    env = VecEnv(p, n_envs=256, seed=1)
    env.reset()
    a = torch.zeros((5, env.spec.na, 256), dtype=torch.float64, device=env.device)
    with pytest.raises(RuntimeError, match="pcg_rollout"):
        env.rollout(a)
    # the collector steps instead (no fused kernel), and an open-loop episode stays finite
    acts = torch.zeros((env.spec.N, env.spec.na, 256), dtype=torch.float64, device=env.device)
    out = collect_rollouts(env, actions=acts)
    assert torch.isfinite(out["x"]).all() and torch.isfinite(out["r"]).all()
    env.close()
    # ... but its steps record into a HIP graph like any other plan's (10-state model: 51 KB of LDS matrices per wave,
    # beyond the default dynamic-LDS limit, raised per launch)
    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p.update(integrator="rodas3", rtol=1e-6, atol=1e-8)
    e1, e2 = VecEnv(copy.deepcopy(p), n_envs=300, seed=1), VecEnv(copy.deepcopy(p), n_envs=300, seed=1)
    e1.reset(), e2.reset()
    gen = torch.Generator(device="cuda").manual_seed(3)
    acts = [torch.rand((2, 300), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(4)]
    g = e2.capture_steps(acts)
    for a in acts:
        e1.step(a)
    g.replay()
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.rew, e2.rew) and torch.equal(e1.obs_soa, e2.obs_soa)
    g.destroy()
    e1.close(), e2.close()


def test_rosenbrock_integrator_inside_a_run_time_compiled_step_kernel():
    """User expressions are compiled (hipRTC) into the plan's own general kernel, whatever its integrator: with
    'rodas3' the states are bit-identical to the ahead-of-time kernel's, and the compiled constraint rows are the
    NumPy evaluation of the same expressions on those states."""
    torch = _torch()
    from pcgym_amd import VecEnv

    sc = SC.scenarios()["cstr_expr_cons_raw"]
    p = copy.deepcopy(sc["env_params"])
    p.update(integrator="rodas3", rtol=1e-7, atol=1e-9)
    q = copy.deepcopy(p)
    q.pop("constraints")
    for k in ("done_on_cons_vio", "r_penalty"):
        q.pop(k, None)
    B = 515
    for per_env_t in (False, True):
        ea = VecEnv(copy.deepcopy(p), n_envs=B, seed=2, per_env_t=per_env_t)
        eb = VecEnv(copy.deepcopy(q), n_envs=B, seed=2, per_env_t=per_env_t)
        assert ea.spec.user_cons_src is not None and eb.spec.user_cons_src is None
        ea.reset(), eb.reset()
        gen = torch.Generator(device="cuda").manual_seed(8)
        for i in range(5):
            a = torch.rand((1, B), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
            if not ea.spec.normalise_a:
                a = torch.tensor(ea.spec.a_low[0] + (a.cpu().numpy() + 1) / 2 * (ea.spec.a_high[0] - ea.spec.a_low[0]), device="cuda")
            _, _, _, _, info = ea.step(a)
            eb.step(a)
            assert torch.equal(ea.x, eb.x) and torch.equal(ea.nsteps, eb.nsteps), (per_env_t, i)
            x = ea.x.cpu().numpy()
            u = a.cpu().numpy()[0]
            if ea.spec.normalise_a:
                u = (u + 1) * (ea.spec.a_high[0] - ea.spec.a_low[0]) / 2 + ea.spec.a_low[0]
            ref = sc["ref_env_params"]["constraints"]
            sp = ea.spec.sp[0, min(i, ea.spec.N - 1)]
            want = np.stack([np.asarray(ref(np.array([x[0, b], x[1, b], sp]), np.array([u[b]])), dtype=float).reshape(-1)
                             for b in range(0, B, 37)], axis=1)
            got = info["g"].cpu().numpy()[:, ::37]
            assert np.allclose(got, want, rtol=1e-9, atol=1e-9 * np.max(np.abs(want))), (per_env_t, i)
        ea.close(), eb.close()


@pytest.mark.parametrize("name,B", [("me_canonical", 1 << 17), ("me_canonical", 3000), ("me_reactive", 70000)])
def test_step_graph_with_adaptive_kernels(name, B):
    """pcg_graph_* around adaptive plans: the work-queue kernel (well-filled tiles) and the classic adaptive kernel (thin
    ones) record and replay like the fixed-step kernels -- bit-identical to stepping"""
    torch = _torch()
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()[name]["env_params"])
    p["integrator"] = "dopri5"
    e1, e2 = VecEnv(copy.deepcopy(p), n_envs=B, seed=1), VecEnv(copy.deepcopy(p), n_envs=B, seed=1)
    e1.reset(), e2.reset()
    gen = torch.Generator(device="cuda").manual_seed(1)
    acts = [torch.rand((e1.spec.na, B), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(4)]
    g = e2.capture_steps(acts)
    for a in acts:
        e1.step(a)
    g.replay()
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.nsteps, e2.nsteps) and torch.equal(e1.rew, e2.rew)
    assert torch.equal(e1.obs_soa, e2.obs_soa) and not e2.status.any()
    g.destroy()
    e1.close(), e2.close()


def test_collector_steps_a_rosenbrock_plan_that_carries_a_reward_expression():
    """multistage_extraction defaults to a Rosenbrock pair; with a reward expression (or a traced custom_reward callable)
    the plan runs from its run-time compiled module, which has no fused rollout kernel for the Rosenbrock pairs --
    collect_rollouts(env, actions=...) used to raise PCG_E_UNSUPPORTED there instead of stepping."""
    torch = _torch()
    from pcgym_amd import VecEnv, collect_rollouts

    p = copy.deepcopy(SC.scenarios()["me_canonical"]["env_params"])
    p["custom_reward"] = {"expr": "-1e2*(X5 - SP_X5)*(X5 - SP_X5)"}
    env = VecEnv(copy.deepcopy(p), n_envs=192, seed=4)
    assert env.spec.integrator == "rodas5" and env.spec.user_reward_src
    ref = VecEnv(copy.deepcopy(p), n_envs=192, seed=4)
    gen = torch.Generator(device="cuda").manual_seed(8)
    acts = 0.3 * (2 * torch.rand((env.spec.N, env.spec.na, 192), generator=gen, device="cuda", dtype=torch.float64) - 1) - 0.6
    out = collect_rollouts(env, actions=acts)
    assert out["r"].shape == (1, env.spec.N, 192) and torch.isfinite(out["x"]).all() and torch.isfinite(out["r"]).all()
    ref.reset()
    for i in range(env.spec.N - 1):  # the same episode, stepped by hand
        o, r, d, _, _ = ref.step(acts[i])
        assert torch.equal(out["r"][0, i + 1], r), i
    # the expression really is the reward: -1e2 (X5 - SP)^2 on the recorded (physical) states, reward against SP[t_new]
    sp = torch.tensor(np.asarray(p["SP"]["X5"], dtype=float), device="cuda")
    want = -1e2 * (out["x"][8, 1:] - sp[1:, None]) ** 2
    assert torch.allclose(out["r"][0, 1:], want, rtol=1e-10, atol=1e-12)
    env.close(), ref.close()


@pytest.mark.parametrize("integrator", ["rk4", "dopri5"])
def test_fused_rollout_with_per_env_parameters_matches_stepping(integrator):
    """pcg_rollout refused plans with per-env uncertain parameters (the reference resamples them at
    every reset(), pcgym.py:300-316, then rolls the episode, policy_evaluation.py:71-130).  The general rollout kernel now
    has the per-env-parameter form of the step kernels: same states, observations (incl. the parameter slots) and rewards
    as stepping, and collect_rollouts() takes the fused path."""
    torch = _torch()
    from pcgym_amd import VecEnv, collect_rollouts

    p = copy.deepcopy(SC.scenarios()["cstr_canonical"]["env_params"])
    p.pop("noise", None), p.pop("noise_percentage", None)
    # (small spreads: no env ignites -- a fixed step on the ignition branch ends in NaN on both paths alike)
    p.update(uncertainty_percentages={"UA": 0.02, "x0": [0.01, 0.003], "Caf": 0.02}, distribution="uniform",
             uncertainty_bounds={"low": np.array([4e4, 0.9]), "high": np.array([6e4, 1.1])}, integrator=integrator)
    if integrator == "rk4":
        p["substeps"] = 4
    B = 1024
    e1, e2 = VecEnv(copy.deepcopy(p), n_envs=B, seed=31), VecEnv(copy.deepcopy(p), n_envs=B, seed=31)
    assert e1.spec.nunc == 2
    N = e1.spec.N
    gen = torch.Generator(device="cuda").manual_seed(2)
    acts = 2 * torch.rand((N, 1, B), generator=gen, device="cuda", dtype=torch.float64) - 1
    e1.reset(), e2.reset()
    assert torch.equal(e1.p_unc, e2.p_unc) and float(e1.p_unc[0].std()) > 3e2  # UA really differs from env to env
    obs_seq, rew_seq = e1.rollout(acts[:N - 1], collect_obs=True)
    for t in range(N - 1):
        o, r, d, _, _ = e2.step(acts[t])
        # (the two kernels contract multiply-adds differently; envs whose UA draw puts them near ignition amplify that
        # last-bit difference step by step: tight for the bulk, 1e-7 for every env)
        assert torch.isfinite(e2.obs_soa).all(), t
        d = (e2.obs_soa - obs_seq[t]).abs()
        assert float(d.median()) <= 1e-13 and torch.allclose(e2.obs_soa, obs_seq[t], rtol=1e-9, atol=1e-11), (t, float(d.max()))
        assert torch.allclose(r, rew_seq[t], rtol=1e-8, atol=1e-10), t
    assert torch.allclose(e1.x, e2.x, rtol=1e-9, atol=1e-11) and not e1.status.any()
    # the collector: fused now, in the reference's axis order, parameter slots included in x
    e3 = VecEnv(copy.deepcopy(p), n_envs=B, seed=31)
    out = collect_rollouts(e3, actions=acts)
    assert out["x"].shape == (e3.spec.nobs, N, B) and torch.isfinite(out["x"]).all()
    assert torch.allclose(out["r"][0, 1:], rew_seq.reshape(N - 1, B), rtol=1e-11, atol=1e-13)  # (the same kernel twice)
    for e in (e1, e2, e3):
        e.close()


@pytest.mark.parametrize("name,B,kw", [("me_canonical", 1 << 18, {}), ("me_canonical", 300001, dict(per_env_t=True, auto_reset=True)),
                                       ("me_reactive", 1 << 18, {}), ("me_reactive", 290011, {})])
def test_work_queue_launch_shapes_of_a_full_batch(name, B, kw):
    """The launch shapes the host only picks for well-filled CUs -- 512-thread workgroups on one tile of up to 2048 slots
    (10-state cascade), two LDS-resident half tiles per workgroup (20-state cascade) -- against the classic kernel on the
    same envs: identical step counts, states to round-off.  (The ragged tests above stay below the batch sizes that select
    these shapes.)"""
    torch = _torch()
    from pcgym_amd import VecEnv

    p = copy.deepcopy(SC.scenarios()[name]["env_params"])
    p["integrator"] = "dopri5"
    q = VecEnv(copy.deepcopy(p), n_envs=B, seed=5, **kw)
    cl = VecEnv(copy.deepcopy(p), n_envs=B, seed=5, variant=1, **kw)
    q.reset()
    cl.reset()
    gen = torch.Generator(device=q.device).manual_seed(8)
    for i in range(3):
        at = 2 * torch.rand((q.spec.na, B), generator=gen, device=q.device, dtype=torch.float64) - 1
        o1, r1, d1, _, _ = q.step(at)
        o2, r2, d2, _, _ = cl.step(at)
        H.adaptive_check(q.spec.model.name, q.x.cpu().numpy(), cl.x.cpu().numpy(), q.nsteps.cpu().numpy(),
                         cl.nsteps.cpu().numpy(), (name, i), tol=1e-12)
        assert torch.equal(d1, d2) and torch.equal(q.status, cl.status), (name, i)
        assert torch.allclose(o1, o2, rtol=1e-11, atol=1e-11) and torch.allclose(r1, r2, rtol=1e-9, atol=1e-11), (name, i)
    q.close()
    cl.close()


@pytest.mark.parametrize("S,threads", [(512, 256), (1024, 256), (2048, 256), (512, 512), (1024, 512), (2048, 512)])
def test_tile_sort_of_the_work_queue_kernel(S, threads):
    """The step results do not depend on the order of a tile, so a sort that does not sort would only make launches slower:
    the bitonic network over registers, cross-lane reads and LDS (sort_tile) on its own, every (tile width, workgroup size)
    the step kernels instantiate, against numpy -- random keys with the slot index in the low bits as the kernel packs them,
    sorted input, reversed input, and the padded tail of a partly filled tile."""
    torch = _torch()
    from pcgym_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(S + threads)
    ntiles = 37
    words = np.empty((ntiles, S), dtype=np.uint32)
    for t in range(ntiles):
        key = rng.integers(0, 1 << 21, S, dtype=np.uint32)
        if t == 1:
            key = np.sort(key)
        if t == 2:
            key = np.sort(key)[::-1].copy()
        if t == 3:
            key[:] = 7  # equal keys: the slot index alone decides
        w = (key << np.uint32(11)) | np.arange(S, dtype=np.uint32)
        if t == 4:  # a tile with n < S real slots: the padding words are the bare slot indices (key 0)
            w[S // 3:] = np.arange(S // 3, S, dtype=np.uint32)
        words[t] = w
    d = torch.tensor(words.view(np.int32), device="cuda")
    rc = lib.pcg_test_sort_tile(d.data_ptr(), S, threads, ntiles, None)
    assert rc == 0
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.uint32)
    want = -np.sort(-words.astype(np.int64), axis=1)
    assert np.array_equal(got.astype(np.int64), want)
    assert lib.pcg_test_sort_tile(d.data_ptr(), 768, threads, ntiles, None) != 0  # not a width the kernels use
