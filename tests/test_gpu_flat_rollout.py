"""The barrier-free fused rollout of the cstr's default plan (pcg_rollout_flat.hpp: PCG_INT_T5G in two passes -- every env
rolled while its guard trusts the fixed step, the handed-over envs each on a lane of their own through the adaptive pair)
against what it replaces, policy_evaluation.py:71-130 stepped env by env: bitwise equal to T pcg_step launches and to
the single-kernel rollout, whatever order the lanes finish in, and within round-off of the oracle's per-env loop."""
import copy
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _launched(lib, what):
    n = lib.pcg_coverage_names(None, 0, 0)
    if n <= 1:
        return False
    buf = C.create_string_buffer(int(n))
    lib.pcg_coverage_names(buf, n, 0)
    return what in buf.value.decode()


def _params(extras):
    import bench

    _, p, _, _, _ = bench.single_workload("cstr_safe")  # the headline's envs on the full x0 box, the model's default plan
    p = copy.deepcopy(p)
    if extras:
        p.update(constraints=lambda x, u: np.array([x[1] - 345.0, 0.72 - x[0]]).reshape(-1,), done_on_cons_vio=False,
                 r_penalty=True, noise=True, noise_percentage=0.001)
    return p


@pytest.mark.parametrize("B,extras", [(1 << 17, False), (70_001, True), (1 << 20, False)])  # the last: BASELINE configs[1]'s batch
def test_flat_rollout_equals_stepping_and_the_oracle(B, extras):
    import torch
    from oracle import oracle as O
    from pcgym_amd import VecEnv

    p = _params(extras)
    envs = [VecEnv(copy.deepcopy(p), n_envs=B, seed=7) for _ in range(3)]
    e_flat, e_step, e_one = envs
    spec = e_flat.spec
    assert spec.integrator == "tsit5g"
    T = spec.N - 1
    gen = torch.Generator(device="cuda").manual_seed(11)
    acts = 2 * torch.rand((T, spec.na, B), generator=gen, device="cuda", dtype=torch.float64) - 1
    for e in envs:
        e.reset()
    x_start = e_flat.x.clone()
    # (a) T step launches (the two-launch form of the guarded plan at this batch size)
    obs_s, rew_s, hot_steps = [], [], 0
    for i in range(T):
        e_step.step(acts[i])
        obs_s.append(e_step.obs_soa.clone()), rew_s.append(e_step.rew.clone())
        hot_steps += int((e_step.nsteps.sum(dim=0) > 0).sum().item())
    assert hot_steps > 0.2 * B * T, "the x0 box of this test is supposed to ignite a third of the batch"
    # (b) the barrier-free rollout
    oq, rq = e_flat.rollout(acts, collect_obs=True, collect_rew=True)
    torch.cuda.synchronize()
    assert _launched(e_flat._lib, "rollout_kernel_hot"), "the two-pass rollout was not taken"
    # (c) the single-kernel rollout (one env per lane for all T steps, fallback inside the lane)
    os.environ["PCG_NO_FLAT"] = "1"
    try:
        o1, r1 = e_one.rollout(acts, collect_obs=True, collect_rew=True)
        torch.cuda.synchronize()
    finally:
        del os.environ["PCG_NO_FLAT"]
    for name, got_o, got_r, env in (("flat", oq, rq, e_flat), ("single kernel", o1, r1, e_one)):
        assert torch.equal(env.x, e_step.x), f"{name}: final state differs from stepping"
        assert torch.equal(env.status, e_step.status) and int(env.status.sum().item()) == 0
        for i in range(T):
            assert torch.equal(got_r[i], rew_s[i]), f"{name}: reward of step {i} differs from stepping"
            assert torch.equal(got_o[i], obs_s[i]), f"{name}: observation of step {i} differs from stepping"
        assert torch.equal(env.obs_soa, e_step.obs_soa) and torch.equal(env.rew, e_step.rew) and torch.equal(env.done, e_step.done)
        assert torch.equal(env.nsteps, e_step.nsteps)  # (the last step's counts)
    # (d) windows of the batch against the oracle's per-env loop
    W = 96
    for lo in (0, B // 2 - 31, B - W):
        orc = O.OracleEnv(spec, W, seed=7, env_offset=lo)
        orc.reset()
        assert np.allclose(orc.x, x_start[:, lo:lo + W].cpu().numpy(), rtol=1e-14)  # (the reset kernel against its twin)
        orc.x[:] = x_start[:, lo:lo + W].cpu().numpy()
        for i in range(T):
            oc, rc, _ = orc.step(acts[i][:, lo:lo + W].cpu().numpy())
            assert np.allclose(rq[i][lo:lo + W].cpu().numpy(), rc, rtol=1e-7, atol=1e-7 * (1 + np.abs(rc).max()))
            if not extras:  # (with noise the observation carries the noise twin's fp32 normal variates: compared via the reward)
                assert np.allclose(oq[i][:, lo:lo + W].cpu().numpy(), oc, rtol=1e-8, atol=1e-9)
        xs = np.maximum(np.abs(orc.x), 1e-9)
        assert np.max(np.abs(e_flat.x[:, lo:lo + W].cpu().numpy() - orc.x) / xs) <= 1e-8
    for e in envs:
        e.close()


def test_collect_rollouts_takes_the_flat_path():
    """rollout.collect_rollouts (the reference's axis order, policy_evaluation.py:155-197) on the default cstr plan"""
    import torch
    from pcgym_amd import VecEnv, collect_rollouts

    B = 1 << 17
    p = _params(False)
    env = VecEnv(copy.deepcopy(p), n_envs=B, seed=3)
    ref = VecEnv(copy.deepcopy(p), n_envs=B, seed=3)
    N = env.spec.N
    gen = torch.Generator(device="cuda").manual_seed(5)
    acts = 2 * torch.rand((N, 1, B), generator=gen, device="cuda", dtype=torch.float64) - 1
    d = collect_rollouts(env, actions=acts)
    assert _launched(env._lib, "rollout_kernel_hot")
    assert d["x"].shape == (env.spec.nobs, N, B) and d["r"].shape == (1, N, B)
    ref.reset()
    for i in range(N - 1):
        ref.step(acts[i])
    assert torch.equal(env.x, ref.x)
    assert torch.equal(d["r"][0, N - 1], ref.rew)
    env.close(), ref.close()
