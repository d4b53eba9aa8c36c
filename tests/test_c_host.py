"""GPU: the C ABI driven from plain C (examples/c_host/step_demo.c: gcc, hipMalloc'ed buffers, no Python / torch in
the process) gives the same episode as pcgym_amd.VecEnv on the same configuration and actions."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_matches_python_host(tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test collected without a GPU")
    from pcgym_amd import VecEnv

    exe = str(tmp_path / "step_demo")
    lib_dir = os.path.join(ROOT, "pc-gym_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_host", "step_demo.c"), "-L" + lib_dir, "-lpcgym_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    B, N = 4096, 60
    out = subprocess.run([exe, str(B)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = {l.split()[0]: [float(v) for v in l.split()[1:]] for l in out.stdout.splitlines()}

    p = {"model": "cstr", "N": N, "tsim": 26,
         "SP": {"Ca": [0.85] * (N // 3) + [0.9] * (N // 3) + [0.87] * (N - 2 * (N // 3))},
         "o_space": {"low": np.array([0.7, 300.0, 0.8]), "high": np.array([1.0, 350.0, 0.9])},
         "a_space": {"low": np.array([295.0]), "high": np.array([302.0])}, "x0": np.array([0.8, 330.0, 0.8]),
         "r_scale": {"Ca": 1e3}, "normalise_a": True, "normalise_o": True, "integrator": "rk4", "substeps": 4}
    env = VecEnv(p, n_envs=B, seed=1, track_status=False)  # the C demo passes no status buffer
    assert got["bytes_per_env_step"][0] == env.bytes_per_env_step == 73
    env.reset()
    e = np.arange(B)
    ret = 0.0
    for t in range(N - 1):
        a = -1.0 + 2.0 * ((e * 7 + t * 13) % 101) / 100.0
        obs, rew, done, _, _ = env.step(torch.tensor(a.reshape(1, B), device=env.device))
        ret += float(rew.sum())
    o = env.obs_soa.cpu().numpy()
    assert abs(got["return_sum"][0] - ret) <= 1e-9 * abs(ret)
    assert np.allclose(got["obs_sum"], o.sum(axis=1), rtol=1e-12)
    assert np.allclose(got["obs_env0"], o[:, 0], rtol=0, atol=1e-15)  # same kernels, same inputs: identical
    assert got["n_done"][0] == B == int(env.done.sum())
    env.close()


def test_plain_c_host_with_a_user_model(tmp_path):
    """PCG_MODEL_USER from C: the right-hand side as a C string in pcg_env_cfg, compiled by pcg_plan_create; the same
    episode as the Python host with the declarative custom_model"""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test collected without a GPU")
    from pcgym_amd import VecEnv
    from pcgym_amd import _abi as abi

    exe = str(tmp_path / "user_model_demo")
    lib_dir = os.path.join(ROOT, "pc-gym_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_host", "user_model_demo.c"), "-L" + lib_dir, "-lpcgym_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    B, N = 2048, 30
    out = subprocess.run([exe, str(B), os.path.join(lib_dir, "csrc")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PCG_JIT_CACHE=str(tmp_path / "jit")))
    assert out.returncode == 0, out.stderr
    got = {l.split()[0]: [float(v) for v in l.split()[1:]] for l in out.stdout.splitlines()}
    cm = {"states": ["X", "S"], "inputs": ["D"], "parameters": {"mumax": 0.53, "Ks": 0.12, "Ki": 22.0, "Y": 0.4, "Sf": 4.0},
          "aux": {"mu": "mumax*S/(Ks + S + S*S/Ki)"}, "rhs": ["(mu - D)*X", "D*(Sf - S) - mu*X/Y"]}
    p = {"custom_model": cm, "N": N, "tsim": 15.0, "x0": np.array([1.2, 0.6, 1.4]),
         "SP": {"X": [1.4] * (N // 2) + [1.0] * (N - N // 2)}, "r_scale": {"X": 10.0},
         "a_space": {"low": np.array([0.0]), "high": np.array([0.45])},
         "o_space": {"low": np.zeros(3), "high": np.array([3.0, 6.0, 3.0])}, "normalise_a": True, "normalise_o": True,
         "integrator": "dopri5", "rtol": 1e-8, "atol": 1e-8}
    env = VecEnv(p, n_envs=B, seed=1)
    env.reset()
    e = np.arange(B)
    ret = 0.0
    for t in range(N - 1):
        a = -1.0 + 2.0 * ((e * 5 + t * 11) % 97) / 96.0
        obs, rew, done, _, _ = env.step(torch.tensor(a.reshape(1, B), device=env.device))
        ret += float(rew.sum())
    o = env.obs_soa.cpu().numpy()
    # (the C string and the Python expressions are different texts of the same formulas: agreement to round-off)
    assert abs(got["return_sum"][0] - ret) <= 1e-9 * abs(ret)
    assert np.allclose(got["obs_sum"], o.sum(axis=1), rtol=1e-10)
    assert np.allclose(got["obs_env0"], o[:, 0], rtol=1e-10, atol=1e-12)
    assert got["n_failed"][0] == 0 and not env.status.any()
    assert got["bad_source_status"][0] == abi.PCG_E_JIT and got["bad_source_log_mentions_nonsense"][0] == 1
    env.close()
