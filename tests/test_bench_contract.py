"""GPU: bench.py prints ONE JSON line with the contract's keys, single-rank and through the N-rank launch path
(two ranks sharing this box's GPU over gloo -- the RCCL run with one rank per GPU is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_rank_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "70", "--warmup", "11",
                        "--preheat-ms", "20"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 70 and d["warmup"] == 11 and d["unit"] == "env-steps/s"
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["config"]["sane"]
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 1.0
    assert abs(d["value"] - d["config"]["global_envs"] * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert cb["kind"] == "port" and cb["unit"] == "env-steps/s" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["reference_shaped"]["cores"] == 1 and cb["reference_shaped"]["value"] > 0
    assert rf["traffic_measured_in_run"] is False and d["config"]["workload"] == "cstr_b2^20_rk4_fp64"
    assert d["config"]["ranks_seen"] == 1


@pytest.mark.parametrize("wl,batch,steps", [("mixed", 30000, 8), ("mixed", 30000, 59), ("me10", 8192, 6), ("me10_ros4", 8192, 6), ("me10_ros5", 8192, 6), ("cryst", 8192, 6), ("cryst_cv8", 8192, 6), ("cstr_safe", 65536, 20), ("four_tank", 65536, 20),
                                            ("cstr_rollout", 65536, 59), ("cstr_rollout", 65536, 100), ("cstr_unc", 65536, 20)])
def test_other_workloads_line(wl, batch, steps):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--batch", str(batch), "--steps",
                        str(steps), "--warmup", "2", "--preheat-ms", "10", "--no-cpu-baseline"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    if wl == "cstr_rollout":  # a launch is a whole 59-step episode: steps and warm-up are rounded up to whole episodes
        assert d["steps"] == -(-steps // 59) * 59 and d["warmup"] == 59 and d["roofline"]["env_steps_per_launch"] == 59
        assert d["roofline"]["algorithmic_bytes_per_env_step"] == 8 * (1 + 3 + 1)
    else:
        assert d["steps"] == steps and d["warmup"] == 2
    assert KEYS <= set(d) and d["config"]["sane"]
    assert abs(d["value"] - d["config"]["global_envs"] * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "fp64_valu", "chain") and 0 < rf["frac"] < 1.0
    assert (rf["bound"] == "chain") == (wl == "cstr_safe")  # the adaptive plan of a cheap model waits for its heaviest env
    if wl == "cstr_safe":
        assert rf["chain"]["heaviest_env_attempts_last_step"] >= 1
    if wl != "mixed":
        assert rf["copy_ceiling_GBps"] > 500 and rf["frac_of_copy_ceiling"] > 0  # the preheat's device copy, timed
    if wl == "mixed":
        assert [s["segment"] for s in rf["segments"]] == ["cstr", "four_tank", "multistage_extraction"]
        assert d["config"]["envs_per_gpu"] == 3 * ((batch // 3) & ~1)
        assert ("graph" in d["config"]["launch"]) == (steps % 59 == 0)  # whole episodes replay as one graph per segment


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_rank_launch_path():
    env = dict(os.environ, PCG_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "70", "--warmup", "11", "--preheat-ms", "20"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" not in d  # the CPU leg runs at N = 1 only
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2 * d["config"]["envs_per_gpu"]
    assert abs(d["value"] - d["config"]["global_envs"] * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["config"]["collective_backend"] == "gloo" and d["config"]["ranks_seen"] == 2


def test_rccl_branch_with_one_rank():
    """RCCL refuses two ranks on one device, so the multi-rank tests here share this box's GPU over gloo -- and the lines of
    bench.py that only run under backend "nccl" (device-bound process group, the reductions' tensors on the device, the
    barrier inside the timed brackets) would first execute on the driver's 8-GPU node.  They run here with ONE rank under
    torch.distributed.run and the communicator path forced."""
    env = dict(os.environ, PCG_BENCH_FORCE_DIST="1")
    env.pop("PCG_BENCH_BACKEND", None)
    for extra in ([], ["--workload", "mixed", "--batch", "30000"]):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                            "--gpus", "1", "--steps", "20", "--warmup", "5", "--preheat-ms", "20", "--no-cpu-baseline"] + extra,
                           capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        d = _json_line(r.stdout)
        assert d["n_gpus"] == 1 and d["config"]["ranks_seen"] == 1 and d["config"]["sane"]
        assert d["config"]["collective_backend"] == "rccl"


def test_two_rank_mixed_workload():
    env = dict(os.environ, PCG_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--workload", "mixed", "--batch", "30000", "--steps", "8", "--warmup", "2",
                        "--preheat-ms", "10"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2 * d["config"]["envs_per_gpu"] and d["config"]["sane"]


def test_gpus_n_as_typed_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with no launcher and no WORLD_SIZE (how the driver types its N = 1 command): bench.py
    becomes the launcher itself -- one JSON line, two ranks seen by the communicator."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PCG_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "70", "--warmup", "11",
                        "--preheat-ms", "20"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["collective_backend"] == "gloo"
    assert d["config"]["global_envs"] == 2 * d["config"]["envs_per_gpu"] and d["config"]["sane"]


@pytest.mark.parametrize("wl,batch,steps", [("cstr", 65536, 70), ("mixed", 30000, 8), ("cstr_safe", 65536, 20),
                                            ("four_tank", 65536, 20), ("me10_ros4", 16384, 6), ("me10_ros5", 16384, 6), ("me20", 16384, 4),
                                            ("cryst_cv8", 16384, 10)])
def test_eight_rank_launch_path_on_one_device(wl, batch, steps):
    """No 8-GPU node is available to the builder: the EIGHT-rank code path of the command the driver types
    (`python bench.py --gpus 8`: bench.py starts its own ranks) runs here with the ranks sharing this box's GPU over gloo,
    for the headline and for the mixed shard.  Checked: one JSON line, eight ranks seen by the communicator, the global batch
    = 8 x the per-GPU batch, every rank's shard starts where the previous one ends (the RNG key of an env is its global
    index: shards must not overlap), and the host-side cost of the launch loop with eight Python ranks on one host is
    reported on the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PCG_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", wl, "--batch", str(batch),
                        "--steps", str(steps), "--warmup", "6", "--preheat-ms", "10"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    c = d["config"]
    assert d["n_gpus"] == 8 and c["ranks_seen"] == 8 and c["collective_backend"] == "gloo" and c["sane"]
    assert c["global_envs"] == 8 * c["envs_per_gpu"] and "cpu_baseline" not in d
    assert abs(d["value"] - c["global_envs"] * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    offs = c["rank_first_env"]
    assert len(offs) == 8 and offs[0] == 0 and offs == sorted(offs) and len(set(offs)) == 8
    if wl != "mixed":
        assert offs == [r_ * c["envs_per_gpu"] for r_ in range(8)]
    else:  # global layout [cstr x 8 | four_tank x 8 | ME x 8]: a rank's first env is its slice of the first segment
        n = c["envs_per_gpu"] // 3
        assert offs == [r_ * n for r_ in range(8)]
    assert 0 < c["host_launch_loop_us_per_step_max_over_ranks"] < 5000
    # attribution of a 1 -> N curve: every rank's own wall time per step, in-run kernel time and host-loop cost are on the line
    # (the maximum over ranks is what `value` is computed from), with the NUMA node each rank was pinned to
    pr = c["per_rank"]
    for k in ("ms_per_step", "kernel_avg_us", "host_launch_loop_us_per_step", "numa_node", "pinned_to_gpu_numa_node"):
        assert len(pr[k]) == 8, k
    assert all(v > 0 for v in pr["ms_per_step"]) and all(v > 0 for v in pr["kernel_avg_us"])
    assert abs(max(pr["ms_per_step"]) - d["ms_per_step"]) <= 1e-9 * d["ms_per_step"]
    assert abs(max(pr["host_launch_loop_us_per_step"]) - c["host_launch_loop_us_per_step_max_over_ranks"]) <= 0.011
