#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused HIP step kernels on BASELINE.json's configurations.

Default workload (config.workload = "cstr_b2^20_rk4_fp64", BASELINE configs[1], the configuration `metric` is quoted on):
  cstr (2 states), B = 1,048,576 envs PER GPU, classical RK4, one step per dt = 1 s (= 1/60 of the model's time
  unit, minutes), fp64, normalised actions/observations, SP schedule 0.85 -> 0.9 -> 0.87 (thirds), N = 60,
  r_scale Ca = 1e3, noise off; x0 ~ [U(0.7,1.0), U(310,334)] drawn in the reset kernel (Philox), actions ~ U(-1,1)
  pre-generated on the device (no policy cost), lock-stepped batch.
A "step" = ONE pass of the hot path over the whole batch (pcg_step launches: one per model segment); episodes are
N-1 steps long and the reset that ends each episode is inside the timed region (fused into the episode's last step
launch, pcg_step_autoreset).  value = total env-steps / wall time (max over ranks), whole job.
Launch form (round 6): the consecutive plain steps of an episode are recorded as ONE HIP graph (pcg_graph_*: the same step
kernels with the same arguments) before the timed region and replayed inside it; --eager issues them one by one.

--workload selects the other BASELINE configurations (parity-test cases made measurable; not the headline):
  cstr_safe  the headline's envs / dt / actions under the model's DEFAULT plan (guarded RK4 with adaptive fallback) on the
             full x0 box U(0.7,1.0) x U(310,350) K of SURVEY.md section 8(d) -- reported beside the headline, not as it
  cstr_rollout  section 8(f-1), the rollout collector: each 59-step episode of the headline's envs as ONE fused launch
             (pcg_rollout_strided) writing x (Nx, N, B) / r (1, N, B) in the reference's axis order; a "step" is still one
             env step of the whole batch (--steps is rounded up to whole episodes of 59)
  cstr_safe_rollout  the same collector on cstr_safe's envs (default plan, full x0 box): the barrier-free two-pass rollout
  cstr_unc   section 8(f-3): the headline's envs with per-env model parameters (UA, Caf ~ U(+-5 %)) sampled at reset
  four_tank  four_tank B = 2^20, one Cooper-Verner order-8 step per dt = 1000/60 (the model's default: 11 right-hand sides;
             --integrator rk4 gives the RK4 x5 plan it replaced)
  me10       configs[2]: multistage_extraction (10 states) B = 262,144, adaptive DOPRI5 rtol = atol = 1e-8, dt = 1,
             (L, G) per env over the FULL action box [5,10]..[500,1000], x0 = doc ICs x (1 + 0.05 U(-1,1))
  me10_ros5  the same envs, actions and starts under the model's DEFAULT plan: the fifth-order Rosenbrock pair (Rodas5) with
             the cascade's structured linear algebra and end-point error control, rtol = atol = 8e-8 (the accuracy class of
             the named pair at 1e-8: <= 1e-6 of a 1e-13 solve) -- beside the named RK45 line, not instead of it
  me10_ros4  ... under the fourth-order pair (Rodas4, 3e-8, cooperative rule on): the default plan of rounds 3-4
  me20       the 20-state reactive variant, same protocol
  cryst      configs[3]: crystallization B = 262,144, RK4 x32 per dt = 1, a_delta on
  mixed      configs[4], one shard: 1,048,572 envs = 349,524 each of cstr + Ti ~ N(350, 2) / four_tank /
             multistage_extraction + X0 ~ N(0.6, 0.02), set-point step changes, three plans on three streams, each episode
             of a segment replayed as one HIP graph (pcg_graph_*; --eager: plain launches);
             --gpus 8 = 8,388,576 envs (weak scaling, global env index keys the RNG)

Multi-GPU: the env batch shards embarrassingly (weak scaling, B per GPU fixed); no collective on the hot path --
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.

Extra objects on the JSON line: "roofline" (dominant kernel: algorithmic bytes or flops per launch / kernel time from
hipEvents on the launch stream) and "cpu_baseline" (the C oracle timed on the host cores on a bounded sample, plus a
reference-shaped one-env-at-a-time Python loop; rank 0, N=1 only).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
FP64_PEAK_TFLOPS = 78.6   # fp64 vector peak (no MFMA on this path)
METRIC = "env-steps/sec at batch 2^20 CSTR, 1/2/4/8 MI355X; achieved HBM GB/s vs peak"
# algorithmic fp64 flops per right-hand-side evaluation -- for the fp64-bound workloads
FLOP_PER_RHS = {"multistage_extraction": 65, "multistage_extraction_reactive": 150,
                "crystallization": 60 + 4 * 25 + 12 + 6 * 10}


def flops_per_env_step(model, nx, integrator, attempts, substeps=1):
    """algorithmic fp64 flops of one env step.  RK4: 4 RHS per sub-step; DOPRI5: 2 + 6 per attempt, each with its
    stage combination (2 x 6 nx); Rodas4 on the 10-state cascade: per attempt 6 RHS + 25 stage axpys (2 nx each) + one
    structured factorisation (~60) + 6 structured solves (33 fused multiply-adds = 66 each) + the error norm (4 nx)."""
    f = FLOP_PER_RHS[model]
    if integrator == "rk4":
        return 4 * substeps * (f + 2 * 6 * nx)
    if integrator == "cv8":  # 11 RHS per step; 44 + 5 stage-sum multiply-adds and 11 axpys per state
        return substeps * (11 * f + 2 * (49 + 11) * nx)
    if integrator == "rodas4":
        return attempts * (6 * f + 25 * 2 * nx + 60 + 6 * 66 + 4 * nx)
    if integrator == "rodas5":  # 8 stages: 7 RHS after f(x) + f(x) itself, 17 + 28 stage axpys, 8 solves
        return attempts * (8 * f + 45 * 2 * nx + 60 + 8 * 66 + 4 * nx)
    return (2 + 6 * attempts) * (f + 2 * 6 * nx)


def _thirds(n, a, b, c):
    k = n // 3
    return [a] * k + [b] * k + [c] * (n - 2 * k)


def workload_params(B=None):
    """BASELINE configs[1] (the headline): see the module docstring."""
    import numpy as np

    N = 60
    return {
        "model": "cstr",
        "N": N,
        "tsim": N * (1.0 / 60.0),  # dt = 1 s in a model whose time unit is minutes
        "SP": {"Ca": _thirds(N, 0.85, 0.9, 0.87)},
        "o_space": {"low": np.array([0.7, 300.0, 0.8]), "high": np.array([1.0, 350.0, 0.9])},
        "a_space": {"low": np.array([295.0]), "high": np.array([302.0])},
        "x0": np.array([0.85, 322.0, 0.85]),
        "uncertainty_percentages": {"x0": [0.15 / 0.85, 12.0 / 322.0]},  # -> U(0.7,1.0) x U(310,334)
        "distribution": "uniform",
        "r_scale": {"Ca": 1e3},
        "normalise_a": True,
        "normalise_o": True,
        "integrator": "rk4",
        "substeps": 1,
    }


def mixed_segments(B_shard):
    """BASELINE configs[4], one shard: floor(B/3) envs each of
      cstr with Ti ~ N(350, 2) clipped to [320, 360] (set-point thirds 0.85/0.9/0.87, canonical dt = 26/60, RK4 x4),
      four_tank (set-point step changes h3 0.5 -> 0.1, h4 0.2 -> 0.3, 4tank_train.py:54-57), undisturbed: the model has
        no disturbance input (model_classes.py:926 lists ["None"]),
      multistage_extraction with X0 ~ N(0.6, 0.02) clipped to [0.5, 0.8] (X5 0.3 -> 0.4 -> 0.3; the model's default
        integrator: Rodas5, rtol = atol = 8e-8 -- configs[4] names none; --integrator rodas4 / dopri5 give the lines of
        rounds 3-4 / round 2).
    Returns [(env_params, n_envs)] in the global layout [cstr | four_tank | ME]."""
    import numpy as np

    import scenarios as SC

    S = SC.scenarios()
    n = (B_shard // 3) & ~1  # even: two envs per lane in the small-model kernels
    N = 60
    c = copy.deepcopy(S["cstr_dist_Ti"]["env_params"])
    c.update(N=N, tsim=26.0, SP={"Ca": _thirds(N, 0.85, 0.9, 0.87)}, disturbances={"Ti": np.full(N, 350.0)},
             disturbance_bounds={"low": np.array([320.0]), "high": np.array([360.0])},
             gaussian_disturbances={"Ti": 2.0}, integrator="rk4", substeps=4)
    f = copy.deepcopy(S["four_tank_canonical"]["env_params"])
    m = copy.deepcopy(S["me_dist_cons"]["env_params"])
    for k in ("constraints", "done_on_cons_vio", "r_penalty"):
        m.pop(k, None)
    m.update(N=N, tsim=60.0, SP={"X5": _thirds(N, 0.3, 0.4, 0.3)}, disturbances={"X0": np.full(N, 0.6)},
             disturbance_bounds={"low": np.array([0.5]), "high": np.array([0.8])},
             gaussian_disturbances={"X0": 0.02}, normalise_a=True, normalise_o=True)
    return [(c, n), (f, n), (m, n)]


def single_workload(name):
    """-> (config.workload string, env_params, default B per GPU, default (steps, warmup), action slabs)"""
    import numpy as np

    import scenarios as SC

    S = SC.scenarios()
    if name == "cstr":
        return "cstr_b2^20_rk4_fp64", workload_params(), 1 << 20, (5900, 590), 64
    if name == "cstr_safe":
        # the headline's envs, dt and actions under the model's DEFAULT plan (guarded RK4, adaptive fallback for the envs
        # the guard refuses) on SURVEY.md section 8(d)'s full x0 box U(0.7,1.0) x U(310,350) K -- a third of which ignites,
        # which is why the headline itself (plain RK4 x 1, BASELINE configs[1]) draws T0 below 334 K
        p = workload_params()
        del p["integrator"], p["substeps"]
        p.update(x0=np.array([0.85, 330.0, 0.85]), uncertainty_percentages={"x0": [0.15 / 0.85, 20.0 / 330.0]})
        return "cstr_b2^20_default-plan(tsit5g)_full-x0-box_fp64", p, 1 << 20, (1180, 118), 64
    if name == "cstr_rollout":
        # the rollout collector (policy_evaluation.py:71-130) -- every episode of the headline's
        # envs as ONE fused launch (pcg_rollout_strided: T = N - 1 = 59 steps, state in registers) that writes the
        # trajectories in the reference's axis order x (Nx, N, B), r (1, N, B) from pre-generated actions u (N, Nu, B)
        return "cstr_b2^20_rk4_fused-rollout_T59_reference-axis-order_fp64", workload_params(), 1 << 20, (590, 59), 60
    if name == "cstr_safe_rollout":
        # the same collector under the model's DEFAULT plan on the full x0 box (cstr_safe): the barrier-free rollout in two
        # passes (pcg_rollout_flat.hpp) -- PCG_NO_FLAT=1 gives the single-kernel rollout it replaced
        wl, p, B, kw, _ = single_workload("cstr_safe")
        return wl.replace("default-plan(tsit5g)", "default-plan(tsit5g)_fused-rollout_T59"), p, B, (590, 59), 60
    if name == "cstr_unc":
        # reset-time parameter uncertainty (pcgym.py:212-316) -- the headline's envs with
        # per-env model parameters UA and Caf ~ U(+-5 %) sampled by the reset, read per lane by the step kernel and
        # appended to the observation
        p = workload_params()
        p["uncertainty_percentages"] = dict(p["uncertainty_percentages"], UA=0.05, Caf=0.05)
        p["uncertainty_bounds"] = {"low": np.array([4e4, 0.9]), "high": np.array([6e4, 1.1])}
        return "cstr_b2^20_rk4_per-env-parameters(UA,Caf)_fp64", p, 1 << 20, (1180, 118), 64
    if name == "four_tank":
        return "four_tank_b2^20_cv8x1_fp64", copy.deepcopy(S["four_tank_canonical"]["env_params"]), 1 << 20, (1180, 118), 16
    if name in ("me10_ros4", "me10_ros5"):
        wl, p, B, kw, na = single_workload("me10")
        p.update(integrator="rodas" + name[-1])
        p.pop("rtol"), p.pop("atol")  # the integrator's own default for this model (config.ROS4_TOL: 3e-8, ROS5_TOL: 8e-8)
        return wl.replace("dopri5_1e-8", "rodas4_3e-8_endpoint" if name == "me10_ros4" else "rodas5_8e-8_endpoint"), p, B, kw, na
    if name in ("me10", "me20"):
        p = copy.deepcopy(S["me_canonical" if name == "me10" else "me_reactive"]["env_params"])
        nx = 10 if name == "me10" else 20
        key = list(p["SP"].keys())[0]
        p.update(N=60, tsim=60.0, SP={key: _thirds(60, 0.3, 0.4, 0.3)}, integrator="dopri5", rtol=1e-8, atol=1e-8,
                 uncertainty_percentages={"x0": [0.05] * nx}, distribution="uniform", normalise_a=True,
                 normalise_o=True)
        p.pop("noise", None), p.pop("noise_percentage", None)
        return f"{'multistage_extraction' if name == 'me10' else 'multistage_extraction_reactive'}_b2^18_dopri5_1e-8_fp64", \
            p, 1 << 18, (118, 12), 8
    if name == "cryst":
        p = copy.deepcopy(S["cryst_adelta"]["env_params"])
        p.update(integrator="rk4", substeps=32)
        return "crystallization_b2^18_rk4x32_adelta_fp64", p, 1 << 18, (116, 12), 8
    if name == "cryst_cv8":  # the same envs under the model's default plan: four order-8 steps per env step (44 RHS)
        p = copy.deepcopy(S["cryst_adelta"]["env_params"])
        return "crystallization_b2^18_cv8x4_adelta_fp64", p, 1 << 18, (116, 12), 8
    raise SystemExit(f"unknown workload {name}")


def _cgroup_cpu_quota():
    """CPUs' worth of CPU time the container may use per period (cgroup v2 cpu.max / v1 cfs quota), None if unlimited"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            q, per = float(fq.read()), float(fp.read())
            return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _physical_cores(avail):
    """one thread per physical core among the CPUs of the affinity mask (thread_siblings_list), `avail` if unknown"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        seen = set()
        for c in cpus:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                seen.add(fh.read().strip())
        return max(1, len(seen))
    except (OSError, AttributeError):
        return avail


def cpu_baseline(spec, seconds_target=8.0, threads=None, all_legs=False):
    """The CPU oracle (oracle/pcg_oracle.c: same algorithm, plain C + OpenMP) on the host cores, on a bounded sample of
    the same workload, with a FIXED thread count; beside it the reference-shaped leg: one env at a time through a
    Python loop with an adaptive integrator at the reference's CVODES-default tolerance class (pcgym.py:350-500 +
    integrator.py:90-107 rebuild a solver per step; BASELINE.md section 3.2 measured 207 us/step for the reference
    itself in the build container)."""
    import numpy as np

    import scenarios as SC
    from oracle import oracle as O
    from pcgym_amd.config import EnvSpec

    O.build()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(threads) if threads else min(16, avail)  # fixed: no probing (round-1 probes were unstable on shared hosts)
    Bs, T = 1 << 18, 8
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (T, spec.na, Bs))
    quota = _cgroup_cpu_quota()  # CPUs' worth of time the container may use (None: unlimited)
    phys = _physical_cores(avail)

    def rate(nthreads, budget_s):
        # pinned team (thread i on the (i n_cpu / n)-th CPU of the mask: spread over the sockets); orc_reset writes every
        # buffer first on the partition orc_step uses, so each thread's slice of every row lives on its own NUMA node
        pinned = O.lib().orc_pin_threads(int(nthreads)) == 0
        try:
            env = O.OracleEnv(spec, Bs, seed=1, n_threads=nthreads)
            env.reset()
            env.step(acts[0])  # warm-up (thread pool, page faults)
            env.reset()
            # stop on ELAPSED time, never on a step count sized from one un-throttled step: under a cgroup CPU quota a team
            # larger than the quota runs its first step at full speed and everything after it throttled (round 5: a 4-s leg
            # of 256 threads took 350 s of the driver's 379)
            reps, t0 = 0, time.perf_counter()
            while True:
                if env.t == spec.N - 1:
                    env.reset()
                env.step(acts[reps % T])
                reps += 1
                dt = time.perf_counter() - t0
                if (dt >= budget_s and reps >= 2) or reps >= 4000:
                    break
        finally:
            O.lib().orc_unpin_threads()
        return reps * Bs / dt, reps, dt, pinned

    one, _, _, _ = rate(1, 2.0)
    value, reps, dt, pinned = rate(cores, seconds_target)
    # SURVEY.md section 8(d) asks for "1 thread and all host cores": the same leg on one thread per physical core and on
    # every logical CPU the process may run on.  A container's CPU-time quota (cgroup cpu.max) can be far below the CPUs
    # its affinity mask shows: a team larger than the quota is throttled, not parallel (round 4 printed 2.2e6 env-steps/s
    # on "256 logical CPUs" against 1.5e8 on 16 threads) -- the quota is on the line, and a leg above it says so.
    legs = {}
    for name, n in (("physical_cores", phys), ("all_logical_cpus", avail)):
        above = quota is not None and n > quota + 0.5
        if above and not all_legs:
            # a team above the quota is throttled, not parallel: the leg says nothing about the host (round 5 measured it:
            # 2.0e6 env-steps/s on 256 threads against 1.5e8 on 16) -- skipped unless --cpu-all-legs
            legs[name] = {"value": None, "unit": "env-steps/s", "cores": n, "above_cgroup_cpu_quota": True,
                          "sample": f"not run: a team of {n} is above the container's CPU quota of {quota:g} CPUs "
                                    "(--cpu-all-legs runs it anyway)"}
            continue
        if n == cores:
            v, r, d = value, reps, dt
        else:
            v, r, d, _ = rate(n, 3.0)
        legs[name] = {"value": v, "unit": "env-steps/s", "cores": n,
                      "sample": f"{r} steps x {Bs} envs, pinned OpenMP team of {n} ({d:.1f} s)",
                      **({"above_cgroup_cpu_quota": True} if above else {})}
    all_value = legs["all_logical_cpus"]["value"] or 0.0
    # accuracy of the workload's integrator setting vs a tight adaptive solve, same starts
    p2 = dict(spec.env_params)
    p2.update(integrator="dopri5", rtol=1e-12, atol=1e-14)
    s2 = EnvSpec(p2)
    nb = 4096
    e1 = O.OracleEnv(spec, nb, seed=2, n_threads=cores)
    e2 = O.OracleEnv(s2, nb, seed=2, n_threads=cores)
    e1.reset()
    e2.reset()
    worst = worst_cls = 0.0
    for i in range(10):
        a = act_box(spec) * rng.uniform(-1, 1, (spec.na, nb)) + act_shift(spec)  # the bench's own action distribution
        e2.x[:] = e1.x
        e2.t = e1.t
        e1.step(a)
        e2.step(a)
        worst = max(worst, float(np.nanmax(np.abs(e1.x - e2.x) / np.maximum(np.abs(e2.x), 1e-9))))
        # the same difference in units of the reference's own tolerances (CasADi CVODES defaults: reltol 1e-6, abstol 1e-8):
        # a purely relative figure explodes on components that are physically ~0 (the reactive extraction model's trace
        # species sit at 1e-4: an absolute 1e-8 there reads as 1e-4 relative while being exactly the reference's abstol)
        worst_cls = max(worst_cls, float(np.nanmax(np.abs(e1.x - e2.x) / (1e-6 * np.abs(e2.x) + 1e-8))))
    # reference-shaped leg: README.md:16-55 quick-start cstr (N = 100, 99 steps per episode), ONE env per Python
    # call, adaptive 5(4) pair at rtol 1e-6 / atol 1e-8, one core
    pq = copy.deepcopy(SC.scenarios()["cstr_quickstart"]["env_params"])
    pq.update(integrator="dopri5", rtol=1e-6, atol=1e-8)
    sq = EnvSpec(pq)
    eq = O.OracleEnv(sq, 1, seed=0, n_threads=1)
    rq = np.random.default_rng(0)
    n_steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        eq.reset()
        for i in range(sq.N - 1):
            eq.step(rq.uniform(-1, 1, (1, 1)))
        n_steps += sq.N - 1
    ref_shaped = n_steps / (time.perf_counter() - t0)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), "unknown")
    except OSError:
        pass
    return {
        "value": value,
        "unit": "env-steps/s",
        "cores": cores,
        "one_thread_env_steps_per_s": one,
        "all_host_cpus": legs["all_logical_cpus"],
        "physical_cores": legs["physical_cores"],
        "best_of_legs_env_steps_per_s": max(one, value, legs["physical_cores"]["value"] or 0.0, all_value),
        "threads_pinned_first_touch_parallel": bool(pinned),
        "cgroup_cpu_quota_cpus": quota,
        "host_cpu": cpu_model,
        "host_logical_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"{reps} steps x {Bs} envs of the same workload, OpenMP over a fixed {cores} host threads "
                  f"({dt:.1f} s of CPU work)",
        "step_vs_tight_max_rel_err": worst,
        "step_vs_tight_max_err_in_reference_tolerances": worst_cls,  # |diff| / (1e-6 |x| + 1e-8): <= O(1) = the reference's class
        "reference_shaped": {
            "value": ref_shaped,
            "unit": "env-steps/s",
            "cores": 1,
            "sample": f"{n_steps} steps: README quick-start cstr (99-step episodes), ONE env per Python call into the C "
                      "oracle, adaptive 5(4) pair at rtol 1e-6 / atol 1e-8 (the reference's CVODES tolerance class)",
            "context": "the reference's own make_env.step behind stubs + SciPy measured 207 us/step = 4.8e3 env-steps/s "
                       "on one 2.1 GHz Xeon vCPU of the build container (BASELINE.md section 3.2); it cannot run on the "
                       "GPU box (pure Python over casadi / jax wheels that are not installed)",
        },
    }


# Normalised actions ~ U(-1,1) over the full action box, except four_tank: pump voltages in the upper 3/4 of the box
# (U(-0.5,1)) -- with the full box ~0.05 % of the envs drain tank 3 and sqrt(2 g h) of a negative level is NaN, in the
# reference just the same (model_classes.py:891-913)
def act_box(spec):
    return 0.75 if spec.model.name == "four_tank" else 1.0


def act_shift(spec):
    return 0.25 if spec.model.name == "four_tank" else 0.0


def x0_box(spec):
    """[low, high] of the initial states the reset kernel draws (x0 (1 +- pct), uniform), or the single x0"""
    import numpy as np

    x0 = np.asarray(spec.x0[:spec.nx], dtype=float)
    if spec.x0_unc is None:
        return [x0.tolist(), x0.tolist()]
    pct = np.asarray(spec.x0_unc, dtype=float)
    return [(x0 * (1 - pct)).round(6).tolist(), (x0 * (1 + pct)).round(6).tolist()]


_PMC = None


def committed_pmc(workload, build_id):
    """per-launch counters of this workload's dominant kernel from the committed rocprofv3 PMC passes (profiles/r6/pmc.json,
    made by tools/prof_all.sh + tools/pmc_json.py on the GPU box; FETCH_SIZE x 2 per MI355X_MICROARCH.md's gfx950 note).
    NOT measured in this run: hardware counters cannot be read from inside the process.  An entry is quoted only when it was
    taken on THIS build of the library (pcg_build_id(): a digest of the kernel headers and the .hip units): -> (entry or None, reason)."""
    global _PMC
    if _PMC is None:
        _PMC = {}
        for tp in ("profiles/r6/pmc.json", "profiles/r5/pmc.json", "profiles/r4/pmc.json", "profiles/r3/pmc.json"):
            tpath = os.path.join(ROOT, tp)
            if os.path.exists(tpath):
                with open(tpath) as fh:
                    _PMC = json.load(fh)
                _PMC["__file__"] = tp
                break
    e = _PMC.get(workload)
    if not e:
        return None, "no committed counter pass for this workload"
    if e.get("build_id") != build_id:
        return None, (f"stale profile: {_PMC.get('__file__')} was taken on build {e.get('build_id', 'unrecorded')}, "
                      f"this library is {build_id}")
    return e, None


COPY_CEILING = {"GBps": None}  # filled by clock_preheat()


def _pin_to_gpu_numa(torch, dev_index):
    """N-rank runs: keep a rank's host threads on the NUMA node of its GPU (its launch loop and the runtime's helper threads
    otherwise migrate across sockets).  Returns what was done, for the JSON line; never fails the run."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as fh:
            node = int(fh.read())
        if node < 0:
            return {"numa_node": None, "pinned": False, "why": "the device reports no NUMA node"}
        with open(f"/sys/devices/system/node/node{node}/cpulist") as fh:
            cpus = set()
            for part in fh.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= set(os.sched_getaffinity(0))
        if not cpus:
            return {"numa_node": node, "pinned": False, "why": "no allowed CPU on that node"}
        os.sched_setaffinity(0, cpus)
        return {"numa_node": node, "pinned": True, "cpus": len(cpus)}
    except Exception as e:  # noqa: BLE001 -- measurement hygiene, not a requirement
        return {"numa_node": None, "pinned": False, "why": f"{type(e).__name__}: {e}"}


def clock_preheat(torch, dev, ms):
    """Untimed, workload-independent GPU busy loop: an idle MI355X needs ~50 ms of work to reach steady clocks.
    Plain torch operations, NOT the bench's own launches, so that `warmup` is exactly the warm-up that ran: a matmul (core
    clock) and a 256 MB streaming copy (memory / fabric clocks: the headline kernel is HBM-bound) alternate."""
    if ms <= 0:
        return 0.0
    import os

    mode = os.environ.get("PCG_BENCH_PREHEAT", "both")  # experiment switch: matmul | copy | both
    a = torch.randn((2048, 2048), device=dev, dtype=torch.float32)
    src = torch.empty(32 << 20, device=dev, dtype=torch.float64)
    dst = torch.empty_like(src)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            if mode != "copy":
                a = torch.tanh(a @ a * 1e-3)
            if mode != "matmul":
                dst.copy_(src)
        torch.cuda.synchronize()
    # the copy it has just been running, timed: the measured on-box ceiling of a streaming kernel (BASELINE.md section 3 asks
    # for the fraction of a measured copy ceiling beside the 8 TB/s vendor peak): 256 MiB read + 256 MiB written per copy
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    COPY_CEILING["GBps"] = 20 * 2.0 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return (time.perf_counter() - t0) * 1e3


def self_launch(n):
    """`python3 bench.py --gpus N` without a launcher: re-run this command line as N ranks of ONE node under
    torch.distributed.run (the command the driver's contract names).  Returns the launcher's exit code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="cstr", choices=["cstr", "cstr_safe", "cstr_rollout", "cstr_safe_rollout", "cstr_unc", "four_tank", "me10", "me10_ros4", "me10_ros5", "me20", "cryst", "cryst_cv8", "mixed"])
    ap.add_argument("--batch", type=int, default=None, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-legs", action="store_true",
                    help="cpu_baseline: also run the legs whose thread count is above the container's cgroup CPU quota")
    ap.add_argument("--cpu-threads", type=int, default=None, help="fixed OpenMP team of the cpu_baseline leg (default min(16, avail))")
    ap.add_argument("--preheat-ms", type=float, default=100.0,
                    help="untimed GPU clock pre-heat (generic matmul loop) before the warm-up steps (0 = off)")
    ap.add_argument("--separate-reset", action="store_true",
                    help="end each episode with a separate pcg_reset launch instead of the fused pcg_step_autoreset (A/B)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the consecutive plain steps of every episode as one HIP graph (pcg_graph_*) instead of eager "
                         "launches -- also partial episodes, e.g. --steps 20 (the mixed workload does by default: one graph per "
                         "segment and episode)")
    ap.add_argument("--work-queue", action="store_true",
                    help="single-model adaptive workloads: route the plan through the in-workgroup work queue whatever the "
                         "model (PCG_OPT_VARIANT 5), e.g. --workload cstr_safe --integrator dopri5 --work-queue")
    ap.add_argument("--eager", action="store_true",
                    help="plain pcg_step launches instead of the HIP graphs (the default since round 6 for every workload: the "
                         "consecutive plain steps of an episode are one graph launch, recorded before the timed region)")
    ap.add_argument("--substeps", type=int, default=None,
                    help="cstr workload: RK4 sub-steps per env step (1 = the headline; other values are probes)")
    ap.add_argument("--status", type=int, default=1, help="write the per-env status byte (0 = off, A/B)")
    ap.add_argument("--integrator", default=None, choices=["dopri5", "rodas4", "rodas5", "rodas3", "rk4", "cv8", "rk4g", "tsit5g"],
                    help="me10 / me20 / mixed: integrator of the extraction envs; four_tank / cstr_safe: the plan "
                         "(default: the workload's named one)")
    ap.add_argument("--coop-thr", type=float, default=None,
                    help="me10_ros4 / mixed: threshold of the cooperative rule of the Rodas4 plan (0 = off; default: the plan's)")
    args = ap.parse_args()
    # Round 6: HIP graphs are the default launch form of the single-model workloads too (the mixed shard's since round 4):
    # same kernels, same arguments, no launch-to-launch gaps -- measured on the headline 13.58 -> 12.50 us per step in the default
    # shape and 15.2 -> 14.5 in a 20-step region (profiles/r6/graph_default.txt).  --eager gives the plain launches.
    if not args.eager and args.workload != "mixed":
        args.graph = True

    import numpy as np
    import torch

    from pcgym_amd import VecEnv, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed as `python3 bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on
        # this node (rendezvous on 127.0.0.1, a free port), same argv; rank 0 of the children prints the JSON line
        raise SystemExit(self_launch(args.gpus))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or without torch.distributed.run: bench.py starts its own ranks)")
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU fallback"
    # PCG_BENCH_BACKEND=gloo is an explicit TEST switch: it lets the N-rank code path run on a box with fewer GPUs than
    # ranks (ranks share devices).  The default is RCCL with one rank per GPU; if RCCL cannot initialise the run FAILS
    # (no silent fallback), and the JSON line records the backend and the rank count the communicator reports.
    backend = os.environ.get("PCG_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()  # == local_rank on a node with one GPU per rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pin = _pin_to_gpu_numa(torch, dev_index) if world > 1 else {"pinned": False, "why": "single rank"}
    dist, ranks_seen = None, 1
    # PCG_BENCH_FORCE_DIST=1 (TEST switch, under torch.distributed.run with one rank): take the communicator path even with
    # one rank, so that the RCCL branch -- device-bound process group, device tensors in the reductions, barrier +
    # synchronize brackets -- runs on a one-GPU box (RCCL refuses two ranks on one device)
    if world > 1 or (os.environ.get("PCG_BENCH_FORCE_DIST") and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" == RCCL on ROCm
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)  # communicator creation is lazy: fail here, not inside the timed region
            torch.cuda.synchronize()
            ranks_seen = int(round(float(probe.item())))  # every rank contributed 1: what the communicator spans
        else:
            dist.init_process_group(backend=backend)
            ranks_seen = dist.get_world_size()
        if ranks_seen != world:
            raise SystemExit(f"communicator spans {ranks_seen} ranks, expected {world}")

    lib = _lib.load()
    build_id = lib.pcg_build_id().decode()
    stream = torch.cuda.current_stream(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    mixed = args.workload == "mixed"

    if mixed:
        B = args.batch or (1 << 20)
        from pcgym_amd import make_mixed_sharded_env

        # global layout [cstr x world | four_tank x world | ME x world]; every rank owns the same slice of every segment
        segs_global = [(params, n * world) for params, n in mixed_segments(B)]
        if args.integrator:
            segs_global[2][0]["integrator"] = args.integrator
            if args.integrator == "dopri5":
                segs_global[2][0].update(rtol=1e-8, atol=1e-8)
        if args.coop_thr is not None:
            segs_global[2][0]["cooperative"] = {"thr": args.coop_thr} if args.coop_thr > 0 else False
        if os.environ.get("PCG_BENCH_ME_TOL"):  # measurement switch (tools/sessions): tolerance of the extraction envs' plan
            segs_global[2][0].update(rtol=float(os.environ["PCG_BENCH_ME_TOL"]), atol=float(os.environ["PCG_BENCH_ME_TOL"]))
        K = args.steps if args.steps is not None else 118
        W = args.warmup if args.warmup is not None else 12
        menv = make_mixed_sharded_env(segs_global, rank=rank, world=world, device=dev, seed=1234, auto_reset=True,
                                      track_status=bool(args.status), timing=True)
        envs = menv.envs
        B_eff = menv.B
        n_act = 4
        acts = [act_box(e.spec) * (2 * torch.rand((n_act, e.spec.na, e.B), generator=gen, device=dev, dtype=torch.float64) - 1)
                + act_shift(e.spec) for e in envs]
        menv.reset()
        torch.cuda.synchronize()
        wl_name = "mixed_cstr+four_tank+me_b2^20_gauss_fp64"
        N = envs[0].N
        assert all(e.N == N for e in envs)

        # --graph: one HIP graph per segment = a reset + a whole episode of step launches on that segment's stream
        # (pcg_graph_*); the timed region replays them -- same kernels, no launch-to-launch gaps
        graphs = None
        if not args.eager and K % (N - 1) == 0:
            graphs = [e.capture_steps([a[j % n_act] for j in range(N - 1)], with_reset=True) for e, a in zip(envs, acts)]
        graph_steps = [0]

        def run(n, timed):
            # actions are resident and nothing consumes the outputs between steps (as for the single-model workloads,
            # whose launches queue back to back on one stream): no cross-stream hand-shake per step, one join at the end
            menv.timing = timed
            if graphs is not None and n % (N - 1) == 0:
                for ep in range(n // (N - 1)):
                    for i, (g, st) in enumerate(zip(graphs, menv.streams)):
                        with torch.cuda.stream(st):
                            if timed:
                                eb, ee = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                                eb.record(st)
                            g.replay()
                            if timed:
                                ee.record(st)
                                menv._events[i].append((eb, ee))
                if timed:
                    graph_steps[0] = N - 1
                menv.join()
                return
            for i in range(n):
                menv.step([a[i % n_act] for a in acts], join=False)
            menv.join()

        spec = envs[2].spec
    else:
        wl_name, params, Bd, (Kd, Wd), n_act = single_workload(args.workload)
        if os.environ.get("PCG_BENCH_NACT"):  # measurement switch (tools/sessions): number of rotating action slabs
            n_act = max(1, int(os.environ["PCG_BENCH_NACT"]))
        if args.integrator and args.workload in ("me10", "me20"):
            params["integrator"] = args.integrator
            for k in ("rtol", "atol"):  # the integrator's own default tolerance for this model (config.ROS4_TOL)
                params.pop(k, None)
            wl_name = wl_name.replace("dopri5_1e-8", args.integrator)
        if args.integrator and args.workload in ("four_tank", "cstr_safe"):
            params["integrator"] = args.integrator
            wl_name = wl_name.replace("cv8x1", args.integrator).replace("default-plan(tsit5g)", "plan(" + args.integrator + ")")
        if args.coop_thr is not None and params.get("integrator") in ("rodas4", "rodas5"):
            params["cooperative"] = {"thr": args.coop_thr} if args.coop_thr > 0 else False
            wl_name += f"+coop{args.coop_thr:g}"
        if os.environ.get("PCG_BENCH_ME_TOL") and args.workload in ("me10", "me10_ros4", "me10_ros5", "me20"):  # measurement switch
            params.update(rtol=float(os.environ["PCG_BENCH_ME_TOL"]), atol=float(os.environ["PCG_BENCH_ME_TOL"]))
            wl_name += "+tol" + os.environ["PCG_BENCH_ME_TOL"]
        B = args.batch or Bd
        K = args.steps if args.steps is not None else Kd
        W = args.warmup if args.warmup is not None else Wd
        if args.substeps is not None:
            params["substeps"] = args.substeps
        # shard: rank r owns global envs [r*B, (r+1)*B)  (weak scaling; RNG streams keyed by global index)
        env = VecEnv(params, n_envs=B, device=dev, seed=1234, auto_reset=True, env_offset=rank * B,
                     track_status=bool(args.status), variant=(5 if args.work_queue else None))
        if args.work_queue:
            wl_name += "+work-queue"
        B_eff = B
        spec = env.spec
        acts = act_box(spec) * (2 * torch.rand((n_act, spec.na, B), generator=gen, device=dev, dtype=torch.float64) - 1) \
            + act_shift(spec)
        env.reset()
        if args.workload in ("cryst", "cryst_cv8"):  # initial moments x (1 + 0.01 U), CV and Ln recomputed (cryst_train.py:80-81)
            x = env.x.clone()
            x[:5] *= 1 + 0.01 * (2 * torch.rand((5, B), generator=gen, device=dev, dtype=torch.float64) - 1)
            x[5] = torch.sqrt(x[2] * x[0] / x[1] ** 2 - 1)
            x[6] = x[1] / x[0]
            env.x.copy_(x)
        torch.cuda.synchronize()
        plan, bufp, buf, sptr = env._plan, env._bufp, env._buf, stream.cuda_stream
        last_t = env.N - 1
        graph = None
        graphs = {}  # --graph: (first step, steps) -> the recorded launches (pcg_graph_*), created before the timed region

        def graph_for(t0, m):
            if (t0, m) not in graphs:
                t_now, env.t = env.t, t0
                try:
                    graphs[(t0, m)] = env.capture_steps([acts[(t0 + j) % n_act] for j in range(m)])
                finally:
                    env.t = t_now
            return graphs[(t0, m)]

        def chunks(t, n):
            """the runs of consecutive steps of one episode that n steps from step t fall into: (first step, steps)"""
            out, i = [], 0
            while i < n:
                m = min(last_t - t, n - i)
                out.append((t, m))
                i += m
                t = (t + m) % last_t
            return out

        def graph_len(t, m):
            """steps of the chunk (t, m) replayed as a graph: all but the episode's last one, whose launch carries the reset"""
            return (m - 1 if t + m == last_t and not args.separate_reset else m) if args.graph else 0
        brackets = []
        stepsum = [0.0, 0.0, 0]  # accepted, rejected, samples (adaptive workloads)

        # Everything the launch loop needs is looked up ONCE, outside the timed region: the action slabs' device addresses
        # (indexing a tensor costs the host 3-4 us per step -- a quarter of this kernel's run time, visible as idle GPU at
        # the start of a 20-step region), the entry points, a pool of event pairs.
        a_ptrs = [acts[i].data_ptr() for i in range(n_act)]
        step_fn, step_ar_fn = lib.pcg_step, lib.pcg_step_autoreset
        n_pairs = 2 + (max(Kd, K) + last_t - 1) // last_t
        ev_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_pairs)]

        roll = args.workload in ("cstr_rollout", "cstr_safe_rollout")
        if roll:
            # a launch is a whole episode: steps and warm-up are rounded UP to whole episodes (the line reports what ran)
            K = max(1, -(-K // last_t)) * last_t
            W = -(-W // last_t) * last_t
            Nn = env.N
            # the collector's storage (rollout.collect_rollouts): x (Nobs, N, B), r (1, N, B), actions (N, na, B)
            traj_x = torch.empty((spec.nobs, Nn, B), dtype=torch.float64, device=dev)
            traj_r = torch.zeros((1, Nn, B), dtype=torch.float64, device=dev)
            a_seq = acts[:Nn].contiguous()
            roll_fn = lib.pcg_rollout_strided
            x1_ptr, r1_ptr, a_ptr = traj_x[:, 1:].data_ptr(), traj_r[:, 1:].data_ptr(), a_seq.data_ptr()

        def run_rollout(n, timed):
            for _ in range(n // last_t):
                env.reset()  # inside the timed region, like the headline's episode-end resets
                traj_x[:, 0].copy_(env.obs_soa)
                if timed:
                    eb, ee = ev_pool[len(brackets)]
                    eb.record(stream)
                rc = roll_fn(plan, bufp, 0, last_t, a_ptr, spec.na * B, B, x1_ptr, B, Nn * B, r1_ptr, B,
                             env._episode_seed(), sptr)
                if rc:
                    _lib.check(rc, "pcg_rollout_strided")
                if timed:
                    ee.record(stream)
                    brackets.append((eb, ee, last_t))
                env.t = last_t

        def run(n, timed):
            if roll:
                return run_rollout(n, timed)
            # hipEvent pairs on the launch stream, one pair around each run of consecutive step launches of an episode
            i = 0
            while i < n:
                t = env.t
                m = min(last_t - t, n - i)  # steps left in this episode
                if timed:
                    eb, ee = ev_pool[len(brackets)]
                    eb.record(stream)
                mg = graph_len(t, m)
                if mg >= 2:  # the chunk's plain steps as ONE graph launch (same kernels, no launch-to-launch gaps)
                    graph_for(t, mg).replay()
                    t += mg
                else:
                    mg = 0
                if mg < m:
                    seed = env._episode_seed()
                    for j in range(m - mg):
                        buf.a = a_ptrs[t % n_act]
                        if t == last_t - 1 and not args.separate_reset:
                            # last step of the episode: the reset of the (lock-stepped) batch happens inside the same
                            # launch (pcg_step_autoreset), with the next episode's RNG key
                            env.episode += 1
                            rc = step_ar_fn(plan, bufp, t, seed, env._episode_seed(), sptr)
                            t = -1
                        else:
                            rc = step_fn(plan, bufp, t, seed, sptr)
                        if rc:
                            _lib.check(rc, "pcg_step")
                        t += 1
                    env.t = t
                if timed:
                    ee.record(stream)
                    brackets.append((eb, ee, m))
                i += m
                if env.t == last_t:  # after a graph replay or with --separate-reset
                    env.reset()

    # what each rank owns, as the communicator sees it: the global index of every rank's first env (the RNG key of an env is
    # its global index: shards must not overlap) -- one all-reduce outside the timed region
    rank_offsets = [0]
    if dist is not None:
        first = (envs[0].env_offset if mixed else env.env_offset)
        offs = torch.zeros(world, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        offs[rank] = float(first)
        dist.all_reduce(offs)
        rank_offsets = [int(v) for v in offs.tolist()]
    if not mixed and args.graph and not roll:
        # every graph the warm-up and the timed region will replay is recorded and instantiated here, outside both
        for t0, m in chunks(env.t, W) + chunks((env.t + W) % last_t, K):
            if graph_len(t0, m) >= 2:
                graph_for(t0, graph_len(t0, m))
    preheat_ms = clock_preheat(torch, dev, args.preheat_ms)
    run(W, False)
    # the timed region: K steps between (synchronize, barrier, synchronize) brackets.  With ONE rank there is no barrier and
    # the second synchronize of each bracket has nothing to wait for -- but still costs its ~15 us round trip through the
    # runtime (tools/region_probe.py), 5 % of a 20-step region of this kernel: it is only issued when there is a barrier
    # whose (RCCL) work it has to drain.
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(K, True)
    t_host = time.perf_counter() - t0  # the launch loop alone (asynchronous launches: the host's cost per step while it is ahead)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    host_us = t_host / max(K, 1) * 1e6
    # this rank's own figures, before the maximum over ranks replaces them: wall time per step, mean in-run kernel time (HIP
    # events on the launch stream; the dominant segment's for the mixed shard), host cost of the launch loop per step, NUMA
    # node -- printed per rank so that a bad 1 -> N curve can be attributed (which rank; host loop or kernel)
    if mixed:
        kern_local_us = max(tot_ms * 1e3 / max(n * max(graph_steps[0], 1), 1) for tot_ms, n in menv.segment_times())
    else:
        kern_local_us = sum(eb.elapsed_time(ee) for eb, ee, _ in brackets) * 1e3 / max(sum(m for _, _, m in brackets), 1)
    per_rank = {"ms_per_step": [elapsed / K * 1e3], "kernel_avg_us": [kern_local_us], "host_launch_loop_us_per_step": [host_us],
                "numa_node": [pin.get("numa_node")], "pinned_to_gpu_numa_node": [bool(pin.get("pinned"))]}
    if dist is not None:
        pr = torch.zeros((world, 5), device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        node = pin.get("numa_node")
        pr[rank] = torch.tensor([elapsed / K * 1e3, kern_local_us, host_us, -1.0 if node is None else float(node),
                                 1.0 if pin.get("pinned") else 0.0], dtype=torch.float64)
        dist.all_reduce(pr)
        prl = pr.cpu().tolist()
        per_rank = {"ms_per_step": [r[0] for r in prl], "kernel_avg_us": [r[1] for r in prl],
                    "host_launch_loop_us_per_step": [r[2] for r in prl],
                    "numa_node": [None if r[3] < 0 else int(r[3]) for r in prl],
                    "pinned_to_gpu_numa_node": [bool(r[4]) for r in prl]}
        tt = torch.tensor([elapsed, host_us], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, host_us = float(tt[0].item()), float(tt[1].item())

    # sanity: a fast kernel producing garbage is not a result.  Physical boxes, not just finiteness (a diverged RK4
    # yields finite values up to 1e91), and the per-env status byte of the last step.
    def sane(e):
        ok = bool(torch.isfinite(e.x).all().item() and torch.isfinite(e.rew).all().item())
        if e.status is not None:
            ok = ok and not bool(e.status.any().item())
        name = e.spec.model.name
        if name == "cstr":
            ok = ok and bool(((e.x[0] >= 0) & (e.x[0] <= 2) & (e.x[1] >= 250) & (e.x[1] <= 600)).all().item())
        if name.startswith("multistage"):
            ok = ok and bool(((e.x >= -1e-9) & (e.x <= 3.0)).all().item())
        if name == "four_tank":
            ok = ok and bool((e.x < 5.0).all().item())
        return ok

    if not mixed and env.nsteps is not None:  # adaptive: accepted / rejected counts of the last timed step (outside the timing)
        ns = env.nsteps.to(torch.float64).mean(dim=1)
        stepsum[:] = [float(ns[0].item()), float(ns[1].item()), 1]
    all_envs = envs if mixed else [env]
    finite = all(sane(e) for e in all_envs)
    total_env_steps = float(B_eff) * K * world
    value = total_env_steps / elapsed

    out = {
        "metric": METRIC,
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": wl_name,
            "envs_per_gpu": B_eff,
            "global_envs": B_eff * world,
            "episode_len": spec.N - 1,
            "dt_model_units": spec.dt,
            "x0_box": x0_box(spec),
            "parallelism": f"env-shard x{world} (no collective on the hot path)",
            "collective_backend": ("rccl" if backend == "nccl" else backend) if dist is not None else "none (single process)",
            "ranks_seen": ranks_seen,
            "clock_preheat_ms": round(preheat_ms, 1),
            "status_byte": bool(args.status),
            "library_build_id": build_id,
            "rank_first_env": rank_offsets,
            "host_launch_loop_us_per_step_max_over_ranks": round(host_us, 2),
            "per_rank": per_rank,
            "sane": finite,
        },
    }
    if rank == 0:
        if mixed:
            segs_out = []
            for e, (tot_ms, n) in zip(envs, menv.segment_times()):
                kern_s = tot_ms * 1e-3 / max(n * max(graph_steps[0], 1), 1)  # (a graph bracket holds a whole episode)
                alg = float(e.bytes_per_env_step) * e.B
                d = {"segment": e.spec.model.name, "envs": e.B, "integrator": e.spec.integrator,
                     "kernel_avg_us": kern_s * 1e6, "algorithmic_bytes_per_env_step": int(e.bytes_per_env_step),
                     "hbm_GBps": alg / kern_s / 1e9, "hbm_frac": alg / kern_s / 1e9 / HBM_PEAK_GBS}
                if e.nsteps is not None:
                    ns = e.nsteps.to(torch.float64).mean(dim=1)
                    att = float(ns.sum().item())
                    fl = flops_per_env_step(e.spec.model.name, e.spec.nx, e.spec.integrator, att) * e.B
                    d.update(attempted_steps_mean=att, fp64_TFLOPs=fl / kern_s / 1e12,
                             fp64_frac=fl / kern_s / 1e12 / FP64_PEAK_TFLOPS)
                segs_out.append(d)
            out["config"]["launch"] = ("eager pcg_step launches, one stream per segment" if graphs is None else
                                       "one HIP graph per segment and episode (reset + 59 steps, pcg_graph_*), one stream per segment")
            pmx, pm_why = committed_pmc("mixed", build_id) if B == (1 << 20) and not args.integrator else (None, "non-default run")
            if pmx:
                for d in segs_out:
                    q = pmx.get("segments", {}).get(d["segment"])
                    if q:
                        d["traffic"] = q["traffic_bytes_per_launch"]
                        d["traffic_over_algorithmic"] = q["traffic_bytes_per_launch"] / (float(d["algorithmic_bytes_per_env_step"]) * d["envs"])
                        if "valu_issue_frac" in q:
                            d["valu_issue_frac"] = q["valu_issue_frac"]
                        if "valu_issue_time_frac_by_class" in q:  # the segment's launch priced by instruction class (tools/pmc_json.py)
                            d["valu_issue_time_frac_by_class"] = q["valu_issue_time_frac_by_class"]
            dom = max(segs_out, key=lambda d: d["kernel_avg_us"])
            out["roofline"] = {
                "bound": "fp64_valu" if "fp64_frac" in dom else "hbm",
                "achieved": dom.get("fp64_TFLOPs", dom["hbm_GBps"]),
                "peak": FP64_PEAK_TFLOPS if "fp64_frac" in dom else HBM_PEAK_GBS,
                "unit": "TFLOP/s" if "fp64_frac" in dom else "GB/s",
                "frac": dom.get("fp64_frac", dom["hbm_frac"]),
                "traffic": dom.get("traffic"),
                "traffic_over_algorithmic": dom.get("traffic_over_algorithmic"),
                "valu_issue_time_frac_by_class": dom.get("valu_issue_time_frac_by_class"),
                "traffic_measured_in_run": False,
                "copy_ceiling_GBps": COPY_CEILING["GBps"],
                **({} if pmx else {"traffic_reason": pm_why}),
                "kernel": f"dominant segment: {dom['segment']} ({dom['integrator']}); the three segments run concurrently "
                          "on their own streams",
                "kernel_avg_us": dom["kernel_avg_us"],
                "segments": segs_out,
            }
        else:
            # launch-weighted mean over all brackets of the timed region
            kern_avg_s = sum(eb.elapsed_time(ee) for eb, ee, _ in brackets) * 1e-3 / sum(m for _, _, m in brackets)
            bpe = env.bytes_per_env_step  # SURVEY.md section 8d formula for this plan and buffer set
            alg_bytes = float(bpe) * B
            if roll:
                # the fused rollout keeps the state in registers: per env step it reads the action and writes the
                # observation row and the reward, 8 (na + Nobs + 1) B; the state is read and the last step's outputs written
                # once per episode.  `kern_avg_s` is the launch's duration / 59 (one bracket per launch), so `achieved` is per
                # env-step batch, like every other workload's
                bpe = 8 * (spec.na + spec.nobs + 1)
                alg_bytes = float(bpe) * B + float(8 * (2 * spec.nx + spec.nobs + 1) + 2) * B / last_t
            achieved = alg_bytes / kern_avg_s / 1e9
            adaptive = spec.integrator not in ("rk4", "cv8")
            fp64 = spec.model.name in FLOP_PER_RHS
            rl = {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "traffic_measured_in_run": False,
                "kernel_avg_us": kern_avg_s * 1e6,
                "algorithmic_bytes_per_env_step": int(bpe),
                "algorithmic_bytes_per_launch": alg_bytes,
                # the 256 MiB device copy clock_preheat() ran just before the warm-up, timed: the measured on-box ceiling
                "copy_ceiling_GBps": COPY_CEILING["GBps"],
                "frac_of_copy_ceiling": (achieved / COPY_CEILING["GBps"]) if COPY_CEILING["GBps"] else None,
            }
            if adaptive and not fp64 and env.nsteps is not None and not roll:
                # adaptive plans of the cheap models (the cstr's guarded default on the full x0 box): neither HBM nor issue
                # bound -- the launch waits for its heaviest env, one lane crossing an ignition front attempt by attempt
                att_max = int(env.nsteps.sum(dim=0).max().item())
                rl.update(bound="chain", chain={"heaviest_env_attempts_last_step": att_max,
                                                "note": "launch time ~ guarded step + heaviest env's attempts x 1.6-2.1 us of a "
                                                        "latency-bound lane (DESIGN.md section 3); the hbm figures beside it are the "
                                                        "algorithmic bytes over that time, not what limits it"})
            if fp64:
                # fp64-issue-bound kernels: algorithmic flops = RHS evaluations x (flop per RHS + RK stage combination)
                att = 0.0
                if adaptive:
                    att = (stepsum[0] + stepsum[1]) / max(stepsum[2], 1)
                    rhs = (2 + 6 * att) if spec.integrator == "dopri5" else 6 * att
                    rl["attempted_steps_mean"] = att
                    rl["accepted_steps_mean"] = stepsum[0] / max(stepsum[2], 1)
                else:
                    rhs = (11 if spec.integrator == "cv8" else 4) * spec.substeps
                fl = flops_per_env_step(spec.model.name, spec.nx, spec.integrator, att, spec.substeps) * B
                tf = fl / kern_avg_s / 1e12
                rl.update(bound="fp64_valu", achieved=tf, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                          frac=tf / FP64_PEAK_TFLOPS, hbm_GBps=achieved, algorithmic_flops_per_launch=fl,
                          rhs_evals_per_env_step=rhs)
            # HBM traffic per launch comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read from inside
            # this process): the committed measurement of this kernel + workload, if present -- NOT measured in this run
            pm, pm_why = (committed_pmc(args.workload, build_id) if (B == Bd and args.substeps is None and not args.integrator)
                          else (None, "non-default run"))
            if not pm:
                rl["traffic_reason"] = pm_why
            if pm:
                rl["traffic"] = pm["traffic_bytes_per_launch"]
                rl["traffic_over_algorithmic"] = pm["traffic_bytes_per_launch"] / (alg_bytes * (last_t if roll else 1))
                rl["traffic_source"] = pm["source"]
                if "valu_issue_frac" in pm:
                    # instruction-issue view of the same kernel: 4 cycles per wave64 VALU instruction on a 16-lane SIMD,
                    # 1024 SIMDs, cycles = GRBM_GUI_ACTIVE of the dispatch -- independent of any flop weighting
                    rl["valu_issue_frac"] = pm["valu_issue_frac"]
                    rl["valu_insts_per_launch"] = pm["SQ_INSTS_VALU_per_launch"]
                    if "valu_issue_time_frac_by_class" in pm:
                        # the same launch priced by instruction class at the measured issue costs (tools/issuebench.hip: fp64
                        # add / mul / fma 2.4 ns, fp64 estimates 7.3 ns, fp32 transcendentals 3.65 ns, the rest 1.35 ns per
                        # wave-instruction and SIMD): the share of the launch its SIMDs spend issuing vector work
                        rl["valu_issue_time_frac_by_class"] = pm["valu_issue_time_frac_by_class"]
                        rl["valu_insts_by_class_per_launch"] = pm["valu_insts_by_class_per_launch"]
                    elif "valu_issue_time_frac_at_2p4ns" in pm:
                        rl["valu_issue_time_frac_at_2p4ns"] = pm["valu_issue_time_frac_at_2p4ns"]
            if fp64:
                f_survey = {"crystallization": 80 + 2 + 3 + 1 + 6}.get(spec.model.name)  # "~80 flop + 2 exp + 3 pow + 1 sqrt + ~6 div" counted as one flop each
                rl["flop_weighting"] = ("flop per RHS evaluation = SURVEY.md section 8(a)'s count" if f_survey is None else
                                        "crystallization: 232 flop per RHS = 60 plain + 4 exp/pow at 25 + sqrt 12 + 6 divides "
                                        "at 10 (instruction-level cost of the transcendentals); SURVEY.md section 8(a) counts "
                                        f"every operation once: {f_survey} flop -> frac_at_survey_flop_count")
                if f_survey is not None:
                    fs = (spec.substeps * (11 * f_survey + 120 * spec.nx) if spec.integrator == "cv8" else
                          4 * spec.substeps * (f_survey + 12 * spec.nx))
                    rl["frac_at_survey_flop_count"] = rl["frac"] * (fs * B) / fl
            out["roofline"] = rl
            if roll and adaptive:
                rl["note"] = ("barrier-free rollout of a guarded plan (pcg_rollout_flat.hpp): two launches per episode; neither "
                              "HBM- nor issue-bound end to end -- the second pass carries one env per lane through the adaptive "
                              "pair; the hbm figures are the algorithmic bytes over the episode's time")
            if roll:
                rl["kernel_launch_us"] = kern_avg_s * 1e6 * last_t
                rl["env_steps_per_launch"] = last_t
                rl["algorithmic_bytes_per_launch"] = alg_bytes * last_t
            out["config"]["launch"] = (f"one pcg_rollout_strided launch per {last_t}-step episode (+ reset kernel, + copy of the "
                                       "first observation row)" if roll else "eager pcg_step launches" if not graphs else
                                       f"HIP graphs (pcg_graph_*) of the consecutive plain steps of an episode: {len(graphs)} recorded before "
                                       "the timed region; an episode's last step (same-launch reset) is a plain launch")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, threads=args.cpu_threads, all_legs=args.cpu_all_legs)
        print(json.dumps(out), flush=True)
    for e in all_envs:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
