"""Gymnasium-shaped façade over the HIP engine.

``make_env(env_params)`` keeps the reference's single-environment surface
(src/pcgym/pcgym.py:31-500: ``reset() -> (obs, info)``,
``step(a) -> (obs, float, bool, False, info)``, the attributes its callers read --
policy_evaluation.py:86-128 -- and the ``env_params`` dict verbatim).
``make_vec_env(env_params, n_envs)`` / ``VecEnv`` is the batched form: B
environments stepped by one kernel launch, state held in torch tensors on the
GPU in SoA layout ``field[component][B]``; observations are returned as the
``(B, Nobs)`` strided view of that storage (no copy).

There is no CPU path: constructing an env without the HIP library / a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _abi as abi
from . import _lib
from .config import EnvSpec
from .spaces import Box


def _torch():
    import torch

    return torch


class VecEnv:
    """B environments of one configuration on one GPU.

    Parameters
    ----------
    env_params : dict        the reference's configuration dict (+ optional new keys, config.py)
    n_envs : int             batch size B
    device : torch.device | int | None   GPU holding the state (default: current CUDA device)
    seed : int               base key of the counter-based RNG (noise / Gaussian disturbances /
                             reset uncertainty); episode k uses seed + k
    per_env_t : bool         every env carries its own step counter (needed for masked
                             auto-reset when episodes can end at different times)
    auto_reset : bool        reset finished envs inside step() (gymnasium "same-step" mode)
    env_offset : int         global index of env 0 (multi-GPU sharding: keeps RNG streams disjoint)
    track_status : bool      keep a per-env health byte (``env.status``: 0 ok, 1 DOPRI5 step budget exhausted, 2 step-size
                             underflow, 3 non-finite state -- include/pcgym_hip.h PCG_ST_*).  Sticky: only failures
                             are written, reset() (or ``env.status.zero_()``) clears; an env whose adaptive
                             integration fails gets a NaN state either way
    """

    def __init__(self, env_params, n_envs=1, device=None, seed=0, per_env_t=False, auto_reset=False,
                 env_offset=0, lds_stages=False, variant=None, track_status=True):
        self.spec = s = EnvSpec(env_params)
        self.env_params = s.env_params
        if s.custom_reward is not None and not getattr(self, "_allow_custom_reward", False):
            # a Python callable cannot run inside the batched kernel -- but one that is a function of this step alone
            # (obs, uk, violated, self.SP[..][self.t], constants) writes its own C expression when run on symbolic
            # scalars (config.trace_reward_callable), which is then compiled into the step kernel like {'expr': ...}
            from .config import trace_reward_callable

            try:
                expr = trace_reward_callable(s.custom_reward, s)
            except ValueError as e:
                raise ValueError(f"{e}.  (custom_reward callables that cannot be traced run in the single-env make_env() "
                                 "façade, which evaluates them on the host exactly like the reference, "
                                 "pcgym.py:470-471.)") from None
            env_params = dict(s.env_params, custom_reward={"expr": expr})
            self.spec = s = EnvSpec(env_params)
            self.env_params = s.env_params
        torch = _torch()
        self._lib = _lib.load()  # raises when the HIP library is missing: no fallback
        if not torch.cuda.is_available():
            raise RuntimeError("pcgym_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU implementation of the step path")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        elif isinstance(device, int):
            device = torch.device("cuda", device)
        self.device = torch.device(device)
        self.B = int(n_envs)
        self.seed0 = int(seed)
        self.episode = 0
        self.per_env_t = bool(per_env_t)
        self.auto_reset = bool(auto_reset)
        if self.auto_reset and s.done_on_constraint and not self.per_env_t:
            raise ValueError("auto_reset with done_on_cons_vio needs per_env_t=True")
        self.t = 0

        # reference attribute names (pcgym.py:105-111, 160-199)
        self.N, self.tsim, self.dt = s.N, s.tsim, s.dt
        self.Nx, self.Nx_oracle = s.nobs, s.nx
        self.Nu, self.Nd, self.Nd_model = s.nu, s.nd, s.ndm
        self.SP, self.model = s.SP, s.model
        self.constraint_active, self.n_con = s.constraint_active, s.ncon
        self.normalise_a, self.normalise_o, self.a_delta = s.normalise_a, s.normalise_o, s.a_delta
        self.observation_space_base = Box(s.o_low, s.o_high)
        if s.normalise_o:
            self.observation_space = Box(-np.ones(s.nobs), np.ones(s.nobs))
        else:
            self.observation_space = self.observation_space_base
        if s.normalise_a:
            self.action_space = Box(-np.ones(s.na_user), np.ones(s.na_user))
        else:
            self.action_space = Box(s.a_low[:s.na_user], s.a_high[:s.na_user])

        cfg, keep = s.to_cfg()
        plan = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.pcg_plan_create(C.byref(plan), C.byref(cfg)), "pcg_plan_create")
        del keep
        self._plan = plan
        self.env_offset = int(env_offset)
        _lib.check(self._lib.pcg_plan_set_env_offset(plan, self.env_offset), "pcg_plan_set_env_offset")
        if lds_stages:
            _lib.check(self._lib.pcg_plan_set_option(plan, abi.PCG_OPT_LDS_STAGES, 1), "pcg_plan_set_option")
        if variant is None:
            variant = int(os.environ.get("PCG_VARIANT", "0"))  # A/B switch for measurements
        if variant:
            _lib.check(self._lib.pcg_plan_set_option(plan, abi.PCG_OPT_VARIANT, int(variant)),
                       "pcg_plan_set_option")
        for key, opt in (("PCG_BPC", abi.PCG_OPT_STREAM_BLOCKS_PER_CU), ("PCG_NT", abi.PCG_OPT_NT_STORES)):
            if os.environ.get(key):  # measurement switches
                _lib.check(self._lib.pcg_plan_set_option(plan, opt, int(os.environ[key])), "pcg_plan_set_option")

        B, dev, f64 = self.B, self.device, torch.float64
        self.x = torch.zeros((s.nx, B), dtype=f64, device=dev)
        self.obs_soa = torch.zeros((s.nobs, B), dtype=f64, device=dev)
        self.rew = torch.zeros(B, dtype=f64, device=dev)
        self.done = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.viol = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.status = torch.zeros(B, dtype=torch.uint8, device=dev) if track_status else None
        self.a_save_t = torch.zeros((s.na, B), dtype=f64, device=dev) if s.a_delta else None
        self.g = torch.zeros((s.ncon, B), dtype=f64, device=dev) if s.ncon else None
        self.g_pre = torch.zeros((s.ncon, B), dtype=f64, device=dev) if s.ncon else None
        self.t_env = torch.zeros(B, dtype=torch.int32, device=dev) if self.per_env_t else None
        self.nsteps = (torch.zeros((2, B), dtype=torch.int32, device=dev)
                       if s.integrator not in ("rk4", "cv8") else None)
        self.p_unc = torch.zeros((s.nunc, B), dtype=f64, device=dev) if s.nunc else None  # per-env parameters
        # previous physical action of the declarative tracking reward; NaN = none yet (custom_reward.py:7-8)
        self.u_prev = (torch.full((s.na, B), float("nan"), dtype=f64, device=dev)
                       if s.reward_track is not None else None)
        b = self._buf = abi.pcg_buffers()
        b.B = B
        b.x = self.x.data_ptr()
        b.obs = self.obs_soa.data_ptr()
        b.rew = self.rew.data_ptr()
        b.done = self.done.data_ptr()
        b.viol = self.viol.data_ptr() if s.ncon else None
        b.a_save = self.a_save_t.data_ptr() if self.a_save_t is not None else None
        b.g = self.g.data_ptr() if self.g is not None else None
        b.g_pre = self.g_pre.data_ptr() if self.g_pre is not None else None
        b.t = self.t_env.data_ptr() if self.t_env is not None else None
        b.nsteps = self.nsteps.data_ptr() if self.nsteps is not None else None
        b.p_unc = self.p_unc.data_ptr() if self.p_unc is not None else None
        b.u_prev = self.u_prev.data_ptr() if self.u_prev is not None else None
        b.status = self.status.data_ptr() if self.status is not None else None
        self._bufp = C.byref(b)
        self._a_hold = None
        self._d_hold = None
        self._zero_a = None

    # ------------------------------------------------------------------
    def close(self):
        if getattr(self, "_plan", None):
            self._lib.pcg_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _stream(self):
        return _torch().cuda.current_stream(self.device).cuda_stream

    @property
    def obs(self):
        """(B, Nobs) view of the SoA observation storage."""
        return self.obs_soa.t()

    @property
    def bytes_per_env_step(self):
        return int(self._lib.pcg_plan_bytes_per_env_step(self._plan, self._bufp))

    def _episode_seed(self):
        return (self.seed0 + self.episode) & 0xFFFFFFFFFFFFFFFF

    def reset(self, seed=None, mask=None):
        """Reset all envs (or those with mask != 0).  Returns (obs (B,Nobs), info)."""
        if seed is not None:
            self.seed0 = int(seed)
        self.episode += 1
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=_torch().uint8).contiguous()
            mptr = mask.data_ptr()
        else:
            self.t = 0
            if self.status is not None:
                self.status.zero_()  # the health bytes are sticky: a full reset opens a fresh window
        _lib.check(self._lib.pcg_reset(self._plan, self._bufp, mptr, self._episode_seed(), self._stream()),
                   "pcg_reset")
        return self.obs, {}

    def bind_outputs(self, obs_soa=None, rew=None):
        """Make the kernels write observations / rewards into caller-provided storage from the next call on (zero-copy
        recording: collect_rollouts hands in one (Nobs, B) / (B,) slice of its trajectory arrays per step).  Tensors must
        be contiguous float64 on this device with the shapes of ``obs_soa`` / ``rew``; returns the previous pair."""
        torch = _torch()
        prev = (self.obs_soa, self.rew)
        for t, ref in ((obs_soa, self.obs_soa), (rew, self.rew)):
            if t is not None and (t.shape != ref.shape or t.dtype != torch.float64 or t.device != ref.device
                                  or not t.is_contiguous()):
                raise ValueError(f"bind_outputs: need a contiguous float64 tensor of shape {tuple(ref.shape)} on {ref.device}")
        # captured HIP graphs hold the OLD pointers: they refuse to replay after a re-binding (StepGraph.replay)
        self._binding_epoch = getattr(self, "_binding_epoch", 0) + 1
        if obs_soa is not None:
            self.obs_soa = obs_soa
            self._buf.obs = obs_soa.data_ptr()
        if rew is not None:
            self.rew = rew
            self._buf.rew = rew.data_ptr()
        return prev

    def _as_soa(self, v, rows, what):
        torch = _torch()
        if what == "action" and self.spec.na_user == 0:  # model without inputs: the kernels' dummy action
            if self._zero_a is None:
                self._zero_a = torch.zeros((1, self.B), dtype=torch.float64, device=self.device)
            return self._zero_a
        if not torch.is_tensor(v):
            v = torch.as_tensor(np.asarray(v, dtype=np.float64), device=self.device)
        v = v.to(device=self.device, dtype=torch.float64)
        if v.dim() == 1:
            v = v.reshape(1, -1) if rows == 1 else v.reshape(-1, 1).expand(rows, self.B)
        if v.shape == (rows, self.B):
            return v.contiguous()
        if v.shape == (self.B, rows):
            return v.t().contiguous()
        raise ValueError(f"{what} must have shape ({rows},{self.B}) [SoA, zero-copy] or ({self.B},{rows})")

    def step(self, action, disturbance=None):
        """One env step for all B envs (reference make_env.step, pcgym.py:350-500).

        action : (na,B) SoA tensor (zero-copy) or (B,na); disturbance : optional (nd,B) per-env
        values overriding the shared schedule.  Returns (obs, rew, done, truncated, info) with
        batched tensors; all device-side and asynchronous on the current stream.
        """
        s = self.spec
        a = self._as_soa(action, s.na, "action")
        self._a_hold = a  # keep alive until the next call (async launch)
        self._buf.a = a.data_ptr()
        if disturbance is not None:
            d = self._as_soa(disturbance, s.nd, "disturbance")
            self._d_hold = d
            self._buf.d = d.data_ptr()
        else:
            self._buf.d = None
        # same-launch auto-reset: every step with per-env counters (envs end at different times), only the last step
        # of the episode for a lock-stepped batch (all envs end together at t = N-2)
        fused_reset = self.auto_reset and (self.per_env_t or self.t == self.N - 2)
        if fused_reset:
            seed = self._episode_seed()
            self.episode += 1  # the resets open the next RNG epoch
            _lib.check(self._lib.pcg_step_autoreset(self._plan, self._bufp, self.t, seed, self._episode_seed(),
                                                    self._stream()), "pcg_step_autoreset")
        else:
            _lib.check(self._lib.pcg_step(self._plan, self._bufp, self.t, self._episode_seed(), self._stream()),
                       "pcg_step")
        self.t += 1
        if fused_reset and not self.per_env_t:
            self.t = 0  # the new episode started inside the launch
        info = {}
        if s.ncon:
            info["viol"] = self.viol
            info["g"] = self.g
        if self.nsteps is not None:
            info["nsteps"] = self.nsteps
        if self.status is not None:
            info["status"] = self.status
        return self.obs, self.rew, self.done.view(_torch().bool), False, info

    def rollout(self, actions, collect_obs=False, collect_rew=True):
        """Fused open-loop rollout: actions (T,na,B); state stays in registers for T steps."""
        torch = _torch()
        if self.per_env_t:
            raise ValueError("rollout() is lock-stepped only")
        s = self.spec
        actions = actions.to(device=self.device, dtype=torch.float64).contiguous()
        T = actions.shape[0]
        if actions.shape != (T, s.na, self.B):
            raise ValueError(f"actions must be (T,{s.na},{self.B})")
        obs_seq = torch.empty((T, s.nobs, self.B), dtype=torch.float64, device=self.device) if collect_obs else None
        rew_seq = torch.empty((T, self.B), dtype=torch.float64, device=self.device) if collect_rew else None
        self._buf.d = None
        _lib.check(self._lib.pcg_rollout(self._plan, self._bufp, self.t, T, actions.data_ptr(),
                                         obs_seq.data_ptr() if collect_obs else None,
                                         rew_seq.data_ptr() if collect_rew else None,
                                         self._episode_seed(), self._stream()), "pcg_rollout")
        self._a_hold = actions
        self.t += T
        return obs_seq, rew_seq

    def capture_steps(self, actions, disturbances=None, with_reset=False):
        """Record ``len(actions)`` consecutive step() launches (starting at the current ``t``, or at a
        full reset if ``with_reset``) as one HIP graph over this env's buffers.

        actions : sequence of (na,B) SoA tensors (kept alive by the returned object; entries may repeat).
        Returns a :class:`StepGraph`; ``graph.replay()`` runs the recorded steps with one host call and
        advances ``t`` -- same kernels and results as the step() loop, without the launch-to-launch gap.
        """
        if self.per_env_t:
            raise ValueError("capture_steps() is lock-stepped only")
        return StepGraph(self, actions, disturbances, with_reset)

    # state_dict for checkpoint/resume of the env batch: everything a later step() reads or a caller may look at
    # before the next step -- state, counters, RNG epoch, accumulators, the per-env uncertain parameters sampled at
    # reset, and the outputs of the last step (the policy's next input is env.obs)
    _STATE_TENSORS = ("x", "obs_soa", "rew", "done", "viol", "status", "a_save_t", "t_env", "u_prev", "p_unc", "g",
                      "g_pre", "nsteps")

    def state_dict(self):
        d = {"t": self.t, "episode": self.episode, "seed0": self.seed0}
        for k in self._STATE_TENSORS:
            v = getattr(self, k)
            if v is not None:
                d[k] = v.clone()
        return d

    def load_state_dict(self, d):
        self.t, self.episode, self.seed0 = int(d["t"]), int(d["episode"]), int(d["seed0"])
        for k in self._STATE_TENSORS:
            v = getattr(self, k)
            if v is None:
                continue
            src = d.get(k, d.get({"a_save_t": "a_save"}.get(k, k)))
            if src is None:
                if k in ("x", "p_unc", "a_save_t", "t_env", "u_prev"):
                    raise KeyError(f"state dict lacks '{k}', which this configuration needs to resume")
                continue
            v.copy_(src)


class StepGraph:
    """T recorded step() launches of one VecEnv (pcg_graph_* in include/pcgym_hip.h)."""

    def __init__(self, env, actions, disturbances=None, with_reset=False):
        s = env.spec
        self.env = env
        self.T = T = len(actions)
        self.with_reset = bool(with_reset)
        self.t0 = 0 if with_reset else env.t
        if T < 1 or self.t0 + T > env.N - 1:
            raise ValueError(f"cannot record {T} steps from t={self.t0}: an episode has {env.N - 1} steps")
        self._a = [env._as_soa(a, s.na, "action") for a in actions]
        self._d = None
        ap = (C.c_void_p * T)(*[a.data_ptr() for a in self._a])
        dp = None
        if disturbances is not None:
            if len(disturbances) != T:
                raise ValueError("one disturbance slab per recorded step")
            self._d = [env._as_soa(d, s.nd, "disturbance") for d in disturbances]
            dp = (C.c_void_p * T)(*[d.data_ptr() for d in self._d])
        self._epoch = getattr(env, "_binding_epoch", 0)
        self._uses_rng = bool(s.noise or s.gauss or (with_reset and (s.nunc or s.x0_unc is not None)))
        self._seed = (env.seed0 + env.episode + (1 if with_reset else 0)) & 0xFFFFFFFFFFFFFFFF
        g = C.c_void_p()
        with _torch().cuda.device(env.device):
            _lib.check(env._lib.pcg_graph_create(C.byref(g), env._plan, env._bufp, ap, dp, self.t0, T, self._seed,
                                                 int(self.with_reset)), "pcg_graph_create")
        self._g = g

    def replay(self):
        """Run the recorded steps on the current stream; returns (obs, rew, done) of the last one."""
        env = self.env
        if self._g is None:
            raise RuntimeError("StepGraph was destroyed")
        if getattr(env, "_binding_epoch", 0) != self._epoch:
            raise RuntimeError("the env's output buffers were re-bound (bind_outputs) after this graph was captured: the "
                               "graph would write to the old storage -- capture again")
        if self.with_reset:
            env.episode += 1
        elif env.t != self.t0:
            raise ValueError(f"graph was recorded at t={self.t0}, env is at t={env.t}")
        seed = env._episode_seed()
        if seed != self._seed and self._uses_rng:  # new episode: re-key the in-kernel RNG, no re-recording
            _lib.check(env._lib.pcg_graph_set_seed(self._g, seed), "pcg_graph_set_seed")
            self._seed = seed
        _lib.check(env._lib.pcg_graph_launch(self._g, env._stream()), "pcg_graph_launch")
        env.t = self.t0 + self.T
        return env.obs, env.rew, env.done.view(_torch().bool)

    def destroy(self):
        if getattr(self, "_g", None) is not None:
            self.env._lib.pcg_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def make_vec_env(env_params, n_envs, **kw):
    return VecEnv(env_params, n_envs=n_envs, **kw)


class make_env(VecEnv):
    """Single environment with the reference's exact call surface (numpy in/out)."""

    _allow_custom_reward = True

    def __init__(self, env_params, device=None, seed=0):
        if not isinstance(env_params, dict):
            raise ValueError("env_params must be a dictionary")
        super().__init__(env_params, n_envs=1, device=device, seed=seed)
        s = self.spec
        self.info = {}
        if s.constraint_active:
            self.info["cons_info"] = np.zeros((s.ncon, s.N, 1))
        self.custom_reward = s.custom_reward is not None
        self.custom_reward_f = s.custom_reward
        self.x0 = s.x0
        self.integration_method = s.integration_method
        self.state = s.x0_full().copy()
        self.obs_np = self.state.copy()
        self.a_save = s.a_0.copy() if s.a_delta else None
        self.a_0 = s.a_0.copy() if s.a_delta else None  # read by callers (tests/environment/test_make_env_delta_u.py:24)
        self.done_flag = False

    # -- host-side mirrors needed only for the callable custom_reward path -------------
    def _host_uk(self, action):
        s = self.spec
        a = np.asarray(action, dtype=np.float64).reshape(-1).copy()
        if s.normalise_a:
            a = (a + 1) * (s.a_high - s.a_low) / 2 + s.a_low
        if s.normalise_a and s.a_delta:
            if s.reference_compat:
                a = (a + 1) * (s.a_high - s.a_low) / 2 + s.a_low
            a = self.a_save + a
        uk = np.zeros(s.nu)
        uk[:s.na] = a
        if s.ndm:
            uk[s.na:] = s.d_default
            tn = min(self.t + 1, s.N - 1)
            for k in range(s.nd):
                uk[s.na + s.d_slot[k]] = s.d_sched[k, tn]
        return uk

    def _sync_state(self):
        s = self.spec
        self.state[:s.nx] = self.x[:, 0].cpu().numpy()
        if s.nunc:
            self.state[s.nx + s.nsp_obs + s.nd:] = self.p_unc[:, 0].cpu().numpy()

    def reset(self, seed=0, **kwargs):
        s = self.spec
        obs, _ = super().reset()
        self.state = s.x0_full().copy()
        self._sync_state()
        if s.a_delta:
            self.a_save = s.a_0.copy()
        if s.nunc:  # the reference setattr()s the sampled values onto its model object (pcgym.py:306-307, 314):
            pu = self.p_unc[:, 0].cpu().numpy()  # callers read them back as env.model.<param>
            for k, v in zip(s.unc_keys, pu):
                self.model.parameters[k] = float(v)
        o = obs[0].cpu().numpy().copy()
        self.obs_np = self.state.copy()
        self.info["obs"] = o.copy()
        self.info["r_init"] = 0
        self.done_flag = False
        return o, self.info

    def step(self, action):
        s = self.spec
        t_old = self.t
        uk = self._host_uk(action) if (self.custom_reward or s.a_delta) else None
        a = np.asarray(action, dtype=np.float64).reshape(s.na, 1)
        obs, rew, done, _, _ = super().step(_torch().as_tensor(a, device=self.device))
        o = obs[0].cpu().numpy().copy()
        self._sync_state()
        tc, tn = min(t_old, s.N - 1), min(t_old + 1, s.N - 1)
        for k in range(s.nsp_obs):
            self.state[s.nx + k] = s.sp[k, tc]
        for k in range(s.nd):
            self.state[s.nx + s.nsp_obs + k] = s.d_sched[k, tn]
        if s.a_delta:
            self.a_save = self.a_save_t[:, 0].cpu().numpy().copy()
        violated = False
        if s.ncon:
            violated = bool(self.viol[0].item())
            if t_old == 0:
                self.info["cons_info"][:, 0, 0] = self.g_pre[:, 0].cpu().numpy()
            if self.t < s.N:
                self.info["cons_info"][:, self.t, 0] = self.g[:, 0].cpu().numpy()
        r = float(rew[0].item())
        if self.custom_reward:
            # the reference hands the callable the un-normalised (noisy) observation (pcgym.py:470-471)
            if s.normalise_o:
                with np.errstate(all="ignore"):
                    obs_un = (o + 1) / 2 * (s.o_high - s.o_low) + s.o_low
            else:
                obs_un = o.copy()
            if s.obs_mask is not None:
                m = s.obs_mask == 0
                obs_un[:s.nx][m] = self.state[:s.nx][m]
            self.obs_np = obs_un
            r = self.custom_reward_f(self, obs_un, uk, violated)
        self.info["obs"] = o.copy()
        # the reference's self.done latches once set (a violation with done_on_cons_vio, pcgym.py:613-614) and stays
        # True until reset()
        self.done_flag = bool(self.done_flag or done[0].item())
        return o, r, self.done_flag, False, self.info
