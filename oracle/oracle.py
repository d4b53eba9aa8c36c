"""ctypes wrapper around oracle/libpcg_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see pcg_oracle.c header).  It reuses the ABI struct *definitions* of
the product (pcgym_amd._abi / EnvSpec marshalling); the product never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpcg_oracle.so")
_lib = None
CALLS = [0]  # how often the oracle library was reached for (tests/conftest.py: which GPU tests check against the oracle)


def build(force=False):
    src = os.path.join(HERE, "pcg_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB


def lib():
    global _lib
    CALLS[0] += 1
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        from pcgym_amd import _abi as abi

        l = C.CDLL(LIB)
        vp = C.c_void_p
        l.orc_rhs.restype = C.c_int
        l.orc_rhs.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int64, vp, vp, vp]
        l.orc_integrate.restype = C.c_int
        l.orc_integrate.argtypes = [C.POINTER(abi.pcg_env_cfg), C.c_int64, vp, vp, vp]
        l.orc_step.restype = C.c_int
        l.orc_step.argtypes = [C.POINTER(abi.pcg_env_cfg), C.POINTER(abi.pcg_buffers), vp, C.c_int32,
                               C.c_uint64, C.c_int64, C.c_int]
        l.orc_reset.restype = C.c_int
        l.orc_reset.argtypes = [C.POINTER(abi.pcg_env_cfg), C.POINTER(abi.pcg_buffers), vp, vp, C.c_uint64,
                                C.c_int64]
        l.orc_set_reset_threads.restype = None
        l.orc_set_reset_threads.argtypes = [C.c_int]
        l.orc_pin_threads.restype = C.c_int
        l.orc_pin_threads.argtypes = [C.c_int]
        l.orc_unpin_threads.restype = C.c_int
        l.orc_unpin_threads.argtypes = []
        l.orc_philox4x32_10.restype = None
        l.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        l.orc_rng_normal.restype = C.c_double
        l.orc_rng_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
        l.orc_rng_uniform.restype = C.c_double
        l.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
        _lib = l
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data


_user_libs = {}  # keep the dlopen handles (and with them the registered function) alive


def register_user_rhs(spec):
    """PCG_MODEL_USER: compile the spec's C statements (the very text the plan hands to hipRTC) with gcc and register
    the function with the C oracle.  One user model at a time (the oracle keeps a single function pointer)."""
    import hashlib
    import tempfile

    src = ("#include <math.h>\nvoid pcg_user_rhs(const double* x, const double* u, const double* p, double* dx) {\n"
           "  (void)x; (void)u; (void)p;\n" + spec.user_rhs_src + "\n}\n")
    key = hashlib.sha1(src.encode()).hexdigest()[:16]
    if key not in _user_libs:
        d = os.path.join(tempfile.gettempdir(), "pcg_oracle_user")
        os.makedirs(d, exist_ok=True)
        c, so = os.path.join(d, key + ".c"), os.path.join(d, key + ".so")
        if not os.path.exists(so):
            with open(c, "w") as f:
                f.write(src)
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so + ".tmp", c, "-lm"])
            os.replace(so + ".tmp", so)
        _user_libs[key] = C.CDLL(so)
    fn = _user_libs[key].pcg_user_rhs
    l = lib()
    l.orc_set_user_rhs.restype = None
    l.orc_set_user_rhs.argtypes = [C.c_void_p]
    l.orc_set_user_rhs(C.cast(fn, C.c_void_p))


def rhs(model_id, params, x, u):
    """x (nx,B), u (nu,B) SoA float64 -> dx (nx,B)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    p = np.ascontiguousarray(params, dtype=np.float64)
    dx = np.empty_like(x)
    lib().orc_rhs(int(model_id), _p(p), x.shape[0], u.shape[0], x.shape[1], _p(x), _p(u), _p(dx))
    return dx


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


class OracleEnv:
    """Batched CPU environment with the same SoA buffers as the HIP VecEnv."""

    def __init__(self, spec, B, seed=0, per_env_t=False, env_offset=0, n_threads=1):
        from pcgym_amd import _abi as abi

        self.abi = abi
        self.spec = s = spec
        self.cfg, self._keep = s.to_cfg()
        self.B = B
        self.seed0, self.episode, self.t = int(seed), 0, 0
        self.env_offset = env_offset
        self.n_threads = n_threads
        self.x = np.zeros((s.nx, B))
        self.obs = np.zeros((s.nobs, B))
        self.rew = np.zeros(B)
        self.done = np.zeros(B, dtype=np.uint8)
        self.viol = np.zeros(B, dtype=np.uint8)
        self.status = np.zeros(B, dtype=np.uint8)
        self.slots = np.zeros((max(s.nsp + s.nd + s.nunc, 1), B))
        self.p_unc = np.zeros((s.nunc, B)) if s.nunc else None
        self.a_save = np.zeros((s.na, B)) if s.a_delta else None
        self.u_prev = np.full((s.na, B), np.nan) if s.reward_track is not None else None
        self.g = np.zeros((s.ncon, B)) if s.ncon else None
        self.g_pre = np.zeros((s.ncon, B)) if s.ncon else None
        self.t_env = np.zeros(B, dtype=np.int32) if per_env_t else None
        self.nsteps = np.zeros((2, B), dtype=np.int32) if s.integrator != "rk4" else None
        b = self.buf = abi.pcg_buffers()
        b.B = B
        b.x, b.obs, b.rew, b.done, b.viol = _p(self.x), _p(self.obs), _p(self.rew), _p(self.done), _p(self.viol)
        b.a_save, b.g, b.g_pre, b.t, b.nsteps = (_p(self.a_save), _p(self.g), _p(self.g_pre), _p(self.t_env),
                                                 _p(self.nsteps))
        b.p_unc = _p(self.p_unc)
        b.u_prev = _p(self.u_prev)
        b.status = _p(self.status)

    def _seed(self):
        return (self.seed0 + self.episode) & 0xFFFFFFFFFFFFFFFF

    def reset(self, mask=None):
        self.episode += 1
        if mask is None:
            self.t = 0
            self.status[:] = 0
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_set_reset_threads(int(self.n_threads))  # the partition of orc_step: first touch places each thread's slice
        lib().orc_reset(C.byref(self.cfg), C.byref(self.buf), _p(self.slots), _p(m), self._seed(), self.env_offset)
        return self.obs

    def step(self, a, d=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (self.spec.na, self.B)
        self.buf.a = _p(a)
        if d is not None:
            d = np.ascontiguousarray(d, dtype=np.float64)
            self.buf.d = _p(d)
        else:
            self.buf.d = None
        lib().orc_step(C.byref(self.cfg), C.byref(self.buf), _p(self.slots), self.t, self._seed(), self.env_offset,
                       self.n_threads)
        self.t += 1
        return self.obs, self.rew, self.done


def integrate(spec, x, u):
    """x (nx,B) in, u (nu,B): returns (x_out, nsteps (2,B))."""
    cfg, keep = spec.to_cfg()
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    u = np.ascontiguousarray(u, dtype=np.float64)
    ns = np.zeros((2, x.shape[1]), dtype=np.int32)
    lib().orc_integrate(C.byref(cfg), x.shape[1], _p(x), _p(u), _p(ns))
    return x, ns
