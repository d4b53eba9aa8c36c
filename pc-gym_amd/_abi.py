"""ctypes mirror of include/pcgym_hip.h (structs + constants).

tests/test_abi.py parses the header and checks every #define / enum value and
struct field order against this file, so the two cannot drift silently.
"""
from __future__ import annotations

import ctypes as C

PCG_ABI_VERSION = 14
PCG_MAX_NX = 24
PCG_MAX_NA = 5
PCG_MAX_NDM = 4
PCG_MAX_NSP = 4
PCG_MAX_NCON = 8
PCG_MAX_NUNC = 8
PCG_MAX_PARAMS = 128
PCG_MAX_USER_PARAMS = 64
PCG_MAX_NOBS = PCG_MAX_NX + PCG_MAX_NSP + PCG_MAX_NDM + PCG_MAX_NUNC
PCG_MAX_NU = PCG_MAX_NA + PCG_MAX_NDM
PCG_MAX_N = 4096

PCG_OK = 0
PCG_ST_OK = 0
PCG_ST_MAX_STEPS = 1
PCG_ST_UNDERFLOW = 2
PCG_ST_NONFINITE = 3
PCG_E_NULL = -1
PCG_E_MODEL = -2
PCG_E_DIM = -3
PCG_E_VALUE = -4
PCG_E_PLAN = -5
PCG_E_UNSUPPORTED = -6
PCG_E_JIT = -7

PCG_OPT_ENV_OFFSET = 1
PCG_OPT_LDS_STAGES = 2
PCG_OPT_VARIANT = 3
PCG_OPT_STREAM_BLOCKS_PER_CU = 4
PCG_OPT_NT_STORES = 5

PCG_INT_RK4 = 0
PCG_INT_DOPRI5 = 1
PCG_INT_RODAS3 = 2
PCG_INT_RODAS4 = 3
PCG_INT_TSIT5 = 4
PCG_INT_RK4G = 5
PCG_INT_T5G = 6
PCG_INT_CV8 = 7
PCG_INT_RODAS5 = 8

PCG_F_NORMALISE_A = 0x0001
PCG_F_NORMALISE_O = 0x0002
PCG_F_A_DELTA = 0x0004
PCG_F_R_PENALTY = 0x0008
PCG_F_DONE_ON_CONS = 0x0010
PCG_F_NOISE = 0x0020
PCG_F_REWARD_BATCH = 0x0040
PCG_F_MAXIMISE = 0x0080
PCG_F_REF_COMPAT = 0x0100
PCG_F_GAUSS_DIST = 0x0200
PCG_F_X0_NORMAL = 0x0400
PCG_F_UNC_EMPIRICAL = 0x0800
PCG_F_REWARD_TRACK = 0x1000
PCG_F_REWARD_CRYST = 0x2000
PCG_MAX_RBOX = 4
PCG_MAX_EMP = 65536

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pu8 = C.POINTER(C.c_uint8)


class pcg_env_cfg(C.Structure):
    _fields_ = [
        ("model_id", C.c_int32),
        ("integrator_id", C.c_int32),
        ("nx", C.c_int32),
        ("na", C.c_int32),
        ("ndm", C.c_int32),
        ("nd", C.c_int32),
        ("nsp", C.c_int32),
        ("nsp_obs", C.c_int32),
        ("ncon", C.c_int32),
        ("nrew", C.c_int32),
        ("nunc", C.c_int32),
        ("N", C.c_int32),
        ("substeps", C.c_int32),
        ("max_steps", C.c_int32),
        ("flags", C.c_uint32),
        ("n_params", C.c_int32),
        ("dt", C.c_double),
        ("rtol", C.c_double),
        ("atol", C.c_double),
        ("params", _pd),
        ("x0", _pd),
        ("x0_unc", _pd),
        ("a_low", _pd),
        ("a_high", _pd),
        ("a_act_low", _pd),
        ("a_act_high", _pd),
        ("a_0", _pd),
        ("o_low", _pd),
        ("o_high", _pd),
        ("obs_mask", _pu8),
        ("sp_index", _pi),
        ("sp", _pd),
        ("r_scale", _pd),
        ("rew_index", _pi),
        ("d_slot", _pi),
        ("d_sched", _pd),
        ("d_default", _pd),
        ("d_sigma", _pd),
        ("d_clip_lo", _pd),
        ("d_clip_hi", _pd),
        ("con_A", _pd),
        ("con_b", _pd),
        ("noise_pct", _pd),
        ("unc_index", _pi),
        ("unc_pct", _pd),
        ("unc_emp", _pd),
        ("unc_emp_off", _pi),
        ("rew_R_du", C.c_double),
        ("rew_R_u", C.c_double),
        ("rew_nbox", C.c_int32),
        ("rew_box_index", _pi),
        ("rew_box_lo", _pd),
        ("rew_box_hi", _pd),
        ("user_cons_src", C.c_char_p),
        ("user_reward_src", C.c_char_p),
        ("jit_include_dir", C.c_char_p),
        ("user_rhs_src", C.c_char_p),
        ("ep_frac", C.c_double),
        ("ep_kmax", C.c_int32),
        ("d_param_index", _pi),
        ("coop_thr", C.c_double),
    ]


class pcg_buffers(C.Structure):
    _fields_ = [
        ("B", C.c_int64),
        ("x", C.c_void_p),
        ("a", C.c_void_p),
        ("d", C.c_void_p),
        ("t", C.c_void_p),
        ("a_save", C.c_void_p),
        ("obs", C.c_void_p),
        ("rew", C.c_void_p),
        ("done", C.c_void_p),
        ("viol", C.c_void_p),
        ("g", C.c_void_p),
        ("g_pre", C.c_void_p),
        ("nsteps", C.c_void_p),
        ("u_prev", C.c_void_p),
        ("p_unc", C.c_void_p),
        ("status", C.c_void_p),
    ]


# every extern "C" symbol the header declares (tests check the .so exports all)
EXPORTS = [
    "pcg_version",
    "pcg_build_id",
    "pcg_strerror",
    "pcg_model_info",
    "pcg_model_default_params",
    "pcg_plan_create",
    "pcg_plan_destroy",
    "pcg_plan_bytes_per_env_step",
    "pcg_step",
    "pcg_reset",
    "pcg_plan_set_env_offset",
    "pcg_plan_set_option",
    "pcg_cfg_validate",
    "pcg_rhs",
    "pcg_integrate",
    "pcg_rollout",
    "pcg_rollout_strided",
    "pcg_step_autoreset",
    "pcg_graph_create",
    "pcg_graph_launch",
    "pcg_graph_set_seed",
    "pcg_graph_destroy",
    "pcg_last_jit_log",
    "pcg_philox4x32_10",
    "pcg_test_sort_tile",
    "pcg_coverage_names",
]


def declare(lib):
    """Attach argtypes/restype to a loaded libpcgym_hip.so."""
    vp = C.c_void_p
    lib.pcg_version.restype = C.c_int
    lib.pcg_version.argtypes = []
    lib.pcg_build_id.restype = C.c_char_p
    lib.pcg_build_id.argtypes = []
    lib.pcg_strerror.restype = C.c_char_p
    lib.pcg_strerror.argtypes = [C.c_int]
    lib.pcg_model_info.restype = C.c_int
    lib.pcg_model_info.argtypes = [C.c_int, _pi, _pi, _pi, _pi]
    lib.pcg_model_default_params.restype = C.c_int
    lib.pcg_model_default_params.argtypes = [C.c_int, _pd, C.c_int32]
    lib.pcg_plan_create.restype = C.c_int
    lib.pcg_plan_create.argtypes = [C.POINTER(vp), C.POINTER(pcg_env_cfg)]
    lib.pcg_plan_destroy.restype = C.c_int
    lib.pcg_plan_destroy.argtypes = [vp]
    lib.pcg_plan_bytes_per_env_step.restype = C.c_int64
    lib.pcg_plan_bytes_per_env_step.argtypes = [vp, C.POINTER(pcg_buffers)]
    lib.pcg_step.restype = C.c_int
    lib.pcg_step.argtypes = [vp, C.POINTER(pcg_buffers), C.c_int32, C.c_uint64, vp]
    lib.pcg_reset.restype = C.c_int
    lib.pcg_reset.argtypes = [vp, C.POINTER(pcg_buffers), vp, C.c_uint64, vp]
    lib.pcg_plan_set_env_offset.restype = C.c_int
    lib.pcg_plan_set_env_offset.argtypes = [vp, C.c_int64]
    lib.pcg_plan_set_option.restype = C.c_int
    lib.pcg_plan_set_option.argtypes = [vp, C.c_int, C.c_int64]
    lib.pcg_cfg_validate.restype = C.c_int
    lib.pcg_cfg_validate.argtypes = [C.POINTER(pcg_env_cfg)]
    lib.pcg_rhs.restype = C.c_int
    lib.pcg_rhs.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
    lib.pcg_integrate.restype = C.c_int
    lib.pcg_integrate.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
    lib.pcg_rollout.restype = C.c_int
    lib.pcg_rollout.argtypes = [vp, C.POINTER(pcg_buffers), C.c_int32, C.c_int32, vp, vp, vp,
                                C.c_uint64, vp]
    lib.pcg_rollout_strided.restype = C.c_int
    lib.pcg_rollout_strided.argtypes = [vp, C.POINTER(pcg_buffers), C.c_int32, C.c_int32, vp, C.c_int64, C.c_int64,
                                        vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_uint64, vp]
    lib.pcg_step_autoreset.restype = C.c_int
    lib.pcg_step_autoreset.argtypes = [vp, C.POINTER(pcg_buffers), C.c_int32, C.c_uint64, C.c_uint64, vp]
    lib.pcg_graph_create.restype = C.c_int
    lib.pcg_graph_create.argtypes = [C.POINTER(vp), vp, C.POINTER(pcg_buffers), C.POINTER(vp), C.POINTER(vp),
                                     C.c_int32, C.c_int32, C.c_uint64, C.c_int]
    lib.pcg_graph_launch.restype = C.c_int
    lib.pcg_graph_launch.argtypes = [vp, vp]
    lib.pcg_graph_set_seed.restype = C.c_int
    lib.pcg_graph_set_seed.argtypes = [vp, C.c_uint64]
    lib.pcg_graph_destroy.restype = C.c_int
    lib.pcg_graph_destroy.argtypes = [vp]
    lib.pcg_last_jit_log.restype = C.c_char_p
    lib.pcg_last_jit_log.argtypes = []
    lib.pcg_test_sort_tile.restype = C.c_int
    lib.pcg_test_sort_tile.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, vp]
    lib.pcg_coverage_names.restype = C.c_int64
    lib.pcg_coverage_names.argtypes = [C.c_char_p, C.c_int64, C.c_int]
    lib.pcg_philox4x32_10.restype = None
    lib.pcg_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32)]
    return lib
