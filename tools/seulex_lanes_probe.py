"""On the GPU: the heaviest envs of a Rodas4 launch (ME-10, the me10_ros4 workload) integrated (a) by the product's Rodas4
through pcg_integrate, one env per lane, and (b) by tools/seulex_lanes_bench.hip: extrapolated linearly implicit Euler on
EIGHT lanes per env.  Both on the same (state, action) pairs, both checked against the oracle's 1e-13 solve; what is
compared is the time of the slowest wave -- the chain a work-queue launch waits for.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o _ab/seulex_lanes_bench.so tools/seulex_lanes_bench.hip
  python tools/seulex_lanes_probe.py [B]
"""
import copy
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pcgym_amd import _lib  # noqa: E402
from pcgym_amd.config import EnvSpec  # noqa: E402
from pcgym_amd.reference_engine import hip_integration_engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(7)
_, p_env, _, _, _ = bench.single_workload("me10_ros4")
r4 = EnvSpec(copy.deepcopy(p_env))
ref = EnvSpec(dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-13, atol=1e-13))
raw, dt = np.array(ref.model.param_vector(), dtype=np.float64), ref.dt
lo, hi = r4.a_low, r4.a_high
x = np.tile(np.array(r4.x0[: r4.nx], dtype=float)[:, None], (1, B)) * (1 + 0.05 * rng.uniform(-1, 1, (r4.nx, B)))
keep_x, keep_u = [], []
for t in range(4):  # the oracle walks the episode on the host; the 128 heaviest pairs of every step are kept
    u = lo[:, None] + rng.uniform(0, 1, (r4.na, B)) * (hi - lo)[:, None]
    x2, ns = O.integrate(r4, x, u)
    a = ns.sum(axis=0)
    sel = np.argsort(a)[-128:]
    keep_x.append(x[:, sel]); keep_u.append(u[:, sel])
    x = x2
xh = np.ascontiguousarray(np.concatenate(keep_x, axis=1)); uh = np.ascontiguousarray(np.concatenate(keep_u, axis=1))
n = xh.shape[1]
want, _ = O.integrate(ref, xh, uh)
rel = lambda y: float(np.nanmax(np.abs(y - want) / np.abs(want)))  # noqa: E731

# (a) the product: Rodas4 at the plan's default tolerance with end-point control, one env per lane
eng = hip_integration_engine(env_params=copy.deepcopy(p_env))
lib = _lib.load()
xd = torch.tensor(xh, device="cuda"); ud = torch.tensor(uh, device="cuda")
nsd = torch.zeros((2, n), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
times = []
for r in range(8):
    xd.copy_(torch.tensor(xh, device="cuda")); torch.cuda.synchronize()
    e0.record()
    _lib.check(lib.pcg_integrate(eng._plan, n, xd.data_ptr(), ud.data_ptr(), nsd.data_ptr(), s), "pcg_integrate")
    e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) * 1e3)
att = nsd.sum(dim=0).cpu().numpy()
print(f"{n} heaviest (state, action) pairs of {B} envs x 4 steps (me10_ros4 workload, dt = {dt})")
print(f"(a) product Rodas4 via pcg_integrate, one env per lane ({(n + 63) // 64} waves): attempts max {att.max()} mean {att.mean():.1f} "
      f"= {6 * att.max()} dependent stages; launch {np.median(times[2:]):.1f} us (median of 6) = {np.median(times[2:]) / att.max():.2f} us per attempt of the heaviest env; "
      f"worst rel err vs 1e-13 solve {rel(xd.cpu().numpy()):.2e}")

# (b) eight lanes per env
so = C.CDLL(os.environ.get("SEULEX_SO", os.path.join(ROOT, "_ab", "seulex_lanes_bench.so")))
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
so.seulex_run.argtypes = [dp, dp, C.c_int, dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp, ip, dp, dp]
P = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
for h0, fmax, saf in ((0.25, 4.0, 0.9), (0.05, 4.0, 0.9), (0.05, 2.0, 0.9), (0.05, 3.0, 0.8), (0.02, 2.0, 0.8)):
    for tol in (3e-7, 1e-7, 3e-8):
        y = np.zeros_like(xh); st = np.zeros((2, n), dtype=np.int32); wus = np.zeros((n + 7) // 8); kus = C.c_double(0)
        rc = so.seulex_run(P(xh, dp), P(uh, dp), n, P(raw, dp), dt, tol, h0, fmax, saf, 6, P(y, dp), P(st, ip), P(wus, dp), C.byref(kus))
        assert rc == 0, rc
        big = st.sum(axis=0)
        print(f"(b) SEULEX, 8 lanes per env ({(n + 7) // 8} waves, one per workgroup) H0 = {h0} dt, growth <= {fmax}, safety {saf}, tol {tol:.0e}: big steps max {big.max()} mean {big.mean():.1f} "
              f"(rejected mean {st[1].mean():.2f}) = {8 * big.max()} dependent sub-steps; slowest wave {wus.max():.1f} us, mean wave {wus.mean():.1f} us = "
              f"{wus.max() / big.max():.2f} us per big step; worst rel err {rel(y):.2e}")
