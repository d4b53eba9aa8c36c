"""On the GPU: the cstr's ignition-front pairs (the envs the guarded default plan escalates and whose chain the fix-up
launch waits for) integrated (a) by the product's DOPRI5 at 1e-10 through pcg_integrate, one env per lane, and (b) by
tools/gbs_lanes_bench.hip: explicit extrapolation on EIGHT lanes per env.  Same pairs, both against the oracle's 1e-13
solve; compared is the slowest wave.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o _ab/gbs_lanes_bench.so tools/gbs_lanes_bench.hip
  python tools/gbs_lanes_probe.py [B]
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
rng = np.random.default_rng(4)
p_env = bench.workload_params()
del p_env["integrator"], p_env["substeps"]
p10 = dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-10, atol=1e-10)
ref = EnvSpec(dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-13, atol=1e-13))
d10 = EnvSpec(copy.deepcopy(p10))
raw, dt = np.array(ref.model.param_vector(), dtype=np.float64), ref.dt
x = np.stack([rng.uniform(0.7, 1.0, B), rng.uniform(310, 350, B)])
keep_x, keep_u = [], []
for t in range(60):
    u = rng.uniform(295, 302, (1, B))
    x2, ns = O.integrate(d10, x, u)
    a = ns.sum(axis=0)
    sel = np.argsort(a)[-16:]
    sel = sel[a[sel] > 30]
    keep_x.append(x[:, sel]); keep_u.append(u[:, sel])
    x = x2
xh = np.ascontiguousarray(np.concatenate(keep_x, axis=1)); uh = np.ascontiguousarray(np.concatenate(keep_u, axis=1))
n = xh.shape[1]
want, _ = O.integrate(ref, xh, uh)
rel = lambda y: float(np.nanmax(np.abs(y - want) / np.abs(want)))  # noqa: E731

eng = hip_integration_engine(env_params=copy.deepcopy(p10))
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
print(f"{n} heaviest (state, action) pairs of a {B}-env episode on the full x0 box (dt = {dt:.5f})")
print(f"(a) product DOPRI5 1e-10 via pcg_integrate, one env per lane ({(n + 63) // 64} waves): attempts max {att.max()} mean {att.mean():.1f} "
      f"= {7 * att.max()} dependent evaluations; launch {np.median(times[2:]):.1f} us (median of 6) = {np.median(times[2:]) / att.max():.2f} us per attempt of the heaviest env; "
      f"worst rel err vs 1e-13 solve {rel(xd.cpu().numpy()):.2e}")

so = C.CDLL(os.environ.get("GBS_SO", os.path.join(ROOT, "_ab", "gbs_lanes_bench.so")))
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
so.gbs_run.argtypes = [dp, dp, C.c_int, dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp, ip, dp, dp]
P = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
for h0, fmax, saf in ((0.25, 4.0, 0.9), (0.05, 2.0, 0.8), (0.02, 2.0, 0.8)):
    for tol in (1e-9, 1e-10, 1e-11):
        y = np.zeros_like(xh); st = np.zeros((2, n), dtype=np.int32); wus = np.zeros((n + 7) // 8); kus = C.c_double(0)
        rc = so.gbs_run(P(xh, dp), P(uh, dp), n, P(raw, dp), dt, tol, h0, fmax, saf, 6, P(y, dp), P(st, ip), P(wus, dp), C.byref(kus))
        assert rc == 0, rc
        big = st.sum(axis=0)
        print(f"(b) GBS, 8 lanes per env ({(n + 7) // 8} waves, one per workgroup) H0 = {h0} dt, growth <= {fmax}, safety {saf}, tol {tol:.0e}: big steps max {big.max()} mean {big.mean():.1f} "
              f"(rejected mean {st[1].mean():.2f}) = {17 * big.max()} dependent evaluations; slowest wave {wus.max():.1f} us, mean wave {wus.mean():.1f} us = "
              f"{wus.max() / big.max():.2f} us per big step; worst rel err {rel(y):.2e}")
