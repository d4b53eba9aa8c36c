#!/usr/bin/env python3
"""Per-wave timeline of the classic cstr step kernel (build with -DPCG_TIMELINE, run on the GPU box).
Prints, relative to the earliest wave start, the distribution of: wave start, loads landed,
arithmetic done, stores issued, stores acknowledged (s_memtime ticks -> us)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "pc-gym_amd", "csrc", "pcg_kernels.hip")
tl = "/tmp/libpcgym_hip_tl.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                       "-fvisibility=hidden", "-DPCG_TIMELINE", "-shared", "-o", tl, src])
import numpy as np
import torch

from pcgym_amd import _lib

_lib.LIB_PATH = tl
import bench
from pcgym_amd import VecEnv

B = 1 << 20
substeps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = bench.workload_params(B)
p["substeps"] = substeps
env = VecEnv(p, n_envs=B, variant=1)
dbg = torch.zeros((B // 64) * 8, dtype=torch.int64, device="cuda")
env._buf.nsteps = dbg.data_ptr()
env.reset()
a = 2 * torch.rand((1, B), device="cuda", dtype=torch.float64) - 1
for i in range(5):
    env.step(a)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 8)
t = d[:, :5].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0)
# s_memtime ticks: find tick rate from the kernel duration measured by events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env.step(a); e1.record(); torch.cuda.synchronize()
dur_us = e0.elapsed_time(e1) * 1e3
d = dbg.cpu().numpy().reshape(-1, 8)
t = d[:, :5].astype(np.float64)
t = t - t[:, 0].min()
span = t[:, 4].max()
print(f"kernel (events) {dur_us:.2f} us ; stamp span {span:.0f} ticks -> {span/dur_us:.1f} ticks/us (assuming span ~ duration)")
tick = span / dur_us
names = ["wave start", "loads landed", "arith done", "stores issued", "stores acked"]
for i, n in enumerate(names):
    v = np.sort(t[:, i]) / tick
    print(f"{n:14s} min {v[0]:6.2f}  p10 {v[len(v)//10]:6.2f}  p50 {v[len(v)//2]:6.2f}  p90 {v[len(v)*9//10]:6.2f}  max {v[-1]:6.2f} us")
dl = (t[:, 1] - t[:, 0]) / tick
dc = (t[:, 2] - t[:, 1]) / tick
ds = (t[:, 4] - t[:, 2]) / tick
for n, v in (("load latency", dl), ("arith span", dc), ("store span", ds)):
    v = np.sort(v)
    print(f"per-wave {n:13s} p10 {v[len(v)//10]:6.2f}  p50 {v[len(v)//2]:6.2f}  p90 {v[len(v)*9//10]:6.2f} us")
# occupancy over time: how many waves are in each phase at sample instants
for ts in np.linspace(0, span, 11)[1:-1]:
    ph = [(t[:, 0] <= ts) & (ts < t[:, 1]), (t[:, 1] <= ts) & (ts < t[:, 2]), (t[:, 2] <= ts) & (ts < t[:, 4])]
    print(f"t={ts/tick:6.2f} us  waves loading {ph[0].sum():6d}  computing {ph[1].sum():6d}  storing {ph[2].sum():6d}  "
          f"not started {(t[:,0] > ts).sum():6d}  done {(t[:,4] <= ts).sum():6d}")
