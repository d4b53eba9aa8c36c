#!/usr/bin/env python3
"""Host-side trajectory gather, measured (GPU box): env-steps/s of the cstr bench workload with the per-step outputs
(a) left on the device, (b) gathered to pinned host memory through pcgym_amd.HostGather (obs + rew + done, and rew +
done only), (c) gathered synchronously (.cpu() per step, the naive way).  PCIe Gen5 x16 is 63 GB/s on paper."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench as BN
from pcgym_amd import HostGather, VecEnv

B = 1 << 20
env = VecEnv(BN.workload_params(), n_envs=B, seed=1, auto_reset=True)
env.reset()
acts = 2 * torch.rand((16, 1, B), device=env.device, dtype=torch.float64) - 1

def loop(K, after=None):
    for i in range(50):
        env.step(acts[i % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    last = None
    for i in range(K):
        env.step(acts[i % 16])
        if after is not None:
            last = after()
    if isinstance(last, int):
        g.wait(last)
    torch.cuda.synchronize()
    return time.perf_counter() - t0

K = 2000
w = loop(K)
print(f"outputs stay on the device          : {w/K*1e6:8.2f} us/step  {B*K/w:.3e} env-steps/s")
for fields in (("obs", "rew", "done"), ("rew", "done")):
    g = HostGather(env, fields)
    K2 = 300
    w = loop(K2, g.push)
    out = g.wait((g._k - 1) & 1)
    assert torch.equal(out["rew"], env.rew.cpu())
    print(f"HostGather {'+'.join(fields):24s}: {w/K2*1e6:8.2f} us/step  {B*K2/w:.3e} env-steps/s  "
          f"{g.bytes_per_step/1e6:.1f} MB/step -> {g.bytes_per_step*K2/w/1e9:.1f} GB/s over PCIe")
K3 = 100
w = loop(K3, lambda: (env.obs_soa.cpu(), env.rew.cpu(), env.done.cpu()) and None)
print(f"synchronous .cpu() per step         : {w/K3*1e6:8.2f} us/step  {B*K3/w:.3e} env-steps/s")
