#!/usr/bin/env python3
"""Per-wave timeline of the headline step kernel (measurement build: csrc compiled with -DPCG_TIMELINE, loaded through
PCGYM_HIP_LIB).  Every wave stamps the 100 MHz wall clock at: 0 start, 1 inputs landed, 2 integration done, 3 stores
issued, 4 stores acknowledged (last tile only); a persistent wave's second tile uses a second record.  Prints when each
phase starts / ends over the launch and how many waves sit in which phase over time.
  PCGYM_HIP_LIB=_ab/var/lib_new_TL.so python tools/timeline_probe.py [PCG_NT / PCG_BPC in the environment]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcgym_amd import VecEnv  # noqa: E402


def main():
    B = 1 << 20
    dev = torch.device("cuda", 0)
    env = VecEnv(bench.workload_params(), n_envs=B, device=dev, seed=1234, auto_reset=True, track_status=True)
    acts = 2 * torch.rand((8, 1, B), device=dev, dtype=torch.float64) - 1
    env.reset()
    nrec = 4096 * 4 * 2  # up to 4096 workgroups x 4 waves x 2 tiles
    tl = torch.zeros((nrec, 8), dtype=torch.int64, device=dev)
    bench.clock_preheat(torch, dev, 100.0)
    for i in range(40):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    env._buf.g = tl.data_ptr()
    for i in range(3):
        tl.zero_()
        env.step(acts[i % 8])
        torch.cuda.synchronize()
    env._buf.g = None
    t = tl.cpu().numpy().astype(np.float64)
    used = t[:, 0] > 0
    first = used & (np.arange(nrec) % 2 == 0)
    second = (t[:, 1] > 0) & (np.arange(nrec) % 2 == 1)
    t0 = t[first, 0].min()
    us = lambda v: (v - t0) / 100.0
    print(f"waves {first.sum()}  waves with a second tile {second.sum()}")

    def pct(name, v):
        q = np.percentile(v, [0, 10, 50, 90, 100])
        print(f"  {name:44s} min {q[0]:6.2f}  p10 {q[1]:6.2f}  p50 {q[2]:6.2f}  p90 {q[3]:6.2f}  max {q[4]:6.2f} us")

    a = t[first]
    pct("wave start (since the first wave's start)", us(a[:, 0]))
    pct("inputs landed", us(a[:, 1]))
    pct("  load latency (start -> landed)", (a[:, 1] - a[:, 0]) / 100)
    pct("integration done", us(a[:, 2]))
    pct("  integration time", (a[:, 2] - a[:, 1]) / 100)
    pct("stores issued", us(a[:, 3]))
    if second.any():
        b = t[second]
        pct("second tile: landed", us(b[:, 1]))
        pct("second tile: integration done", us(b[:, 2]))
        pct("  second tile: integration time", (b[:, 2] - b[:, 1]) / 100)
        pct("second tile: stores issued", us(b[:, 3]))
    ack = np.where(t[:, 4] > 0)[0]
    pct("stores acknowledged (wave end)", us(t[ack, 4]))
    pct("  last store issue -> acknowledged", (t[ack, 4] - t[ack, 3]) / 100)
    end = us(t[ack, 4]).max()
    print(f"  launch span by the stamps: {end:.2f} us")
    # occupancy of the phases over time
    edges = np.arange(0.0, end + 0.5, 0.5)
    print("  t(us)   loading integrating storing(ack pending)")
    recs = [t[first]] + ([t[second]] if second.any() else [])
    for lo in edges:
        n_load = n_int = n_st = 0
        for r in recs:
            s0, s1, s2, s3 = us(r[:, 0]), us(r[:, 1]), us(r[:, 2]), us(r[:, 3])
            n_load += int(((s0 <= lo) & (lo < s1)).sum())
            n_int += int(((s1 <= lo) & (lo < s2)).sum())
        s3a, s4a = us(t[ack, 3]), us(t[ack, 4])
        n_st = int(((s3a <= lo) & (lo < s4a)).sum())
        print(f"  {lo:5.1f}  {n_load:7d} {n_int:11d} {n_st:8d}")
    env.close()


if __name__ == "__main__":
    main()
