#!/usr/bin/env python3
"""Where the ~2 us per step between ms_per_step and the kernel time of a driver-shaped run (--steps 20 --warmup 5) goes:
the same 20 launches bracketed three ways -- torch.cuda.synchronize() (the contract's bracket), a spin on event.query()
before it, hipStreamSynchronize -- each repeated, with the launch loop's host time and the GPU-side span by events."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcgym_amd import VecEnv  # noqa: E402


def main():
    B, K = 1 << 20, 20
    dev = torch.device("cuda", 0)
    env = VecEnv(bench.workload_params(), n_envs=B, device=dev, seed=1234, auto_reset=True, track_status=True)
    acts = 2 * torch.rand((64, 1, B), device=dev, dtype=torch.float64) - 1
    env.reset()
    stream = torch.cuda.current_stream(dev)
    lib, plan, bufp, buf = env._lib, env._plan, env._bufp, env._buf
    sp = stream.cuda_stream

    def launches(t0):
        for j in range(K):
            buf.a = acts[(t0 + j) % 64].data_ptr()
            lib.pcg_step(plan, bufp, (t0 + j) % 50, 1, sp)

    bench.clock_preheat(torch, dev, 100.0)
    for mode in ("sync", "spin", "spin_nosleep_events_off"):
        rows = []
        for rep in range(12):
            launches(0)
            torch.cuda.synchronize()
            time.sleep(0.002)
            eb, ee = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode != "spin_nosleep_events_off":
                eb.record(stream)
            launches(5)
            t1 = time.perf_counter()
            ee.record(stream)
            if mode != "sync":
                while not ee.query():
                    pass
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            gpu = eb.elapsed_time(ee) * 1e3 if mode != "spin_nosleep_events_off" else float("nan")
            rows.append(((t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, gpu))
        rows.sort()
        med = rows[len(rows) // 2]
        print(f"{mode:26s} region {med[0]:7.1f} us = {med[0] / K:6.2f} us/step | host launch loop {med[1]:6.1f} | wait {med[2]:6.1f} | "
              f"final synchronize {med[3]:5.1f} | GPU span by events {med[4]:7.1f} ({med[4] / K:5.2f}/step)  [min region {rows[0][0]:.1f}]")
    env.close()


if __name__ == "__main__":
    main()
