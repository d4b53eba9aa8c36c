#!/usr/bin/env python3
"""What the work-queue kernel's waves do (measurement build: the unit of the model compiled with -DPCG_QSTATS, loaded
through PCGYM_HIP_LIB).  Every wave records wall-clock stamps (100 MHz) at the phase boundaries and the counts of its
phase-2 loop: iterations, attempts executed, busy lanes summed over them, refills and the time inside them.
  UNIT=pcg_inst_j tools/fastlib.sh _ab/qstats_j.so -DPCG_QSTATS
  PCGYM_HIP_LIB=_ab/qstats_j.so python tools/queue_probe.py me20 [coop_thr]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcgym_amd import VecEnv  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "me20"
    dev = torch.device("cuda", 0)
    name, params, B, _, n_act = bench.single_workload(wl)
    if len(sys.argv) > 2:  # threshold of the cooperative rule of a Rodas4 plan (0 = off)
        thr = float(sys.argv[2])
        params["cooperative"] = {"thr": thr} if thr > 0 else False
        name += f" coop_thr {thr:g}"
    env = VecEnv(params, n_envs=B, device=dev, seed=1234, auto_reset=True)
    spec = env.spec
    gen = torch.Generator(device=dev).manual_seed(99)
    acts = bench.act_box(spec) * (2 * torch.rand((n_act, spec.na, B), generator=gen, device=dev, dtype=torch.float64) - 1) \
        + bench.act_shift(spec)
    env.reset()
    nrec = 1024 * 4
    st = torch.zeros((nrec, 16), dtype=torch.int64, device=dev)
    bench.clock_preheat(torch, dev, 100.0)
    for i in range(12):
        env.step(acts[i % n_act])
    torch.cuda.synchronize()
    env._buf.g = st.data_ptr()
    rows = []
    for i in range(6):
        st.zero_()
        env.step(acts[i % n_act])
        torch.cuda.synchronize()
        rows.append(st.cpu().numpy().astype(np.float64))
    env._buf.g = None
    print(name)
    for r in rows[-3:]:
        u = r[r[:, 0] > 0]
        t0 = u[:, 0].min()
        us = lambda v: (v - t0) / 100.0
        q = lambda v: "min %8.1f  p50 %8.1f  p90 %8.1f  max %8.1f" % tuple(np.percentile(v, [0, 50, 90, 100]))
        print(f"waves {len(u)}   launch span by the stamps {us(u[:, 5]).max():.1f} us")
        print("  phase 1 (load, pre, h_init, park)  us ", q((u[:, 1] - u[:, 0]) / 100))
        print("  sort (+ barrier)                   us ", q((u[:, 2] - u[:, 1]) / 100))
        print("  phase 2, own wave                  us ", q((u[:, 3] - u[:, 2]) / 100))
        if (u[:, 13] > 0).any():  # cooperative phase of a Rodas4 tile (stamp 13 = its end, 14 / 15 = big steps / busy groups)
            cp = (u[:, 13] - u[:, 2]) / 100
            print("  ... of it the cooperative phase   us ", q(cp))
            print("      big steps executed per wave       ", q(u[:, 14]))
            print("      busy groups per big step (of 8)   ", q(u[:, 15] / np.maximum(u[:, 14], 1)))
            print(f"      time per big step: {np.median(cp[u[:, 14] > 0] / u[u[:, 14] > 0, 14]):.3f} us; envs through the phase {u[:, 15].sum():.0f} group-steps")
        print("  wait for the workgroup's last wave us ", q((u[:, 4] - u[:, 3]) / 100))
        print("  phase 3                            us ", q((u[:, 5] - u[:, 4]) / 100))
        print("  end of the wave since launch start us ", q(us(u[:, 5])))
        att, busy, it, rf, rclk, pop = u[:, 7], u[:, 8], u[:, 6], u[:, 9], u[:, 10], u[:, 11]
        print("  attempts executed per wave            ", q(att))
        print("  lane utilisation inside attempts      ", q(busy / np.maximum(att, 1) / 64))
        print("  loop iterations / refills / pops      ", f"{it.mean():.1f} / {rf.mean():.1f} / {pop.mean():.1f}")
        print("  time in refills per wave           us ", q(rclk / 100))
        p2 = (u[:, 3] - u[:, 2]) / 100
        print(f"  phase-2 time per attempt: {np.median(p2 / np.maximum(att, 1)):.3f} us (incl. refills); "
              f"excl. refills {np.median((p2 - rclk / 100) / np.maximum(att, 1)):.3f} us")
        print(f"  lanes that saw an out-of-order pair after the sort: {int(u[:, 12].sum())}")
        print(f"  busy-lane attempts in total {busy.sum():.0f} = {busy.sum() / B:.2f} per env")
    env.close()


if __name__ == "__main__":
    main()
