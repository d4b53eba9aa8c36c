#!/usr/bin/env python3
"""cstr_safe (the default cstr plan on the full x0 box: a guarded launch for every env + a fix-up launch that is as long as its
heaviest env) with the batch cut into S chunks on S streams: does one chunk's fix-up chain hide the other chunks' guarded
launches?  Same envs (global env index keys the RNG), same actions.  usage: chunk_safe_probe.py [workload] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from pcgym_amd import MixedVecEnv

wl = sys.argv[1] if len(sys.argv) > 1 else "cstr_safe"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 236
name, params, B, _, n_act = bench.single_workload(wl)
dev = torch.device("cuda:0")
bench.clock_preheat(torch, dev, 100.0)
for S in (1, 2, 4, 8, 16):
    n = B // S
    menv = MixedVecEnv([(params, n)] * S, device=0, seed=1234, auto_reset=True, track_status=True)
    gen = torch.Generator(device=dev).manual_seed(1234)
    spec = menv.envs[0].spec
    acts = [2 * torch.rand((n_act, spec.na, n), generator=gen, device=dev, dtype=torch.float64) - 1 for _ in range(S)]
    menv.reset()
    torch.cuda.synchronize()
    def run(k):
        for i in range(k):
            menv.step([a[i % n_act] for a in acts], join=False)
        menv.join()
    run(24)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / K
    bad = sum(int(e.status.any()) for e in menv.envs)
    print(f"{wl} S {S:2d}: {dtm*1e6:9.2f} us per step of {n*S} envs -> {n*S/dtm:.4e} env-steps/s  status flags {bad}", flush=True)
    menv.close()
