#!/usr/bin/env python3
"""Does splitting the batch into C independent chunks on C HIP streams hide the per-launch fill/drain?
Same workload as bench.py; value = total env-steps / wall."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pcgym_amd import VecEnv, _lib

B, K = 1 << 20, 1180
lib = _lib.load()
for C in (1, 2, 3, 4):
    Bc = B // C // 2 * 2
    envs = [VecEnv(bench.workload_params(Bc), n_envs=Bc, seed=1234, env_offset=c * Bc) for c in range(C)]
    streams = [torch.cuda.Stream() for _ in range(C)]
    gen = torch.Generator(device="cuda").manual_seed(1)
    acts = [2 * torch.rand((64, 1, Bc), generator=gen, device="cuda", dtype=torch.float64) - 1 for _ in range(C)]
    for e, s in zip(envs, streams):
        with torch.cuda.stream(s):
            e.reset()
    torch.cuda.synchronize()

    def run(n):
        for i in range(n):
            for e, s, a in zip(envs, streams, acts):
                e._buf.a = a[i % 64].data_ptr()
                rc = lib.pcg_step(e._plan, e._bufp, e.t, e._episode_seed(), s.cuda_stream)
                assert rc == 0
                e.t += 1
                if e.t == e.N - 1:
                    with torch.cuda.stream(s):
                        e.reset()
    run(59)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"chunks {C}: {Bc*C*K/el:.3e} env-steps/s, {el/K*1e6:.2f} us per step of {Bc*C} envs, GB/s {73*Bc*C*K/el/1e9:.0f}", flush=True)
    for e in envs:
        e.close()
