#!/usr/bin/env python3
"""Does recording one episode (N-1 pcg_step launches) in a HIP graph shrink the launch-to-launch gap?
Run on the GPU box.  Prints us per step for: eager loop, graph incl. reset, graph + eager reset, the latter with
event brackets (bench.py's shape)."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench as B
from pcgym_amd import VecEnv

def timeit(fn, T, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * T) * 1e6

def main():
    nB = 1 << 20
    env = VecEnv(B.workload_params(nB), n_envs=nB, device="cuda:0", seed=1234, auto_reset=False)
    acts = 2 * torch.rand((64, 1, nB), device="cuda:0", dtype=torch.float64) - 1
    T = env.N - 1
    al = [acts[i % 64] for i in range(T)]
    def eager():
        env.reset()
        for i in range(T):
            env.step(al[i])
    print(f"eager                      {timeit(eager, T):.2f} us/step")
    g1 = env.capture_steps(al, with_reset=True)
    print(f"graph incl. reset          {timeit(g1.replay, T):.2f} us/step")
    env.reset()
    g2 = env.capture_steps(al)
    def g_reset():
        env.reset()
        g2.replay()
    print(f"graph + eager reset        {timeit(g_reset, T):.2f} us/step")
    st = torch.cuda.current_stream()
    evs = []
    def g_reset_ev():
        env.reset()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        g2.replay()
        b.record(st)
        evs.append((a, b))
    print(f"graph + eager reset + evs  {timeit(g_reset_ev, T):.2f} us/step")
    print(f"   bracket mean            {sum(a.elapsed_time(b) for a, b in evs) / len(evs) / T * 1e3:.2f} us/step")
    evs.clear()
    def g1_ev():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        g1.replay()
        b.record(st)
        evs.append((a, b))
    print(f"graph incl. reset + evs    {timeit(g1_ev, T):.2f} us/step")
    print(f"   bracket mean            {sum(a.elapsed_time(b) for a, b in evs) / len(evs) / T * 1e3:.2f} us/step (reset included)")
    t0 = time.perf_counter()
    for _ in range(50):
        g1.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host time per graph launch {(t1 - t0) / 50 * 1e6:.1f} us (queue backed up: includes blocking)")

main()
