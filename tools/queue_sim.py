#!/usr/bin/env python3
"""Offline model of the work-queue step kernel (pcg_step_queue.hpp) on the ORACLE's attempt counts of one env step of a bench
workload: what the launch pays when every CU owns a fixed tile, and what a global pool for the tail of every tile would buy
(VERDICT r5 "next" 4) -- to decide before building.

Model: 256 workgroups (one per CU), each with W waves of 64 lanes on a contiguous tile of B / 256 envs sorted by a cost key
(true cost x log-normal noise: correlation ~0.9, as the kernels' keys have); a lane runs one env at a time, one attempt per wave
iteration; idle lanes refill from the tile's queue when >= 8 are idle or nobody is busy.  The launch ends with its slowest wave.
Pool: the cheapest fraction p of every tile goes to one global list instead (cheapest first in, the list is served heaviest
first); a wave whose tile queue is dry pulls from it under the same refill rule.
usage: queue_sim.py [me10|me10_ros5|me20] [log2 B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import numpy as np
import bench
from oracle import oracle as O
from pcgym_amd.config import EnvSpec

wl = sys.argv[1] if len(sys.argv) > 1 else "me10_ros5"
B = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 18)
_, params, _, _, _ = bench.single_workload(wl)
spec = EnvSpec(params)
env = O.OracleEnv(spec, B, seed=1234, n_threads=min(8, os.cpu_count() or 1))
env.reset()
rng = np.random.default_rng(0)
env.step(rng.uniform(-1, 1, (spec.na, B)))   # the state the bench's steady state looks like: one step in
env.step(rng.uniform(-1, 1, (spec.na, B)))
cost = env.nsteps.sum(axis=0).astype(np.int64)
key = cost * np.exp(0.18 * rng.standard_normal(B))
print(f"# {wl}: B = {B}, attempts per env step mean {cost.mean():.1f}, max {cost.max()}, cv {cost.std() / cost.mean():.2f}; "
      f"correlation of the sort key with the cost {np.corrcoef(key, cost)[0, 1]:.2f}")
NWG = 256
W = 8 if wl == "me10" else 4  # the explicit pair of the 10-state cascade runs 512-thread workgroups


def run(p):
    """-> (iterations of every wave [NWG, W], busy lane-iterations)"""
    per = B // NWG
    own, pool = [], []
    for g in range(NWG):
        idx = np.arange(g * per, (g + 1) * per)
        o = idx[np.argsort(-key[idx], kind="stable")]
        k = int(round(per * (1 - p)))
        own.append(o[:k])
        pool.append(o[k:])
    pool = np.concatenate(pool) if p > 0 else np.zeros(0, dtype=np.int64)
    pool = pool[np.argsort(-key[pool], kind="stable")]
    ph = 0
    heads = np.zeros(NWG, dtype=np.int64)
    rem = np.zeros((NWG, W, 64), dtype=np.int64)
    iters = np.zeros((NWG, W), dtype=np.int64)
    alive = np.ones((NWG, W), dtype=bool)
    busy_sum = 0
    # initial hand-out
    for g in range(NWG):
        n0 = min(len(own[g]), W * 64)
        r = np.zeros(W * 64, dtype=np.int64)
        r[:n0] = cost[own[g][:n0]]
        rem[g] = r.reshape(W, 64) if True else r
        heads[g] = n0
    while alive.any():
        # refill, wave by wave (the order of the waves within an iteration does not matter for the totals)
        idle = rem <= 0
        nidle = idle.sum(axis=2)
        nobusy = nidle == 64
        want = alive & ((nidle >= 8) | nobusy)
        for g, w in np.argwhere(want):
            k = int(nidle[g, w])
            take_own = min(k, len(own[g]) - heads[g])
            vals = []
            if take_own > 0:
                vals.append(cost[own[g][heads[g]:heads[g] + take_own]])
                heads[g] += take_own
            k2 = k - take_own
            if k2 > 0 and ph < len(pool):
                t = min(k2, len(pool) - ph)
                vals.append(cost[pool[ph:ph + t]])
                ph += t
            if vals:
                v = np.concatenate(vals)
                pos = np.where(idle[g, w])[0][: len(v)]
                rem[g, w, pos] = v
        busy = rem > 0
        anyb = busy.any(axis=2)
        alive = anyb
        iters += anyb
        busy_sum += int(busy.sum())
        rem = np.where(busy, rem - 1, rem)
    return iters, busy_sum


base = None
for p in (0.0, 0.05, 0.1, 0.15, 0.25, 0.4):
    it, bs = run(p)
    end = it.max()
    base = base or end
    wg = it.max(axis=1)
    print(f"pool {p:4.2f}: launch = {end} wave iterations ({end / base:5.3f} of the fixed tiles), slowest / median workgroup {wg.max() / np.median(wg):.3f}, "
          f"mean wave {it.mean():.1f}, lane utilisation over the launch {bs / (end * NWG * W * 64):.3f}, needed per lane {cost.sum() / (NWG * W * 64):.1f}")
