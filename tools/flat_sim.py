#!/usr/bin/env python3
"""Offline model of the second pass of the barrier-free rollout (pcg_rollout_flat.hpp) on the ORACLE's attempt counts: what a
wave of 64 lanes pays, in instruction-issue slots, under different loop policies -- to decide what to build before building it.

A hot env hands over at step t*, then needs per step one boundary (START + POST: c_B slots for the wave whenever ANY of its
lanes runs it) and k attempts (c_A slots per iteration in which any lane attempts).  Lanes pull the next env from the list when
their episode ends.  Policies: boundaries every m-th iteration (or when no lane is mid-step), a attempts per iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import numpy as np
import bench
from oracle import oracle as O
from pcgym_amd.config import EnvSpec

B = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
_, params, _, _, _ = bench.single_workload("cstr_safe")
spec = EnvSpec(params)
env = O.OracleEnv(spec, B, seed=1234, n_threads=min(8, os.cpu_count() or 1))
env.reset()
rng = np.random.default_rng(0)
T = spec.N - 1
att = np.zeros((T, B), dtype=np.int64)
for t in range(T):
    env.step(rng.uniform(-1, 1, (spec.na, B)))
    att[t] = env.nsteps.sum(axis=0)
hot = att > 0
first = np.where(hot.any(axis=0), hot.argmax(axis=0), T)
order = np.argsort(first, kind="stable")
order = order[first[order] < T]  # the hand-over list, in the order the first pass appends (by step)
print(f"# B = {B}: {len(order)} hot envs, hot env steps {hot.sum()}, attempts {att.sum()}, per hot step {att.sum() / hot.sum():.2f}")
k_hot = att[hot]
print("attempts of a hot step: quantiles 50/90/99/99.9/max =", [int(np.quantile(k_hot, q)) for q in (0.5, 0.9, 0.99, 0.999, 1.0)],
      f"; share of all attempts in steps with > 12 attempts: {k_hot[k_hot > 12].sum() / k_hot.sum():.2f} (such steps: {np.mean(k_hot > 12):.3f} of hot steps)")


def simulate(cA, cB, every, natt, lanes_total, calm_steps_cost=None, split=None):
    """-> issue slots of the slowest wave and the mean (a wave = 64 lanes; lanes_total lanes pull from one list)"""
    nw = lanes_total // 64
    # deal the list to waves dynamically: simulate all waves in rounds of one iteration each, pulling from a shared head
    head = 0
    lst = order if split is None else split
    n = len(lst)
    e = -np.ones((nw, 64), dtype=np.int64)
    s = np.zeros((nw, 64), dtype=np.int64)
    rem = np.zeros((nw, 64), dtype=np.int64)   # attempts left in the current step (0 = at a boundary)
    cost = np.zeros(nw)
    fresh = np.zeros((nw, 64), dtype=bool)
    it = 0
    active = np.ones(nw, dtype=bool)
    while active.any():
        # pull (at boundaries only: an idle lane is "at a boundary")
        idle = (e < 0)
        need = int((idle & active[:, None]).sum())
        if need and head < n:
            take = min(need, n - head)
            idx = np.argwhere(idle & active[:, None])[:take]
            ids = lst[head:head + take]
            head += take
            e[idx[:, 0], idx[:, 1]] = ids
            s[idx[:, 0], idx[:, 1]] = first[ids]
            rem[idx[:, 0], idx[:, 1]] = -1  # needs START
        busy = e >= 0
        active = busy.any(axis=1)
        if not active.any():
            break
        adapting = busy & (rem > 0)
        any_ad = adapting.any(axis=1)
        if natt < 0:  # cohort policy: as many attempts as the lanes that started a step at the last boundary need, up to -natt
            want = np.where(busy & fresh, rem, 0).max(axis=1)
            nrun = np.clip(want, 1, -natt)
        else:
            nrun = np.full(nw, natt)
        cost += np.where(any_ad, cA * nrun, 0)
        rem = np.where(adapting, np.maximum(rem - nrun[:, None], 0), rem)
        fresh = np.zeros_like(busy)
        boundary = active & ((it % every == 0) | ~(busy & (rem > 0)).any(axis=1))
        at_b = busy & (rem <= 0) & boundary[:, None]
        cost += np.where(at_b.any(axis=1), cB, 0)
        # POST of the finished step (rem == 0) then START of the next (or of the first: rem == -1)
        fin = at_b & (rem == 0)
        s = np.where(fin, s + 1, s)
        done = at_b & (s >= T)
        e = np.where(done, -1, e)
        st = at_b & ~done
        ee, ss = e[st], s[st]
        fresh = st.copy()
        rem[st] = att[ss, ee]  # 0 attempts = a calm step of a handed-over env (the FIX block): counts as a boundary only
        rem = np.where(done, 0, rem)
        it += 1
    return cost.max(), cost.mean(), it


lanes = 1024 * 3 * 64 * B // (1 << 20)  # 3 waves per SIMD, scaled to this sample
lanes = max(64, lanes // 64 * 64)
print(f"# {lanes} lanes ({lanes // 64} waves) for {len(order)} hot envs = {len(order) / lanes:.2f} envs per lane")
base = None
for name, cA, cB, every, natt in (("first build: boundaries every iteration", 550, 650, 1, 1), ("boundaries every 2nd", 550, 650, 2, 1),
                                  ("every 3rd", 550, 650, 3, 1), ("every 2nd, two attempts per iteration", 550, 650, 2, 2),
                                  ("... with a lean boundary (350)", 550, 350, 2, 2), ("... lean boundary, every iteration", 550, 350, 1, 2),
                                  ("boundary only when nobody is mid-step (step-synchronous wave)", 550, 650, 10**9, 1),
                                  ("cohort: boundary every iteration, attempts = what the fresh lanes need, cap 3", 550, 650, 1, -3),
                                  ("cohort, cap 4", 550, 650, 1, -4), ("cohort, cap 6", 550, 650, 1, -6), ("cohort, cap 10", 550, 650, 1, -10),
                                  ("cohort cap 4, lean boundary", 550, 350, 1, -4), ("cohort cap 6, lean boundary", 550, 350, 1, -6)):
    mx, mean, it = simulate(cA, cB, every, natt, lanes)
    base = base or mx
    print(f"{name:64s}: slowest wave {mx / 1e6:7.3f} M slots ({mx / base:5.2f} of the first build), mean {mean / 1e6:7.3f} M, {it} iterations")
ideal = (att.sum() * 550 + hot.sum() * 650) / lanes
print(f"ideal (every lane always busy with useful work): {ideal / 1e6:.3f} M slots per wave")


def simulate_sorted(cA, cB, waves_per_wg, lanes_total, natt=1, overhead=80):
    """contexts of a WORKGROUP sorted by phase every round (LDS), its waves taking slices of 64 of a kind; a round ends with a
    barrier (its slowest wave); `overhead` slots per round for the sort, the context traffic and the barrier"""
    per_wg = waves_per_wg * 64
    nwg = max(1, lanes_total // per_wg)
    n = len(order)
    head = 0
    e = -np.ones((nwg, per_wg), dtype=np.int64)
    s = np.zeros((nwg, per_wg), dtype=np.int64)
    rem = np.zeros((nwg, per_wg), dtype=np.int64)
    cost = np.zeros(nwg)
    while True:
        idle = e < 0
        need = int(idle.sum())
        if need and head < n:
            take = min(need, n - head)
            idx = np.argwhere(idle)[:take]
            ids = order[head:head + take]
            head += take
            e[idx[:, 0], idx[:, 1]] = ids
            s[idx[:, 0], idx[:, 1]] = first[ids]
            rem[idx[:, 0], idx[:, 1]] = -1
        busy = e >= 0
        act = busy.any(axis=1)
        if not act.any():
            break
        nA = (busy & (rem > 0)).sum(axis=1)
        nB = (busy & (rem <= 0)).sum(axis=1)
        # slices of 64 of a kind; a remainder of fewer than 32 contexts of a kind WAITS for the next round (no mixed wave, no
        # second turn): the round costs its dearest kind
        rc = np.zeros(nwg)
        run_a = np.zeros_like(busy)
        run_b = np.zeros_like(busy)
        for g in np.where(act)[0]:
            ia = np.where(busy[g] & (rem[g] > 0))[0]
            ib = np.where(busy[g] & (rem[g] <= 0))[0]
            sa_, sb_ = len(ia) // 64 + (1 if len(ia) % 64 >= 32 else 0), len(ib) // 64 + (1 if len(ib) % 64 >= 32 else 0)
            if sa_ + sb_ == 0:  # nothing fills half a wave: run what there is
                sa_, sb_ = (1 if len(ia) else 0), (1 if len(ib) else 0)
            while sa_ + sb_ > waves_per_wg:
                if sa_ >= sb_:
                    sa_ -= 1
                else:
                    sb_ -= 1
            run_a[g, ia[: sa_ * 64]] = True
            run_b[g, ib[: sb_ * 64]] = True
            rc[g] = overhead + max(cA * natt if sa_ else 0, cB if sb_ else 0)
        cost += rc
        ad = run_a
        rem = np.where(ad, np.maximum(rem - natt, 0), rem)
        atb = run_b
        fin = atb & (rem == 0)
        s = np.where(fin, s + 1, s)
        done = atb & (s >= T)
        e = np.where(done, -1, e)
        st = atb & ~done
        rem[st] = att[s[st], e[st]]
        rem = np.where(done, 0, rem)
    return cost.max(), cost.mean()


print("# contexts sorted by phase across the waves of a workgroup every round (not built):")
for wpw in (4, 8, 12):
    for natt in (1, 2):
        mx, mean = simulate_sorted(550, 650, wpw, lanes, natt)
        print(f"  {wpw:2d} waves per workgroup, {natt} attempt(s) per round: slowest workgroup {mx / 1e6:7.3f} M slots ({mx / base:5.2f} of the first build), mean {mean / 1e6:7.3f} M")
