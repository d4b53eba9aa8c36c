"""Mixed-model batches (BASELINE.json configs[4]: "mixed {cstr, four_tank, ME} ... sorted by model").

A wavefront must be homogeneous in the model it integrates, so a mixed batch is a list of model-homogeneous
SEGMENTS: contiguous env ranges, each with its own configuration, plan and HIP stream.  The segments of one shard run
concurrently (independent streams, no dependency between them); global env indices are contiguous over the
segments, so the counter-based RNG streams are disjoint and a sharded mixed batch draws the same numbers as an
unsharded one.  The reference has no counterpart (one Python object per env, `pcgym.py:31`); this is the batched
form of "a list of make_env objects of different models".
"""
from __future__ import annotations

from .env import VecEnv


def _torch():
    import torch

    return torch


class MixedVecEnv:
    """segments: sequence of (env_params, n_envs[, global_env_offset]).  Actions / results are lists, one entry per
    segment.  Without explicit offsets the segments are laid out back to back from `env_offset`."""

    def __init__(self, segments, device=None, seed=0, env_offset=0, timing=False, **kw):
        torch = _torch()
        self.envs, self.offsets = [], []
        self.timing = bool(timing)  # record a hipEvent pair around every segment's step launch (bench.py roofline)
        self._events = None
        off = int(env_offset)
        for seg in segments:
            params, n = seg[0], int(seg[1])
            if len(seg) > 2:
                off = int(seg[2])
            self.offsets.append(off)
            self.envs.append(VecEnv(params, n_envs=n, device=device, seed=seed, env_offset=off, **kw))
            off += n
        if not self.envs:
            raise ValueError("a mixed batch needs at least one segment")
        self.device = self.envs[0].device
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.Stream(device=self.device) for _ in self.envs]
        self.B = sum(e.B for e in self.envs)
        self._events = [[] for _ in self.envs]

    def __len__(self):
        return len(self.envs)

    def _each(self, fn, timed=False, join=True):
        """Run fn(i, env) for every segment on that segment's stream; with `join` the caller's stream waits for all of
        them (so results can be consumed on it without a host synchronisation) and every segment's stream has waited for
        the caller's (so inputs produced on it are visible).  join=False skips both hand-shakes: each segment simply
        continues on its own stream -- for callers whose inputs are already resident and who consume the outputs later
        (call join() then): consecutive steps of a segment then run back to back instead of meeting the slowest
        segment of the previous step at a cross-stream barrier."""
        torch = _torch()
        cur = torch.cuda.current_stream(self.device)
        out = []
        for i, (e, s) in enumerate(zip(self.envs, self.streams)):
            if join:
                s.wait_stream(cur)
            with torch.cuda.stream(s):
                if timed:
                    eb, ee = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    eb.record(s)
                out.append(fn(i, e))
                if timed:
                    ee.record(s)
                    self._events[i].append((eb, ee))
        if join:
            for s in self.streams:
                cur.wait_stream(s)
        return out

    def join(self):
        """make the caller's stream wait for everything the segments have been given so far (after step(join=False))"""
        cur = _torch().cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def reset(self, seed=None):
        return self._each(lambda i, e: e.reset(seed=seed))

    def step(self, actions, disturbances=None, join=True):
        """actions: one (na_i, B_i) SoA tensor (or (B_i, na_i)) per segment -> list of step() tuples."""
        if len(actions) != len(self.envs):
            raise ValueError(f"one action tensor per segment ({len(self.envs)}) is required")
        d = disturbances or [None] * len(self.envs)
        return self._each(lambda i, e: e.step(actions[i], d[i]), timed=self.timing, join=join)

    def segment_times(self):
        """[(total ms, launches)] per segment of the step launches recorded while `timing` was on (synchronises)."""
        _torch().cuda.synchronize(self.device)
        return [(sum(eb.elapsed_time(ee) for eb, ee in ev), len(ev)) for ev in self._events]

    @property
    def bytes_per_step(self):
        return sum(e.bytes_per_env_step * e.B for e in self.envs)

    def close(self):
        for e in self.envs:
            e.close()


def mixed_shard_layout(segments_global, rank, world):
    """[(env_params, n_local, global_env_offset)] of rank `rank`: every rank takes the same fraction of each global
    segment (so every shard holds the same model mix and the per-GPU work is balanced); env offsets follow the
    global layout [segment 0 | segment 1 | ...].  Pure host logic (tested on CPU with two gloo ranks)."""
    from .shard import shard_range

    local, base = [], 0
    for params, n in segments_global:
        lo, hi = shard_range(int(n), rank, world)
        local.append((params, hi - lo, base + lo))
        base += int(n)
    return local


def make_mixed_sharded_env(segments_global, rank=None, world=None, device=None, **kw):
    """MixedVecEnv over this rank's slices (mixed_shard_layout) of a global mixed batch; RANK / WORLD_SIZE /
    LOCAL_RANK from the environment when not given.  kw is forwarded to MixedVecEnv / VecEnv."""
    import os

    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    return MixedVecEnv(mixed_shard_layout(segments_global, rank, world), device=device, **kw)
