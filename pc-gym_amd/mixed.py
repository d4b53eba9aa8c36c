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

    def __init__(self, segments, device=None, seed=0, env_offset=0, **kw):
        torch = _torch()
        self.envs, self.offsets = [], []
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

    def __len__(self):
        return len(self.envs)

    def _each(self, fn):
        """Run fn(i, env) for every segment on that segment's stream; the caller's stream waits for all of them
        (so results can be consumed on it without a host synchronisation)."""
        torch = _torch()
        cur = torch.cuda.current_stream(self.device)
        out = []
        for i, (e, s) in enumerate(zip(self.envs, self.streams)):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                out.append(fn(i, e))
        for s in self.streams:
            cur.wait_stream(s)
        return out

    def reset(self, seed=None):
        return self._each(lambda i, e: e.reset(seed=seed))

    def step(self, actions, disturbances=None):
        """actions: one (na_i, B_i) SoA tensor (or (B_i, na_i)) per segment -> list of step() tuples."""
        if len(actions) != len(self.envs):
            raise ValueError(f"one action tensor per segment ({len(self.envs)}) is required")
        d = disturbances or [None] * len(self.envs)
        return self._each(lambda i, e: e.step(actions[i], d[i]))

    @property
    def bytes_per_step(self):
        return sum(e.bytes_per_env_step * e.B for e in self.envs)

    def close(self):
        for e in self.envs:
            e.close()


def make_mixed_sharded_env(segments_global, rank=None, world=None, device=None, **kw):
    """Every rank takes the same fraction of each global segment (so every shard holds the same model mix and the
    per-GPU work is balanced); env offsets follow the global layout [segment 0 | segment 1 | ...]."""
    import os

    from .shard import shard_range

    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    local, base = [], 0
    for params, n in segments_global:
        lo, hi = shard_range(int(n), rank, world)
        local.append((params, hi - lo, base + lo))
        base += int(n)
    return MixedVecEnv(local, device=device, **kw)
