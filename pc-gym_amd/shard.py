"""Multi-GPU sharding of the environment batch.

The envs of a batch are independent (no cross-env term in any RHS, reward or
constraint), so the batch is partitioned into contiguous ranges, one per GPU /
process, and every rank steps its own range with its own plan and stream.
There is NO collective on the hot path.  torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm, "gloo" on CPU in the tests) is used only

  * for the barrier / max-reduction bench.py needs for timing, and
  * to gather per-rank trajectory slices or reward statistics to rank 0
    ("host-side trajectory gather only", BASELINE.json north_star).

The global env index (rank offset + local index) keys the counter-based RNG, so
a sharded run draws exactly the same random numbers as a single-device run of
the whole batch.
"""
from __future__ import annotations


def shard_range(n_envs: int, rank: int, world: int):
    """Contiguous [lo, hi) of rank `rank`: sizes differ by at most one, earlier ranks get the extras."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if n_envs < 0:
        raise ValueError("n_envs must be >= 0")
    base, extra = divmod(n_envs, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_sizes(n_envs: int, world: int):
    return [shard_range(n_envs, r, world)[1] - shard_range(n_envs, r, world)[0] for r in range(world)]


def make_sharded_env(env_params, n_envs_global, rank=None, world=None, device=None, **kw):
    """VecEnv over this rank's slice of a global batch (RANK / WORLD_SIZE / LOCAL_RANK from the
    environment when not given).  kw is forwarded to VecEnv."""
    import os

    from .env import VecEnv

    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    lo, hi = shard_range(n_envs_global, rank, world)
    return VecEnv(env_params, n_envs=hi - lo, device=device, env_offset=lo, **kw)


def gather_to_rank0(t, n_envs_global, dim=-1, group=None):
    """Gather the per-rank slices of a batched tensor (env axis = `dim`) onto rank 0, in global env
    order.  Works with any backend; returns the concatenated tensor on rank 0 and None elsewhere.
    Uses all_gather on equal-sized padded buffers (ranks may differ by one env)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return t
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_envs_global, world)
    t = t.movedim(dim, 0).contiguous()
    pad = max(sizes)
    buf = t.new_zeros((pad,) + tuple(t.shape[1:]))
    buf[: t.shape[0]] = t
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    if rank != 0:
        return None
    full = torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)
    return full.movedim(0, dim)


def reduce_stats(local_sum, local_count, group=None):
    """Global mean of a per-env quantity from per-rank (sum, count): the only 'collective' a training
    loop typically wants from the env side (e.g. mean episode reward)."""
    import torch
    import torch.distributed as dist

    v = torch.stack([local_sum.reshape(()).to(torch.float64), local_count.reshape(()).to(torch.float64)])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return (v[0] / torch.clamp(v[1], min=1.0)).item()
