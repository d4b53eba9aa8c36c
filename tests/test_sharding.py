"""CPU: multi-GPU sharding logic + a real world_size-2 gloo run of the N>1 code path
(partition, RNG offsets, host-side gather, timing reduction)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pcgym_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 7, 8, 1 << 20, (1 << 23) + 3):
        for w in (1, 2, 3, 4, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == shard.shard_sizes(n, w)
    with pytest.raises(ValueError):
        shard.shard_range(8, 2, 2)


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["PCG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PCG_ROOT"], "tests", "golden"))
import numpy as np, torch, torch.distributed as dist
import scenarios as SC
from oracle import oracle as O
from pcgym_amd import shard
from pcgym_amd.config import EnvSpec
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
NG, T = 1001, 5                      # ragged global batch
p = SC.scenarios()["cstr_dist_Ti"]["env_params"]
p.update(noise=True, noise_percentage=0.01, gaussian_disturbances={"Ti": 1.0})
spec = EnvSpec(p)
lo, hi = shard.shard_range(NG, rank, world)
# each rank steps ITS slice with the CPU oracle standing in for the device (same buffers, same
# env_offset contract as VecEnv) -- what is tested here is the sharding / gather / RNG-offset logic
env = O.OracleEnv(spec, hi - lo, seed=11, env_offset=lo)
env.reset()
rng = np.random.default_rng(5)
acts = rng.uniform(-1, 1, (T, 1, NG))
rsum = 0.0
for i in range(T):
    env.step(acts[i][:, lo:hi])
    rsum += env.rew.sum()
full_x = shard.gather_to_rank0(torch.tensor(env.x), NG, dim=1)
full_obs = shard.gather_to_rank0(torch.tensor(env.obs), NG, dim=1)
mean_r = shard.reduce_stats(torch.tensor(rsum), torch.tensor(float((hi - lo) * T)))
t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)      # the timing reduction bench.py does
assert abs(t.item() - 0.1 * world) < 1e-12
if rank == 0:
    ref = O.OracleEnv(spec, NG, seed=11, env_offset=0)   # the un-sharded run
    ref.reset()
    rs = 0.0
    for i in range(T):
        ref.step(acts[i]); rs += ref.rew.sum()
    assert full_x.shape == (2, NG) and np.array_equal(full_x.numpy(), ref.x), "sharded state != single-device state"
    assert np.array_equal(full_obs.numpy(), ref.obs), "sharded obs (incl. noise + Gaussian disturbance) differ"
    assert abs(mean_r - rs / (NG * T)) < 1e-9 * abs(mean_r)
    print("SHARD_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_world_size_2_gloo_sharded_run_equals_single_device_run(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PCG_ROOT=ROOT, OMP_NUM_THREADS="1")
    import socket

    with socket.socket() as sk:  # any free port: a fixed one can still be held by an earlier run
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SHARD_OK" in r.stdout


def test_mixed_shard_layout_partitions_every_segment_and_keeps_global_offsets():
    """BASELINE configs[4] sharding (pcgym_amd.mixed_shard_layout): every rank gets the same fraction of every
    model segment, offsets follow the global layout [segment 0 | segment 1 | ...]; stepping the shards with the oracle
    standing in for the device (same env_offset contract) reproduces the unsharded mixed batch bit for bit, Gaussian
    disturbance streams included."""
    import copy

    import bench as BN
    from oracle import oracle as O
    from pcgym_amd import mixed_shard_layout
    from pcgym_amd.config import EnvSpec

    sizes = [1001, 700, 333]
    segs = [(p, n) for (p, _), n in zip(BN.mixed_segments(6), sizes)]
    for world in (1, 2, 3, 8):
        lay = [mixed_shard_layout(segs, r, world) for r in range(world)]
        for k, n in enumerate(sizes):
            base = sum(sizes[:k])
            spans = sorted((l[k][2], l[k][2] + l[k][1]) for l in lay)
            assert spans[0][0] == base and spans[-1][1] == base + n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(s[1] - s[0] for s in spans) - min(s[1] - s[0] for s in spans) <= 1
    rng = np.random.default_rng(3)
    T = 3
    world = 2
    lay = [mixed_shard_layout(segs, r, world) for r in range(world)]
    for k, (p, n) in enumerate(segs):
        spec = EnvSpec(copy.deepcopy(p))
        base = sum(sizes[:k])
        full = O.OracleEnv(spec, n, seed=9, env_offset=base)
        full.reset()
        parts = [O.OracleEnv(spec, l[k][1], seed=9, env_offset=l[k][2]) for l in lay]
        for e in parts:
            e.reset()
        for i in range(T):
            a = rng.uniform(-0.5, 1, (spec.na, n))
            full.step(a)
            for e, l in zip(parts, lay):
                lo = l[k][2] - base
                e.step(a[:, lo:lo + l[k][1]])
        assert np.array_equal(np.concatenate([e.x for e in parts], axis=1), full.x), spec.model.name
        assert np.array_equal(np.concatenate([e.obs for e in parts], axis=1), full.obs), spec.model.name
