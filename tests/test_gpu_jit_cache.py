"""GPU: the disk cache of the run-time compiled kernels (pcg_abi.hip: jit_kernels) -- ADVICE r2.
  * the key covers the CONTENT of the kernel headers (a header that changes without changing a struct size gets a new
    code object), not only the generated translation unit;
  * the directory is private (0700, owner-only files), a directory others can write is not used;
  * a truncated / edited cache file is detected by its digest and recompiled, never loaded."""
import copy
import ctypes as C
import os
import shutil
import stat
import subprocess
import sys

import numpy as np
import pytest

import scenarios as SC

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import copy, os, sys, ctypes as C
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
import scenarios as SC
import torch
from pcgym_amd import VecEnv, _lib
from pcgym_amd.config import EnvSpec
p = copy.deepcopy(SC.scenarios()["cstr_expr_cons_raw"]["env_params"])
spec = EnvSpec(p)
inc = os.environ.get("TEST_JIT_INC")
if inc:
    import pcgym_amd.config as CFG
    orig = EnvSpec.to_cfg
    def to_cfg(self):
        cfg, keep = orig(self)
        cfg.jit_include_dir = inc.encode()
        keep.append(cfg.jit_include_dir)
        return cfg, keep
    EnvSpec.to_cfg = to_cfg
env = VecEnv(p, n_envs=64, seed=1)
env.reset()
o, r, d, _, _ = env.step(torch.zeros((1, 64), device="cuda", dtype=torch.float64))
torch.cuda.synchronize()
print("OK", float(o.sum()))
'''


def _run(cache, inc=None):
    env = dict(os.environ, PCG_JIT_CACHE=str(cache))
    if inc:
        env["TEST_JIT_INC"] = str(inc)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]
    return r.stdout.strip().splitlines()[-1]


def _files(cache):
    return sorted(f for f in os.listdir(cache) if f.endswith(".pco")) if os.path.isdir(cache) else []


def test_cache_is_private_keyed_by_header_content_and_self_checking(tmp_path):
    cache = tmp_path / "a" / "jit"
    out0 = _run(cache)
    f0 = _files(cache)
    assert len(f0) == 1
    st = os.stat(cache)
    assert stat.S_IMODE(st.st_mode) == 0o700 and stat.S_IMODE(os.stat(cache / f0[0]).st_mode) == 0o600
    # second process: hit (no new file), same result
    assert _run(cache) == out0 and _files(cache) == f0
    # a copy of the kernel headers with one changed comment: same generated translation unit, same struct sizes --
    # and a different key
    tree = tmp_path / "tree"
    shutil.copytree(os.path.join(ROOT, "pc-gym_amd", "csrc"), tree / "pc-gym_amd" / "csrc",
                    ignore=shutil.ignore_patterns("build", "*.o"))
    shutil.copytree(os.path.join(ROOT, "include"), tree / "include")
    inc = tree / "pc-gym_amd" / "csrc"
    assert _run(cache, inc) == out0 and _files(cache) == f0  # identical content: the same object serves
    with open(inc / "pcg_integrators.hpp", "a") as fh:
        fh.write("\n// edited\n")
    assert _run(cache, inc) == out0
    f1 = _files(cache)
    assert len(f1) == 2 and set(f0) < set(f1)
    # truncation and a flipped byte are caught by the digest: recompiled and rewritten, never loaded
    # (a recompiled object is not byte-identical to the first one -- hipRTC embeds run-specific names -- so the check is
    # "replaced by a complete file that the next process accepts unchanged")
    path = cache / f0[0]
    good = path.read_bytes()
    for damage in ("truncate", "flip"):
        cur = path.read_bytes()
        bad = cur[: len(cur) // 2] if damage == "truncate" else cur[:-100] + bytes([cur[-100] ^ 0x40]) + cur[-99:]
        path.write_bytes(bad)
        os.chmod(path, 0o600)
        assert _run(cache) == out0
        fixed = path.read_bytes()
        assert fixed != bad and fixed.startswith(b"PCGJIT2\n") and abs(len(fixed) - len(good)) < 4096, damage
        assert _run(cache) == out0 and path.read_bytes() == fixed, damage


def test_a_directory_others_can_write_is_not_used(tmp_path):
    cache = tmp_path / "shared"
    cache.mkdir()
    os.chmod(cache, 0o777)
    _run(cache)
    assert os.listdir(cache) == []  # compiled and run from memory; nothing read from or written to the shared place
    # a planted file is ignored too
    cache2 = tmp_path / "mine"
    _run(cache2)
    f = _files(cache2)[0]
    os.chmod(cache2 / f, 0o666)
    _run(cache2)  # group/other-writable file: not trusted -> recompiled; the rename replaces it with a private one
    assert stat.S_IMODE(os.stat(cache2 / f).st_mode) == 0o600
