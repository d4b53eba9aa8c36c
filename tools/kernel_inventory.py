#!/usr/bin/env python3
"""Every gfx950 kernel libpcgym_hip.so carries, with the resources the compiler gave it.

Pure Python (no llvm tools needed on the GPU box): the host ELF's `.hip_fatbin` section holds one clang offload bundle per
translation unit; each bundle's `hipv4-amdgcn-amd-amdhsa--gfx950` entry is a device ELF whose NT_AMDGPU_METADATA note
(msgpack) lists the kernels with their register / scratch / LDS figures.

    python tools/kernel_inventory.py                    # table: family, count, worst registers / scratch
    python tools/kernel_inventory.py --json out.json    # every kernel: mangled name, demangled name, resources
    python tools/kernel_inventory.py --scratch          # the kernels with scratch > 0 (spills), largest first

`tests/test_kernel_coverage.py` and `tools/kernel_coverage.py` read the same list to say which instantiations the GPU
suite launched.
"""
import argparse
import json
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pc-gym_amd", "libpcgym_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(buf):
    """(name, offset, size, type) of every section of a 64-bit little-endian ELF held in `buf`."""
    assert buf[:4] == b"\x7fELF" and buf[4] == 2 and buf[5] == 1, "not a 64-bit LE ELF"
    shoff, = struct.unpack_from("<Q", buf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", buf, 0x3A)
    raw = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", buf, shoff + i * shentsize)
        raw.append((name, off, size, typ))
    stroff = raw[shstrndx][1]
    out = []
    for name, off, size, typ in raw:
        end = buf.index(b"\0", stroff + name)
        out.append((buf[stroff + name:end].decode(), off, size, typ))
    return out


def device_objects(path=LIB, arch="gfx950"):
    """The device ELFs (bytes) of every bundle in the library's .hip_fatbin."""
    with open(path, "rb") as f:
        host = f.read()
    fat = [s for s in _sections(host) if s[0] == ".hip_fatbin"]
    assert fat, "no .hip_fatbin section in " + path
    _, off, size, _ = fat[0]
    blob = host[off:off + size]
    objs = []
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tsize].decode()
            p += 24 + tsize
            if arch in triple and esize:
                objs.append(blob[pos + eoff:pos + eoff + esize])
        pos += len(MAGIC)
    return objs


def _kernels_of(obj):
    import msgpack

    out = []
    for name, off, size, typ in _sections(obj):
        if typ != 7:  # SHT_NOTE
            continue
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", obj, p)
            p += 12
            nm = obj[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = obj[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and nm.startswith(b"AMDGPU"):
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out.append({
                        "name": k[".name"],
                        "vgpr": k.get(".vgpr_count", 0),
                        "agpr": k.get(".agpr_count", 0),
                        "sgpr": k.get(".sgpr_count", 0),
                        "sgpr_spill": k.get(".sgpr_spill_count", 0),
                        "vgpr_spill": k.get(".vgpr_spill_count", 0),
                        "scratch": k.get(".private_segment_fixed_size", 0),
                        "dyn_stack": bool(k.get(".uses_dynamic_stack", False)),
                        "lds": k.get(".group_segment_fixed_size", 0),
                        "max_wg": k.get(".max_flat_workgroup_size", 0),
                    })
    return out


def demangle(names):
    """c++filt in one batch (binutils or llvm-cxxfilt, whichever the box has); identity when neither exists."""
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True)
            out = r.stdout.split("\n")
            if len(out) >= len(names):
                return out[:len(names)]
        except (OSError, subprocess.CalledProcessError):
            continue
    return list(names)


def family(dem):
    """template family of a demangled kernel name: 'pcg::step_kernel_queue' of 'void pcg::step_kernel_queue<...>(...)'."""
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)", dem)
    return m.group(1) if m else dem


def inventory(path=LIB):
    """Every kernel of the library once (a template instantiated in two units is linkonce: the loader keeps one)."""
    seen = {}
    for obj in device_objects(path):
        for k in _kernels_of(obj):
            seen.setdefault(k["name"], k)
    ks = sorted(seen.values(), key=lambda k: k["name"])
    for k, d in zip(ks, demangle([k["name"] for k in ks])):
        k["demangled"] = d
        k["family"] = family(d)
    return ks


def covering_tests():
    """mangled kernel name -> one passing test of the GPU suite that launched it against the oracle / a fixture (the record
    tools/kernel_coverage.py condensed from the last full GPU run, committed under profiles/)"""
    for rnd in ("r6",):
        p = os.path.join(ROOT, "profiles", rnd, "kernel_tests.json")
        if os.path.exists(p):
            with open(p) as f:
                return json.load(f)
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--json")
    ap.add_argument("--scratch", action="store_true", help="list every kernel with scratch > 0")
    a = ap.parse_args()
    ks = inventory(a.lib)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(ks, f, indent=0)
    fams = {}
    for k in ks:
        fams.setdefault(k["family"], []).append(k)
    print(f"{len(ks)} kernels in {os.path.relpath(a.lib, ROOT)}")
    # (.vgpr_count of the metadata is the unified register file's allocation, accumulation registers included: <= 512)
    print(f"{'family':42s} {'n':>5s} {'max vgprs':>10s} {'n scratch>0':>12s} {'max scratch B':>14s} {'max LDS B':>10s}")
    for fam, v in sorted(fams.items(), key=lambda kv: -len(kv[1])):
        print(f"{fam:42s} {len(v):5d} {max(k['vgpr'] for k in v):10d} "
              f"{sum(1 for k in v if k['scratch'] > 0):12d} {max(k['scratch'] for k in v):14d} {max(k['lds'] for k in v):10d}")
    if a.scratch:
        tests = covering_tests()
        print("\nkernels with scratch > 0: bytes per lane; registers (of them accumulation); scalar / vector spills; the kernel; "
              "a passing oracle test that launches it (profiles/r6/kernel_tests.json)")
        for k in sorted((k for k in ks if k["scratch"] > 0), key=lambda k: -k["scratch"]):
            print(f"{k['scratch']:7d} {k['vgpr']:4d} ({k['agpr']:3d}) {k['sgpr_spill']:5d}/{k['vgpr_spill']:<5d} {k['demangled']}   <- "
                  f"{tests.get(k['name'], 'NO COVERING TEST RECORDED')}")


if __name__ == "__main__":
    sys.exit(main())
