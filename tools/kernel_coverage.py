#!/usr/bin/env python3
"""profiles/rN/kernel_coverage.txt: which of the kernels libpcgym_hip.so carries the GPU suite launched, and against what.

    python tools/kernel_coverage.py [gpurun_out/kernel_coverage.json ...] > profiles/r6/kernel_coverage.txt

Input: the record tests/conftest.py writes at the end of a `pytest -m gpu` session (PCG_COVERAGE: the library notes every
launch, the conftest attributes it to the running test and classes the test as oracle / golden fixture / other);
tools/kernel_inventory.py supplies the list of shipped kernels."""
import fnmatch
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import kernel_inventory as KI  # noqa: E402


def main():
    paths = sys.argv[1:] or [os.path.join(ROOT, "gpurun_out", "kernel_coverage.json")]
    cov, state = {}, {}
    for path in paths:  # several records (sessions that ran parts of the suite) are merged
        rec = json.load(open(path))
        for k, v in rec["state"].items():
            state[k] = state.get(k, 0) + v
        for name, d in rec["kernels"].items():
            c = cov.setdefault(name, {kind: {"n": 0, "tests": []} for kind in ("oracle", "golden", "other")})
            for kind in c:
                c[kind]["n"] += d[kind]["n"]
                c[kind]["tests"] = (c[kind]["tests"] + d[kind]["tests"])[:4]
    ks = KI.inventory()
    allow = []
    with open(os.path.join(ROOT, "tests", "kernel_coverage_allow.txt")) as f:
        for line in f:
            if line.strip() and not line.lstrip().startswith("#"):
                pat, _, why = line.rstrip("\n").partition("  # ")
                allow.append((pat.strip(), why.strip()))
    print(f"# kernel-instantiation coverage of `pytest tests -m gpu` ({state['gpu_ran']} GPU tests ran, {state['gpu_failed']} failed, "
          f"{state['gpu_deselected']} deselected)")
    print(f"# shipped kernels: {len(ks)} (tools/kernel_inventory.py: the code objects of pc-gym_amd/libpcgym_hip.so)")
    fams = {}
    for k in ks:
        c = cov.get(k["name"])
        k["cls"] = ("oracle" if c and c["oracle"]["n"] else "golden" if c and c["golden"]["n"] else
                    "other" if c else "never")
        fams.setdefault(k["family"], []).append(k)
    print(f"{'family':34s} {'shipped':>8s} {'vs oracle':>10s} {'vs fixture only':>16s} {'launched, self-consistency only':>32s} {'never launched':>15s}")
    tot = dict(oracle=0, golden=0, other=0, never=0)
    for fam, v in sorted(fams.items(), key=lambda kv: -len(kv[1])):
        n = {c: sum(1 for k in v if k["cls"] == c) for c in tot}
        for c in tot:
            tot[c] += n[c]
        print(f"{fam:34s} {len(v):8d} {n['oracle']:10d} {n['golden']:16d} {n['other']:32d} {n['never']:15d}")
    print(f"{'total':34s} {len(ks):8d} {tot['oracle']:10d} {tot['golden']:16d} {tot['other']:32d} {tot['never']:15d}")
    jit = sorted(n for n in cov if n.startswith("jit:"))
    print(f"\n# run-time compiled kernels launched (hipRTC: user expressions / user models; not in the library): {len(jit)}")
    for n in jit:
        c = cov[n]
        print(f"  {n}  oracle {c['oracle']['n']} golden {c['golden']['n']} other {c['other']['n']}")
    for cls, title in (("never", "never launched"), ("other", "launched, but by no test that checks against the oracle or a fixture")):
        rows = [k for k in ks if k["cls"] == cls]
        print(f"\n# {title}: {len(rows)}")
        for k in rows:
            why = next((w for p, w in allow if fnmatch.fnmatchcase(k["demangled"], p)), None)
            tests = ""
            if cls == "other":
                tests = "   <- " + ", ".join(cov[k["name"]]["other"]["tests"][:2])
            print(f"  {k['demangled']}{tests}" + (f"   [allowed: {why}]" if why else "   [NOT ALLOWED]"))
    # the compact map build() and tools/kernel_inventory.py --scratch print beside each spilling kernel
    tests_out = os.environ.get("KERNEL_TESTS_JSON")
    if tests_out:
        with open(tests_out, "w") as f:
            json.dump({k["name"]: (cov[k["name"]]["oracle"]["tests"] or cov[k["name"]]["golden"]["tests"])[0]
                       for k in ks if k["cls"] in ("oracle", "golden")}, f, indent=0, sort_keys=True)
    print("\n# every checked kernel, with the number of passing tests that launched it (oracle / fixture / other) and one of them")
    for k in ks:
        if k["cls"] in ("oracle", "golden"):
            c = cov[k["name"]]
            t = (c["oracle"]["tests"] or c["golden"]["tests"])[0]
            print(f"  {c['oracle']['n']:4d} {c['golden']['n']:4d} {c['other']['n']:4d}  {k['demangled']}   {t}")


if __name__ == "__main__":
    main()
