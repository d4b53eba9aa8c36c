#!/usr/bin/env python3
"""Cross-check of the in-library launch record (pcg_coverage_names -> gpurun_out/kernel_coverage.json) against rocprofv3:
the distinct kernel names of `rocprofv3 --kernel-trace --stats -- python -m pytest tests -m gpu` (every process of the run,
the subprocesses of the bench / C-host tests included) must be the kernels the library's own record lists, up to the
run-time compiled ones.   usage: coverage_crosscheck.py <dir with *kernel_stats.csv> <kernel_coverage.json>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_inventory as KI

stats_dir, cov_json = sys.argv[1], sys.argv[2]
ks = KI.inventory()
by_dem = {k["demangled"]: k["name"] for k in ks}
prof = {}
for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        prof[r["Name"]] = prof.get(r["Name"], 0) + int(r["Calls"])
ours = {n: c for n, c in prof.items() if "pcg::" in n or "sort_tile_test_kernel" in n}
lib_named = {n for n in ours if n in by_dem}
rec = set(json.load(open(cov_json))["kernels"])
rec_lib = {n for n in rec if not n.startswith("jit:")}
prof_mangled = {by_dem[n] for n in lib_named}
print(f"# rocprofv3 --kernel-trace --stats over the whole GPU suite: {len(prof)} distinct kernel names, {len(ours)} of them this "
      f"library's or its run-time compiled modules' ({sum(ours.values())} launches), {len(lib_named)} match a shipped kernel by name")
print(f"# in-library record of the same kind of run: {len(rec_lib)} shipped kernels launched (+ {len(rec) - len(rec_lib)} run-time compiled)")
print(f"shipped kernels: {len(ks)}; launched per rocprofv3: {len(prof_mangled)}; launched per the library's record: {len(rec_lib)}")
print("in rocprofv3's list but not in the library's record:", sorted(prof_mangled - rec_lib)[:10])
print("in the library's record but not in rocprofv3's list:", sorted(rec_lib - prof_mangled)[:10])
print("shipped but in neither:", sorted({k['name'] for k in ks} - rec_lib - prof_mangled)[:10])
