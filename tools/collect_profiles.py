#!/usr/bin/env python3
"""Copy the profile set tools/prof_all.sh left under gpurun_out/ into profiles/<round>/ (summaries only: kernel stats,
PMC means, the bench line of the traced run, pmc.json) and, if given, the bench lines of a session directory.
usage: collect_profiles.py <round> [gpurun_out/<session>]"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
sess = sys.argv[2] if len(sys.argv) > 2 else None
dst = os.path.join(ROOT, "profiles", rnd)


def last(f):
    lines = [x for x in open(f) if x.startswith("{")]
    return lines[-1] if lines else None


for w in "cstr cstr_safe cstr_rollout cstr_safe_rollout cstr_unc four_tank me10 me10_ros4 me10_ros5 me20 cryst cryst_cv8 mixed".split():
    src = os.path.join(ROOT, "gpurun_out", "prof_" + w)
    if not os.path.exists(os.path.join(src, "summary.txt")):
        continue
    d = os.path.join(dst, w)
    os.makedirs(d, exist_ok=True)
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(d, "rocprofv3_summary.txt"))
    bt = os.path.join(src, "bench_trace.json")
    if os.path.exists(bt) and last(bt):
        open(os.path.join(d, "bench_under_rocprof.json"), "w").write(last(bt))
    ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(d, "kernel_stats.csv"))
pm = os.path.join(ROOT, "gpurun_out", "pmc.json")
if os.path.exists(pm):
    shutil.copy(pm, os.path.join(dst, "pmc.json"))
if sess:
    s = os.path.join(ROOT, sess)

    def put(name, files):
        files = [f for f in files if os.path.exists(f) and last(f)]
        if files:
            with open(os.path.join(dst, name), "w") as fh:
                for f in files:
                    fh.write(last(f))

    put("bench_default.json", [os.path.join(s, "bench_default.json")])
    put("bench_runs.jsonl", [os.path.join(s, f"bench_default_run{i}.json") for i in (1, 2, 3)])
    put("bench_driver_shape_runs.jsonl", [os.path.join(s, f"bench_driver_shape_run{i}.json") for i in (1, 2, 3)])
    put("bench_workloads.jsonl", [os.path.join(s, f"bench_{w}.json") for w in
                                  "cstr_safe four_tank four_tank_rk4 me10 me10_ros4 me10_ros5 me20 cryst cryst_cv8 mixed".split()])
    put("bench_graph.json", [os.path.join(s, "bench_graph.json")])
    if os.path.exists(os.path.join(s, "bench_all.txt")):
        shutil.copy(os.path.join(s, "bench_all.txt"), os.path.join(dst, "bench_all.txt"))
print("collected into", dst)
