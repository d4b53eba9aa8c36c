#!/usr/bin/env python3
"""Condense the PMC passes of tools/prof_all.sh into one JSON (committed as profiles/r3/pmc.json, read by bench.py):
per workload the dominant step kernel's mean counters per launch, HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE: the gfx950
FETCH_SIZE correction of /opt/skills/guides/MI355X_MICROARCH.md "HBM"), and the VALU issue fraction
4 SQ_INSTS_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32).  Every entry carries the build id of the library the passes ran on
(pcg_build_id(): digest of the kernel headers and the .hip units), and bench.py quotes an entry only for that build.
usage: pmc_json.py <gpurun_out> <workload ...>      PMC_ROUND=r4 names the profiles/ directory in the source note"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, wls = sys.argv[1], sys.argv[2:]
RND = os.environ.get("PMC_ROUND", "r6")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from pcgym_amd import _lib

    BUILD_ID = _lib.load().pcg_build_id().decode()
except Exception as e:  # noqa: BLE001
    BUILD_ID = f"unknown ({type(e).__name__})"
SEG = {"Model<0>": "cstr", "Model<1>": "four_tank", "Model<2>": "multistage_extraction", "Model<18>": "multistage_extraction",
       "Model<3>": "multistage_extraction_reactive", "Model<19>": "multistage_extraction_reactive", "Model<4>": "crystallization"}


def counters(d):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if "step_kernel" in k or "rollout_kernel" in k:
                    agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"_n": len(next(iter(cs.values())))} for k, cs in agg.items()}


def durations(d):
    out = {}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                out[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]))
    return out


def entry(kname, c, dur, wl):
    e = {"kernel": kname[:120], "launches_in_pmc_pass": c["_n"]}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = c["FETCH_SIZE"] * 1024 * 2.0, c["WRITE_SIZE"] * 1024
        e.update(FETCH_SIZE_KiB=c["FETCH_SIZE"], WRITE_SIZE_KiB=c["WRITE_SIZE"], gfx950_fetch_correction=2.0,
                 read_bytes=rd, write_bytes=wr, traffic_bytes_per_launch=rd + wr)
    if "SQ_INSTS_VALU" in c:
        e["SQ_INSTS_VALU_per_launch"] = c["SQ_INSTS_VALU"]
        for k in ("SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if k in c:
                e[k + "_per_launch"] = c[k]
    if "GRBM_GUI_ACTIVE" in c:
        e["GRBM_GUI_ACTIVE_per_launch"] = c["GRBM_GUI_ACTIVE"]  # (includes dispatch overhead: not used as the cycle count)
    if kname in dur:
        e["rocprof_avg_us"] = dur[kname][1] / 1e3
        e["rocprof_calls"] = dur[kname][0]
    if "SQ_BUSY_CYCLES" in c and "SQ_INSTS_VALU" in c and c["SQ_BUSY_CYCLES"] > 0:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4): /32 = busy cycles of the launch -- calibrated
        # against the kernel duration it gives 2.05-2.15 GHz on every workload.  A wave64 VALU instruction occupies its
        # 16-lane SIMD for 4 cycles; 1024 SIMDs.
        cyc = c["SQ_BUSY_CYCLES"] / 32.0
        e["busy_cycles_per_launch"] = cyc
        e["valu_issue_frac"] = 4.0 * c["SQ_INSTS_VALU"] / (1024.0 * cyc)
        if kname in dur:
            e["sq_clock_GHz"] = cyc / dur[kname][1]
            # the same instruction count priced at the MEASURED issue cost of an fp64 wave-instruction (tools/issuebench.hip:
            # 2.1-2.5 ns per SIMD, i.e. ~5 cycles, not 4): the share of the launch its SIMDs spend issuing vector work
            e["valu_issue_time_frac_at_2p4ns"] = c["SQ_INSTS_VALU"] * 2.4 / 1024.0 / dur[kname][1]
            # ... and priced BY CLASS in CYCLES against the launch's own busy cycles (round 5).  The flat 2.4 ns of round 4 read
            # 1.07 for me10 and 1.09 for cryst: tools/issuebench.hip's 2.4 ns (~5 cycles at its ~2.0 GHz) is the cost of an fp64
            # instruction in a DEPENDENT chain of one wave, not its issue slot -- a SIMD issues an fp64 wave-instruction in 4 cycles
            # (16 lanes x 4), and these kernels run at 2.1-2.25 GHz.  Costs: fp64 add / mul / fma 4 cycles (architectural), fp64
            # estimates (rcp / rsq) 16 (quarter rate; measured 17), fp32 transcendentals 8 (measured 8.6), everything else
            # (32-bit moves, selects, integer, conversions) 3 (measured: v_mov_b32 / v_and_b32 1.35 ns at 2.26 GHz).
            if all(k in c for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")):
                f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
                t64, t32 = c["SQ_INSTS_VALU_TRANS_F64"], c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
                rest = max(0.0, c["SQ_INSTS_VALU"] - f64 - t64 - t32)
                e["valu_insts_by_class_per_launch"] = {"fp64_add_mul_fma": f64, "fp64_trans": t64, "fp32_trans": t32, "other": rest}
                e["valu_issue_cycles_by_class"] = {"fp64_add_mul_fma": 4, "fp64_trans": 16, "fp32_trans": 8, "other": 3}
                e["valu_issue_time_frac_by_class"] = (f64 * 4.0 + t64 * 16.0 + t32 * 8.0 + rest * 3.0) / (1024.0 * cyc)
    e["build_id"] = BUILD_ID
    e["source"] = (f"profiles/{RND}/{wl}/rocprofv3_summary.txt (tools/prof_all.sh: FETCH_SIZE, WRITE_SIZE, SQ and GRBM counters in "
                   "separate --pmc passes with --kernel-trace only; FETCH_SIZE x 2 per MI355X_MICROARCH.md HBM section); NOT "
                   "measured in the bench run itself")
    return e


res = {}
for wl in wls:
    d = os.path.join(root, "prof_" + wl)
    cs, dur = counters(d), durations(d)
    if not cs:
        continue
    if wl == "mixed":
        segs = {}
        for k, c in cs.items():
            name = next((v for q, v in SEG.items() if q in k), None)
            if name and (name not in segs or c["_n"] > segs[name]["launches_in_pmc_pass"]):
                segs[name] = entry(k, c, dur, wl)
        res[wl] = {"segments": segs, "build_id": BUILD_ID}
    elif wl in ("cstr_safe", "cstr_safe_rollout"):
        # (cstr_safe_rollout: an EPISODE is two launches -- the barrier-free rollout's two passes, pcg_rollout_flat.hpp)
        # a guarded plan's step is TWO launches (the guarded step of every env + the work-queue fix-up of the envs it marked):
        # per-step counters = the sum over both kernels' per-launch means; the duration is the sum of the two averages
        ks = [q for q in cs if cs[q]["_n"] >= 0.5 * max(v["_n"] for v in cs.values())]
        tot = defaultdict(float)
        for q in ks:
            for cn, v in cs[q].items():
                if cn != "_n":
                    tot[cn] += v
        tot["_n"] = min(cs[q]["_n"] for q in ks)
        name = " + ".join(q[:60] for q in ks)
        d2 = dict(dur)
        if all(q in dur for q in ks):
            d2[name] = (min(dur[q][0] for q in ks), sum(dur[q][1] for q in ks))
        res[wl] = entry(name, dict(tot), d2, wl)
        res[wl]["kernels_per_step"] = len(ks)
    else:
        k = max(cs, key=lambda q: cs[q]["_n"])  # the kernel of (almost) every launch of the workload
        res[wl] = entry(k, cs[k], dur, wl)
print(json.dumps(res, indent=1))
