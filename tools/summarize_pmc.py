#!/usr/bin/env python3
"""Summarises gpurun_out/<tag>/ (tools/profile.sh) into profiles/<tag>_*.{csv,json,md}.

HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB-ish units of
the fabric request counters; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read, so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 reads exactly for wide streaming stores.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "dam1m"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n


stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
kstats = {}
if stats:
    shutil.copy(stats[0], os.path.join("profiles", tag + "_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        kstats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}

counters = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [values per dispatch]
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))

summary = {}
for k, cs in counters.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["dispatches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes"] = 2.0 * d["FETCH_SIZE"] * 1024.0  # gfx950: FETCH_SIZE counts 64 B per 128-B request
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024.0
    if "hbm_read_bytes" in d and "hbm_write_bytes" in d:
        d["hbm_bytes"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
    if "SQ_ACTIVE_INST_VALU" in d and "SQ_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; 256 CUs x 4 SIMDs issue slots
        d["valu_busy_frac"] = min(1.0, d["SQ_ACTIVE_INST_VALU"] * 4.0 / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))  # (GRBM clock: an upper estimate)
    if k in kstats and d.get("SQ_INSTS_VALU"):
        # issue rate: shader cycles (2.1 GHz under this load, s_memtime against s_memrealtime) per VALU wave-instruction
        # and SIMD (1024 SIMDs).  fp32 fma / mul / add issue every 2 cycles at >= 4 waves, most other VALU ops every
        # 4, rcp / rsq every 8 (tools/ubench/valu_issue.hip), so 3-3.5 means the VALU pipe is the limiter.
        d["valu_cycles_per_wave_instr"] = kstats[k]["avg_ns"] * 1e-9 * 2.1e9 * 1024.0 / d["SQ_INSTS_VALU"]
    if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
        d["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64.0)
    if k in kstats:
        d["avg_ns"] = kstats[k]["avg_ns"]
    summary[k] = d

json.dump(summary, open(os.path.join("profiles", tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)

tf = os.path.join("profiles", "traffic.json")
traffic = json.load(open(tf)) if os.path.exists(tf) else {}
for k, d in summary.items():
    if k.startswith("k_force<true") and "hbm_bytes" in d:
        traffic.setdefault(workload, {})["k_force_bytes_per_launch"] = d["hbm_bytes"]
        traffic[workload]["source"] = "profiles/%s_pmc_summary.json" % tag
json.dump(traffic, open(tf, "w"), indent=1, sort_keys=True)

with open(os.path.join("profiles", tag + "_summary.md"), "w") as f:
    f.write("# rocprofv3 summary `%s` (workload %s)\n\n" % (tag, workload))
    f.write("| kernel | calls | avg us | % | HBM read MB | HBM write MB | cycles / VALU wave-instr | lane util | LDS bank conflict/idx |\n|---|---|---|---|---|---|---|---|---|\n")
    for k, s in sorted(kstats.items(), key=lambda kv: -kv[1]["pct"]):
        d = summary.get(k, {})
        fm = lambda v, sc=1.0, fmt="%.2f": (fmt % (v * sc)) if v is not None else "-"  # noqa: E731
        conf = None
        if d.get("SQ_LDS_IDX_ACTIVE"):
            conf = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        f.write("| %s | %d | %.1f | %.2f | %s | %s | %s | %s | %s |\n" % (
            k, s["calls"], s["avg_ns"] / 1e3, s["pct"], fm(d.get("hbm_read_bytes"), 1e-6), fm(d.get("hbm_write_bytes"), 1e-6),
            fm(d.get("valu_cycles_per_wave_instr")), fm(d.get("valu_lane_utilisation")), fm(conf)))
print(open(os.path.join("profiles", tag + "_summary.md")).read())
