#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>cli (tools/profile_round3.sh c: exe/cuCLARK-l under rocprofv3) into
profiles/<tag>_cli_kernel_stats.csv (the --stats table, verbatim) and profiles/<tag>_cli_ingest_kernels.json (per kernel:
calls, time, counters summed over the run's launches).    python tools/summarize_cli_profile.py r03"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join("gpurun_out", f"prof_{tag}cli")
stats = sorted(glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(stats, os.path.join("profiles", f"{tag}_cli_kernel_stats.csv"))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*\)$", "", name)


kernels = collections.OrderedDict()
for r in csv.DictReader(open(stats)):
    kernels[short(r["Name"])] = {"calls": int(r["Calls"]), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3),
                                 "avg_us": round(float(r["AverageNs"]) / 1e3, 1)}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        agg[(short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, c), v in agg.items():
        if k in kernels:
            kernels[k][c + "_total"] = v
out_txt = open(os.path.join(src, "kt.out")).read() + open(os.path.join(src, "kt.err")).read()
m = re.search(r"Assignment time:\s*([0-9.]+)", out_txt)
inp = int(open(os.path.join(src, "input_bytes.txt")).read().split()[0])
doc = {"what": "exe/cuCLARK-l -O reads.fq -n 12 under rocprofv3 (tools/profile_round3.sh c): 16 M x 150 bp FASTQ "
               f"({inp / 1e9:.2f} GB), light table (60 M k-mers); kernel trace + PMC passes (FETCH_SIZE / WRITE_SIZE in KB, summed over "
               "the run's launches)",
       "assignment_s": float(m.group(1)) if m else None, "kernels": kernels}
json.dump(doc, open(os.path.join("profiles", f"{tag}_cli_ingest_kernels.json"), "w"), indent=1)
for k, v in list(kernels.items())[:12]:
    print(k, v["calls"], v["total_ms"])
