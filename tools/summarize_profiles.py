#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into
kernel_stats.csv and pmc_traffic.json (per kernel, per launch, bytes)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def kname(full):
    """`void orbgpu::k_fast_cells<48>(args...)` -> `k_fast_cells`."""
    return re.sub(r"<.*>", "", full.split("(")[0]).replace("void ", "").replace("orbgpu::", "").strip()


src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
shutil.copy(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0], os.path.join(dst, "kernel_stats.csv"))
traffic = {}
for tag, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(src, tag, "*", "*_counter_collection.csv"))[0])):
        if "orbgpu" in r["Kernel_Name"] and r["Counter_Name"] == name:
            acc[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        traffic.setdefault(k, {})[name + "_KB_per_launch"] = sum(v) / len(v)
        traffic[k]["launches_" + tag] = len(v)
json.dump({"batch": 512,
           "command": "python3 bench.py --batch 512 --sub-batches 1 --streams 1 --steps 3 --warmup 1 --no-cpu --no-latency --no-configs (one rocprofv3 --pmc pass "
                      "per counter; one stream so that dispatches do not overlap; 512 frames per launch as in the default run)",
           "unit": "rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; hbm_bytes = (FETCH*corr + WRITE) * 1024",
           "calibration": "k_pyr_copy0 (16 B/lane loads of a known byte count) reads 2.1x its FETCH_SIZE -> corr 2 for "
                          "128-bit streams; k_blur (4 B/lane loads, known volume) reads ~1.1x -> corr 1 for 32-bit loads",
           "kernels": traffic}, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for r in csv.DictReader(open(os.path.join(dst, "kernel_stats.csv"))):
    if "orbgpu" in r["Name"]:
        print("%-24s calls %4s avg_us %9.1f" % (kname(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3))
