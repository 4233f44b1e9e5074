"""Prints the kernel timeline of single-frame host calls from a rocprofv3 kernel trace (tools/prof_latency.sh): the
calls number -5, -4, -3 from the end (one line per kernel: start relative to the call's first kernel, duration)."""
import csv, glob, os, sys
f = max(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_lat") + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = "k_pyr_copy0" if any("k_pyr_copy0" in r["Kernel_Name"] for r in rows) else "k_pyr_fused"  # a call's first kernel
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
for call in (-5, -4, -3):
    i0 = idx[call]
    t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0:idx[call + 1]]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-28s start %7.1f us  dur %6.1f us  queue %s" % (r["Kernel_Name"].split("(")[0][-26:], (s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "")))
    print()
