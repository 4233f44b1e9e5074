"""Prints the kernel timeline of one single-frame host call from a rocprofv3 kernel trace (tools/prof_latency.sh)."""
import csv, glob, sys
f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_lat") + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_pyr_copy0" in r["Kernel_Name"]]
i0 = idx[-5]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:idx[-4]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s start %7.1f us  dur %6.1f us  queue %s" % (r["Kernel_Name"].split("(")[0][-26:], (s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "")))
