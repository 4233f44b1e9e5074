#!/bin/bash
# Runs on the GPU box: kernel trace of a one-stream bench run; prints the mean duration of every kernel by grid size
# (the resize launches of the pyramid levels differ in grid size).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_lv -- python3 bench.py --no-cpu --no-latency --batch 512 --streams 1 --steps 3 --warmup 1 > gpurun_out/prof_lv.log 2>&1 || { tail -3 gpurun_out/prof_lv.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_lv/*/*_kernel_trace.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("orbgpu::", "")[:28]
    acc[(n, r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in acc.items():
    print("%-30s grid %8s x %5s calls %4d mean_us %8.1f" % (k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3))
PY
