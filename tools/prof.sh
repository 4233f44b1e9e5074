#!/bin/bash
# Run bench.py under rocprofv3 on the GPU box and print the per-kernel averages.
# usage (inside gpurun): bash tools/prof.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu "$@" > gpurun_out/bench_prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob("gpurun_out/prof_%s/*/*_kernel_stats.csv" % tag)[0]
for r in csv.DictReader(open(f)):
    name = r["Name"].split("(")[0][-40:]
    print("%-42s calls %5s avg_us %10.1f total_ms %8.2f" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -1 gpurun_out/bench_prof_$tag.log | cut -c1-200
