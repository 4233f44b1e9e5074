#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of one configuration of bench.py's configs block.
# usage (inside gpurun): bash tools/prof_cfg.sh <tag> <config> [batch]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 tools/cfg_probe.py "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob("gpurun_out/prof_%s/*/*_kernel_stats.csv" % tag)[0]
for r in csv.DictReader(open(f)):
    name = r["Name"].split("(")[0][-44:]
    print("%-46s calls %5s avg_us %10.1f total_ms %8.2f" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
