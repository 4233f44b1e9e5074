#!/bin/bash
# Runs on the GPU box (via gpurun): one rocprofv3 PMC pass of the named counters over a short bench.py run
# (one stream, 512 frames per launch); prints per-kernel per-launch averages.
# usage: bash tools/pmc_any.sh <tag> COUNTER...   (at most 8 SQ_, 4 TCC_, 4 TCP_ counters per pass)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 bench.py --batch 512 --sub-batches 1 --streams 1 --steps 2 --warmup 1 --no-cpu --no-latency --no-configs > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out "$@" <<'PY'
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0])):
    if "orbgpu" in r["Kernel_Name"]:
        k = re.sub(r"<.*>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").replace("orbgpu::", "").strip()
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sys.argv[2:]
print("%-20s" % "kernel" + "".join("%18s" % n[-17:] for n in names))
for k, d in acc.items():
    print("%-20s" % k + "".join("%18.0f" % (sum(d[n]) / max(1, len(d[n]))) for n in names))
PY
