#!/bin/bash
# Runs on the GPU box (via gpurun): one rocprofv3 PMC pass of SQ instruction counters over a
# short bench.py run (one stream, 512 frames per launch: per-dispatch counters need kernels that
# do not overlap); prints per-kernel wave-instruction counts per launch.
# usage: bash tools/pmc_sq.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_sq_$tag
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out -- python3 bench.py --batch 512 --sub-batches 1 --streams 1 --steps 2 --warmup 1 --no-cpu --no-latency --no-configs > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out <<'PY'
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0])):
    if "orbgpu" in r["Kernel_Name"]:
        k = re.sub(r"<.*>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").replace("orbgpu::", "").strip()
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_BUSY_CYCLES"]
print("%-20s" % "kernel" + "".join("%16s" % n[3:] for n in names))
for k, d in acc.items():
    print("%-20s" % k + "".join("%16.0f" % (sum(d[n]) / max(1, len(d[n]))) for n in names))
PY
