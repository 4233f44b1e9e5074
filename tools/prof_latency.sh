cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lat -- python3 tools/latency_probe.py 100 > gpurun_out/lat.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_lat/*/*_kernel_stats.csv")[0]
tot=0
for r in csv.DictReader(open(f)):
    name = r["Name"].split("(")[0][-40:]
    print("%-42s calls %5s avg_us %8.1f" % (name, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 gpurun_out/lat.log
