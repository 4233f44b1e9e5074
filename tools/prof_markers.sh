# Runs on the GPU box: marker (roctx) + kernel trace of a few single-frame calls with ORBGPU_ROCTX=1; prints the ranges seen.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ORBGPU_ROCTX=1
timeout -k 10 200 rocprofv3 --marker-trace --kernel-trace --output-format csv -d gpurun_out/prof_markers -- python3 tools/latency_probe.py 20 > gpurun_out/markers.log 2>&1 || { tail -5 gpurun_out/markers.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_markers/*/*_marker_api_trace.csv")
if not f:
    print("no marker trace file:", glob.glob("gpurun_out/prof_markers/*/*"))
    raise SystemExit(1)
acc = collections.Counter()
for r in csv.DictReader(open(f[0])):
    acc[r.get("Function", r.get("Name", "?"))] += 1
for k, v in sorted(acc.items()):
    print("%-40s %d" % (k, v))
PY
