# experiment: single-frame latency with parts of the call outside the HIP graph (ORBGPU_EXP bits: 1 upload, 2 download, 4 no memset)
cd "$GRAFT_REPO_ROOT"
for g in 0 1 2 3 4 7; do
ORBGPU_EXP=$g timeout -k 10 200 python3 bench.py --no-configs --no-cpu --steps 2 --warmup 1 > gpurun_out/lat_exp$g.log 2>&1 || exit 1
python3 - gpurun_out/lat_exp$g.log $g <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
l=d["single_frame_latency"]
print("exp", sys.argv[2], "python extract %.4f  c++ extract %.4f  device_frame_build %.4f" % (l["extract_ms"], l["cpp"]["extract_ms"], l["cpp"]["device_frame_build_ms"]))
PY
done
