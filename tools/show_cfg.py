#!/usr/bin/env python3
"""Print the headline numbers of a tools/cfg_probe.py / bench.py configs-block JSON line (last line of the file)."""
import json
import sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
j = j.get("configs", {"": j})
for name, c in j.items():
    if "frames_per_call" in c:
        for k, v in c["frames_per_call"].items():
            print(name, k, round(v["frames_per_s"]), v.get("stages_ms_isolated"))
    for key in ("stereo_frames_per_s", "stereo_match_ms_per_batch", "pairs_per_s", "error"):
        if key in c:
            print(name, key, c[key])
    for vn, v in c.get("variants", {}).items():
        print(name, vn[:24], "pairs/s", round(v["pairs_per_s"]), "ms", round(v["ms_per_launch"], 4))
    if "parity_in_run" in c:
        print(name, "parity", c["parity_in_run"])
