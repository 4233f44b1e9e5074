#!/usr/bin/env python3
"""Print the headline fields of a `bench.py --workload 4k` JSON line."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"]), "frames/s; isolated ms:", {k: round(v, 3) for k, v in d["stages_ms_isolated"].items()})
