#!/usr/bin/env python3
"""Writes two frames of the benchmark sequence in the raw format examples/mono_frontend reads (width, height, pixels)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import synth  # noqa: E402

f = synth.sequence(640, 480, 2, seed=1, step=(2, 1), per_frame_density=True)
open(sys.argv[1], "wb").write(np.array([640, 480], np.int32).tobytes() + f.tobytes())
