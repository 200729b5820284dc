#!/usr/bin/env python3
"""A/B of functional.USE_ROW_SHIFT / USE_BWD_STATS on bench.py's training step: python tools/ab_row_shift.py cfg5 [steps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
steps = sys.argv[2] if len(sys.argv) > 2 else "15"
code = ("import sys; sys.path.insert(0, %r); sys.argv = ['bench.py', '--workload', %r, '--mode', 'train', '--steps', %r, "
        "'--warmup', '3', '--no-cpu-baseline']; import bench; import ttrnn_hip.functional as F; "
        "F.USE_ROW_SHIFT = bool(%d); F.USE_BWD_STATS = bool(%d); bench.main()")
for rep in range(2):
    for shift, stats in ((1, 1), (0, 1), (0, 0)):
        out = subprocess.run([sys.executable, "-c", code % (ROOT, w, steps, shift, stats)], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, universal_newlines=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print(w, "row_shift", shift, "bwd_stats", stats, "ms %.3f median %.3f" % (d["ms_per_step"], d["ms_per_step_median"]), flush=True)
