#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950):
    python tools/kernel_resources.py tensorized-rnn_amd/csrc/ttrnn_fast_f10q.hip [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "tensorized-rnn_amd", "csrc"), "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, universal_newlines=True).stderr
cur = None
rows = []
for ln in err.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", ln)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
        name = re.sub(r"ttrnn::", "", name.split("(")[0])
        name = re.sub(r"Shp<(\d+), ([\d, ]+)>", lambda mm: "Shp[" + mm.group(2).replace(", ", ".") + "]", name)
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if flt and flt not in r["name"]:
        continue
    print("%-110s VGPR %3d AGPR %3d spill %3d scratch %4d occ %d LDS %6d" % (
        r["name"][:110], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
        r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
