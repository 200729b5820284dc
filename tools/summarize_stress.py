"""Condense the logs of tools/stress_boxes.sh (gpurun_out/stress/b*.log: one file per GPU box) into one JSON:
per box the GPU's unique id, the stress records of tools/stress_determinism.py (launches compared, launches that
differed) and the verdicts of the pytest repetitions run on the same box.

    python tools/summarize_stress.py gpurun_out/stress [log prefix, default b] > profiles/r2/determinism_boxes.json
"""
import glob
import json
import os
import re
import sys


def main(root, prefix="b"):
    boxes = []
    for path in sorted(glob.glob(os.path.join(root, prefix + "*.log"))):
        name = os.path.basename(path)
        if ".pytest." in name:
            continue
        text = open(path, errors="replace").read()
        uid = re.search(r"Unique ID: (0x[0-9a-f]+)", text)
        recs = []
        for line in text.splitlines():
            line = line.strip()
            if line.startswith("{") and '"bad_launches"' in line:
                try:
                    recs.append(json.loads(line))
                except ValueError:
                    pass
        launches = sum(r["reps"] for r in recs)
        pyt = []
        for p in sorted(glob.glob(os.path.join(root, name.replace(".log", "") + ".pytest.*.log"))):
            t = open(p, errors="replace").read()
            m = re.findall(r"(\d+) passed", t)
            f = re.findall(r"(\d+) failed", t)
            pyt.append({"passed": int(m[-1]) if m else 0, "failed": int(f[-1]) if f else 0})
        boxes.append({"log": name, "gpu_unique_id": uid.group(1) if uid else None,
                      "stress_runs": len(recs), "poisoned_runs": sum(1 for r in recs if r.get("poison")),
                      "launches_compared": launches, "launches_that_differed": sum(r["bad_launches"] for r in recs),
                      "cases": sorted({r["case"] + "/" + r["route"] for r in recs}),
                      "pytest_repeats": pyt})
    out = {"boxes": boxes,
           "distinct_gpus": len({b["gpu_unique_id"] for b in boxes if b["gpu_unique_id"]}),
           "launches_compared": sum(b["launches_compared"] for b in boxes),
           "launches_that_differed": sum(b["launches_that_differed"] for b in boxes),
           "pytest_runs": sum(len(b["pytest_repeats"]) for b in boxes),
           "pytest_runs_with_failures": sum(1 for b in boxes for p in b["pytest_repeats"] if p["failed"]),
           "note": "every launch is compared bit for bit with the first launch of its run (outputs, final states, packed "
                   "cores and, without --poison, the whole workspace); pytest failures, where any, are listed per box and "
                   "explained in DESIGN.md section 9"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stress", sys.argv[2] if len(sys.argv) > 2 else "b")
