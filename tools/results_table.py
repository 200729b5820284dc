#!/usr/bin/env python3
"""Markdown result tables from one collection run (tools/collect_round.sh): python tools/results_table.py profiles/r4
Prints the DESIGN.md section 8 table; every number is read from the bench lines / rocprof summaries in that directory."""
import csv
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "profiles/r6"
PREV = {"cfg1": (0.369, 1.04), "cfg2": (0.493, 1.64), "cfg3": (0.473, 1.88), "cfg3_fp32": (0.652, 2.11), "cfg4": (1.555, 4.94), "cfg5": (7.039, 20.85),
        "spk": (1.44, 5.53)}      # round 5 (spk: the harness figures of profiles/r5/variants_benchmarking.txt)
NAMES = {"cfg1": "cfg1 TT-LSTM H=128 d=2 r=4 B=32 T=784 fp32", "cfg2": "**cfg2** TT-LSTM H=256 d=3 r=8 B=64 T=784 fp32 (headline)",
         "cfg3": "cfg3 TT-GRU H=256 d=3 r=8 B=256 T=784 bf16", "cfg4": "cfg4 3-layer TT-LSTM H=256 r=16 in=40 B=512 T=160",
         "cfg5": "cfg5 TT-LSTM H=in=1024 d=4 r=32 B=128 T=1024",
         "cfg3_fp32": "cfg3 in the reference's own dtype: TT-GRU H=256 d=3 r=8 B=256 T=784 fp32",
         "spk": "the reference's speaker-encoder layer: TT-LSTM in=40 H=768 d=2 r=2 B=512 T=160 fp32"}
KERN = {"cfg1": "k_lstm_fwd_f2", "cfg2": "k_lstm_fwd_f10q", "cfg3": "k_gru_fwd_f10v", "cfg3_fp32": "k_gru_fwd_f10vh", "cfg4": "k_lstm_fwd_f10q",
        "cfg5": "k_lstm_fwd_big2h", "spk": "k_lstm_fwd_w2"}


def line(path):
    with open(path) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def kernel_avg(w):
    path = os.path.join(src, "rocprof_kernel_stats_%s.csv" % w)
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if KERN[w] in r["Name"]:
            return int(r["Calls"]), float(r["AverageNs"]) / 1e3
    return None


print("| workload (per GPU) | forward ms / step (prepared) | timesteps/s | `exact` mode ms | roofline frac (basis) / algorithmic / executed; chip occupancy | "
      "dominant kernel, rocprof avg µs (calls) | HBM bytes per call (PMC) / algorithmic | train step ms | round 5: fwd / train |")
print("|---|---|---|---|---|---|---|---|---|")
for w in ("cfg2", "spk", "cfg1", "cfg3", "cfg3_fp32", "cfg4", "cfg5"):
    f = line(os.path.join(src, "bench_%s.json" % w))
    tp = os.path.join(src, "bench_train_%s.json" % w)
    t = line(tp) if os.path.exists(tp) else None
    ka = kernel_avg(w)
    ex = (f.get("other_fp32_math") or {}).get("ms_per_step")
    algo = {"cfg1": 516, "cfg2": 1028, "cfg3": 514, "cfg3_fp32": 1028, "cfg4": 1184, "cfg5": 8192, "spk": 3232}[w] * f["config"]["per_gpu_batch"] * f["config"]["seq_len"]
    exe = (f["roofline"].get("executed") or {}).get("frac")
    r = f["roofline"]
    if not r.get("traffic"):      # (a workload whose PMC summary was taken after its bench line: profiles/traffic.json has it)
        try:
            r["traffic"] = json.load(open(os.path.join(os.path.dirname(os.path.abspath(src)), "traffic.json")))[w]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
    basis = "executed" if (r.get("frac_basis") or "").startswith("executed") else "algorithmic"
    print("| %s | **%.3f** (%.3f) | %s | %s | %.3f (%s) / %.3f / %s; %.2f | `%s` %s | %.3g / %.3g | %s | %s / %s |" % (
        NAMES[w], f["ms_per_step"], f["prepared"]["ms_per_step"], "{:,.0f}".format(f["value"]).replace(",", " "),
        "%.2f" % ex if ex else "–", r["frac"] if r.get("frac") is not None else float("nan"), basis, r["frac_algorithmic"], "%.3f" % exe if exe else "–", r["chip_occupancy"], KERN[w],
        "%.1f (%d)" % (ka[1], ka[0]) if ka else "–", r["traffic"] or float("nan"), algo,
        "**%.2f**" % t["ms_per_step"] if t else "–",
        PREV[w][0], PREV[w][1] if PREV[w][1] else "–"))
c = line(os.path.join(src, "bench_cfg2.json"))["cpu_baseline"]
print()
print("CPU baseline (oracle = op-for-op restatement of the reference's path, same weights and input, %s, %d physical cores): cfg2 %s "
      "timesteps/s at 1 / 8 / all threads." % (c["cpu_model"], c["physical_cores"],
                                               " / ".join("%.0f" % c["by_threads"][k] for k in sorted(c["by_threads"], key=int))))
gp = os.path.join(src, "bench_grid.json")
if os.path.exists(gp):
    g = line(gp)
    print("Grid (144 shapes, B = 64, T = 64, in = 40, forward): routes %s, geometric-mean speed-up over the VALU kernels %.2fx." % (g["routes"], g["value"]))
gp = os.path.join(src, "bench_grid_train_graph.json")
if os.path.exists(gp):
    g = line(gp)
    rows = g["grid"]
    small = [r for r in rows if r["H"] <= 128]
    print("Grid, training step (forward + BPTT of sum(outputs)): bwd routes %s; eager ms min / median %.3f / %.3f, hipGraph replay %.3f / %.3f; "
          "H <= 128 shapes: eager %.3f ... %.3f, replay %.3f ... %.3f." % (
              g["bwd_routes"], g["eager_ms"]["min"], g["eager_ms"]["median"], g["graph_replay_ms"]["min"], g["graph_replay_ms"]["median"],
              min(r["ms"] for r in small), max(r["ms"] for r in small), min(r["graph_ms"] for r in small), max(r["graph_ms"] for r in small)))
