#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of tools/profile.sh (one per workload): HBM bytes per ttrnn_rnn_forward call
= sum over the library's kernels of (mean 2 x FETCH_SIZE + WRITE_SIZE per dispatch) x (dispatches per forward call), the
number of forward calls taken from the recurrent kernel's dispatch count.

    python tools/update_traffic.py gpurun_out/r3 profiles/r3    # reads prof_cfgN_summary.json, writes profiles/traffic.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REC = {"cfg1": "k_lstm_fwd_f2", "cfg2": "k_lstm_fwd_f10q", "cfg3": "k_gru_fwd_f10", "cfg3_fp32": "k_gru_fwd_f10vh", "cfg4": "k_lstm_fwd_f10q",
       "cfg5": "k_lstm_fwd_big2h", "spk": "k_lstm_fwd_w2"}
LAYERS = {"cfg1": 1, "cfg2": 1, "cfg3": 1, "cfg3_fp32": 1, "cfg4": 3, "cfg5": 1, "spk": 1}
# kernels of the OTHER math mode that bench.py also times in the same process (not part of the default forward)
SKIP = ("k_lstm_fwd_f10x", "k_f10x_prep", "k_lstm_fwd_fused", "k_ttlinear_fwd_fast[J=4x8x8", "k_ttlinear_fwd_fast[J=2x4x5", "k_lstm_fwd_big2[",
        "k_ttlinear_fwd_big", "k_rnn_fwd_fast", "k_rnn_fwd_bf16")


def main(src, label):
    out = {}
    for w in sorted(REC):
        path = os.path.join(src, "prof_%s_summary.json" % w)
        if not os.path.exists(path):
            continue
        pm = json.load(open(path))["pmc_mean_per_dispatch"]
        skip = tuple(s for s in SKIP if REC[w] != s.rstrip("["))
        rec = [k for k in pm if k.startswith(REC[w]) and not any(k.startswith(s) for s in skip)]
        if not rec or "hbm_bytes_per_dispatch" not in pm[rec[0]]:
            continue
        calls = pm[rec[0]]["_dispatches"] / float(LAYERS[w])          # forward calls of the model in the PMC pass
        total, parts = 0.0, {}
        for k, v in pm.items():
            if any(k.startswith(s) for s in skip) or "hbm_bytes_per_dispatch" not in v:
                continue
            b = v["hbm_bytes_per_dispatch"] * v["_dispatches"] / calls
            total += b
            if b > 0.01 * 1e6:
                parts[k] = round(b)
        out[w] = {"kernel": REC[w], "hbm_bytes_per_launch": total / LAYERS[w], "hbm_bytes_per_forward": total,
                  "by_kernel_per_forward": parts,
                  "source": "%s/prof_%s_summary.json (tools/profile.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                            "passes; read side doubled per the gfx950 correction; all kernels of the default-mode forward call, "
                            "weight-only prep kernels included at their per-forward share)" % (label, w)}
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items()}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
