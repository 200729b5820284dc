#!/usr/bin/env python3
"""BASELINE.md section 3 / SURVEY.md 8(d): the CPU baseline `bench.py` times on the GPU box is the oracle
(oracle/ttrnn_oracle.py, an op-for-op torch-CPU restatement of the reference path), because the reference itself cannot
travel.  This script — build container only, it imports /root/reference — checks that the restatement is a fair stand-in
for the reference's SPEED as well as for its numbers: same weights, same input, same thread count, interleaved runs,
median of 5; the wall-time ratio has to stay within +-10 %.

    python tools/validate_oracle_speed.py [--threads 1,8] [--steps 196] [--out profiles/r2/oracle_speed_validation.json]
"""
import argparse
import contextlib
import io
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,8")
    ap.add_argument("--steps", type=int, default=196, help="timesteps of the cfg2 sequence to run (of 784)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r2", "oracle_speed_validation.json"))
    args = ap.parse_args()

    sys.path.insert(0, "/root/reference")
    with contextlib.redirect_stdout(io.StringIO()):
        from tensorized_rnn.tt_lstm import TTLSTM           # the reference's own class
        torch.manual_seed(1111)
        ref = TTLSTM(1, 256, 1, torch.device("cpu"), n_cores=3, tt_rank=8)
    from oracle import ttrnn_oracle as O
    sd = {k: v.detach() for k, v in ref.state_dict().items()}
    layers, _ = O.layers_from_state_dict(sd, 1)
    g = torch.Generator().manual_seed(1111)
    x = torch.rand(64, args.steps, 1, generator=g)           # cfg2: batch 64, in = 1

    def t_ref():
        t0 = time.perf_counter()
        with torch.no_grad():
            out, _ = ref(x)
        return time.perf_counter() - t0, out

    def t_orc():
        t0 = time.perf_counter()
        with torch.no_grad():
            out, _ = O.lstm_forward(layers, x)
        return time.perf_counter() - t0, out

    rows = []
    ok = True
    for n in [int(v) for v in args.threads.split(",")]:
        torch.set_num_threads(n)
        t_ref(); t_orc()                                      # warm-up
        a, b = [], []
        for _ in range(5):                                    # interleaved: drift hits both alike
            ta, oa = t_ref()
            tb, ob = t_orc()
            a.append(ta); b.append(tb)
        err = float((oa - ob).abs().max())
        ra, rb = statistics.median(a), statistics.median(b)
        ratio = rb / ra
        rows.append({"threads": n, "reference_s": ra, "oracle_s": rb, "oracle_over_reference": ratio,
                     "reference_timesteps_per_s": args.steps / ra, "oracle_timesteps_per_s": args.steps / rb,
                     "max_abs_output_diff": err})
        ok = ok and abs(ratio - 1.0) <= 0.10 and err <= 1e-6
    rec = {"workload": "cfg2 TT-LSTM in=1 H=256 ncores=3 ttrank=8, batch 64, first %d of 784 timesteps, fp32, no_grad" % args.steps,
           "host": {"logical_cpus": os.cpu_count(), "torch": torch.__version__},
           "runs": "median of 5, reference and oracle interleaved", "rows": rows, "within_10_percent": ok}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
