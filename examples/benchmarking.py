#!/usr/bin/env python3
"""Synthetic timing harness with the semantics of the reference's experiments/digit_classification/
benchmarking.py:16-108 (fixed uniform[0,1) batch, one warm-up, N timed runs, eval = no_grad forward,
train = zero_grad + forward + nll_loss + backward + Adam step) — plus the explicit device synchronisation the
reference lacks.  Example:  python examples/benchmarking.py --tt --ncores 3 --ttrank 8 --hidden_size 256 \
    --in_size 1 --seq_len 784 --batch_size 64 [--gru] [--train]"""
import argparse
import contextlib
import io
import time

import numpy as np
import torch
import torch.nn.functional as F

from models import MNISTClassifier


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tt", action="store_true")
    ap.add_argument("--ncores", type=int, default=3)
    ap.add_argument("--ttrank", type=int, default=8)
    ap.add_argument("--batch_size", type=int, default=512)
    ap.add_argument("--in_size", type=int, default=256)
    ap.add_argument("--hidden_size", type=int, default=512)
    ap.add_argument("--n_layers", type=int, default=1)
    ap.add_argument("--seq_len", type=int, default=160)
    ap.add_argument("--emb_size", type=int, default=256)
    ap.add_argument("-n", "--nruns", type=int, default=20)
    ap.add_argument("--gru", action="store_true")
    ap.add_argument("--naive_tt", action="store_true")
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="--train only: record the step once into a hipGraph (ttrnn_hip.CapturedTrainStep) and time replays")
    args = ap.parse_args()
    device = torch.device("cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        model = MNISTClassifier(args.in_size, args.emb_size, args.hidden_size, args.n_layers, device, gru=args.gru,
                                n_cores=args.ncores, tt_rank=args.ttrank, naive_tt=args.naive_tt).to(device)
    data = torch.from_numpy(np.random.rand(args.batch_size, args.seq_len, args.in_size).astype("float32")).to(device)
    target = torch.from_numpy(np.random.randint(0, args.emb_size, args.batch_size).astype("int64")).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    captured = None
    if args.graph and args.train:
        import ttrnn_hip
        opt = ttrnn_hip.adam_for_capture(model.parameters(), lr=1e-3)
        captured = ttrnn_hip.CapturedTrainStep(model, opt, lambda m, d, t: F.nll_loss(m(d), t), (data, target))

    def step():
        if captured is not None:
            return captured(data, target)
        if not args.train:
            with torch.no_grad():
                return model(data)
        opt.zero_grad()
        loss = F.nll_loss(model(data), target)
        loss.backward()
        opt.step()
        return loss

    step()
    torch.cuda.synchronize()
    durations = []
    for _ in range(args.nruns):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        durations.append(time.perf_counter() - t0)
    d = np.array(durations)
    print("mean time: {:.6f} s \t std time: {:.6f} s \t timesteps/s: {:.0f}".format(d.mean(), d.std(), args.seq_len / d.mean()))
    print("model has: {} parameters".format(model.param_count()))


if __name__ == "__main__":
    main()
