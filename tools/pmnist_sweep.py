#!/usr/bin/env python3
"""Training / eval time of the pMNIST classifier (in = 1, 784 steps; pmnist_test.py:47-56 flag space) over hidden size, number
of cores, TT-rank, extra core and cell type — to catch flag combinations that fall off the fast routes.
    python tools/pmnist_sweep.py [--batch 64] > gpurun_out/pmnist_sweep.txt"""
import argparse
import contextlib
import io
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorized-rnn_amd"), os.path.join(ROOT, "examples")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from models import MNISTClassifier  # noqa: E402
from ttrnn_hip import functional as TF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--naive", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda")
    rows = []
    for H, d, r, extra, gru in itertools.product((128, 256, 512), (2, 3, 4), (4, 8, 16), (None, "first", "last"), (False, True)):
        torch.manual_seed(1)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                m = MNISTClassifier(1, 10, H, 1, dev, gru=gru, n_cores=d, tt_rank=r, extra_core=extra, naive_tt=args.naive).to(dev)
        except Exception as e:  # shapes the reference's tt_shape rejects
            print("skip", H, d, r, extra, gru, type(e).__name__)
            continue
        x = torch.rand(args.batch, 784, 1, device=dev)
        y = torch.randint(0, 10, (args.batch,), device=dev)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)

        def train():
            opt.zero_grad()
            F.nll_loss(m(x), y).backward()
            opt.step()

        def ev():
            with torch.no_grad():
                m(x)
        res = []
        for fn in (ev, train):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 3 * 1e3)
        spec = m.rnn._all_layers[0]._layer_spec()
        route = TF.rnn_route(spec, args.batch, 784)
        broute = TF.rnn_backward_route(spec, args.batch, 784)
        rows.append((res[1], res[0], H, d, r, extra, "gru" if gru else "lstm", route, broute))
        print("H %4d d %d r %2d extra %-5s %-4s eval %7.2f ms train %7.2f ms  ratio %4.1f  %s / %s" % (
            H, d, r, extra, "gru" if gru else "lstm", res[0], res[1], res[1] / res[0], route, broute), flush=True)
    rows.sort(reverse=True)
    print("---- slowest training steps")
    for r_ in rows[:12]:
        print(r_)
    print("---- largest train / eval ratios")
    for r_ in sorted(rows, key=lambda q: -q[0] / q[1])[:12]:
        print(round(r_[0] / r_[1], 1), r_)


if __name__ == "__main__":
    main()
