#!/usr/bin/env python3
"""pMNIST training loop with the flags and step structure of the reference's
experiments/digit_classification/pmnist_test.py:17-63,111-180 (view to [B,784,1], fixed pixel permutation,
MNIST_Classifier, NLL loss, Adam lr 1e-2, StepLR(10), optional grad clipping) on synthetic MNIST-shaped data —
torchvision / the dataset / Comet are not available offline.  Example (BASELINE configs[0] flags):
    python examples/pmnist_synthetic.py --tt --ncores 2 --ttrank 4 --hidden_size 128 --batch_size 32 --permute"""
import argparse
import contextlib
import io

import torch
import torch.nn.functional as F

from models import MNISTClassifier


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tt", action="store_true")
    ap.add_argument("--gru", action="store_true")
    ap.add_argument("--naive_tt", action="store_true")
    ap.add_argument("--extra_core", default=None, choices=[None, "first", "last"])
    ap.add_argument("--ncores", type=int, default=2)
    ap.add_argument("--ttrank", type=int, default=4)
    ap.add_argument("--hidden_size", type=int, default=128)
    ap.add_argument("--n_layers", type=int, default=1)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--permute", action="store_true")
    ap.add_argument("--clip", type=float, default=-1)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1111)
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    device = torch.device("cuda")
    seq_length, input_channels, n_classes = 784, 1, 10
    with contextlib.redirect_stdout(io.StringIO()):
        model = MNISTClassifier(input_channels, n_classes, args.hidden_size, args.n_layers, device, gru=args.gru,
                                n_cores=args.ncores, tt_rank=args.ttrank, naive_tt=args.naive_tt,
                                extra_core=args.extra_core).to(device)
    permute = torch.randperm(seq_length) if args.permute else torch.arange(seq_length)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=10, gamma=0.5)
    # a fixed synthetic "dataset": class-dependent means so that the loss can actually go down
    protos = torch.rand(n_classes, 28 * 28)
    for it in range(args.steps):
        target = torch.randint(0, n_classes, (args.batch_size,))
        images = (protos[target] + 0.1 * torch.randn(args.batch_size, 28 * 28)).clamp(0, 1)
        data = images.view(-1, seq_length, input_channels)[:, permute, :].to(device)
        opt.zero_grad()
        loss = F.nll_loss(model(data), target.to(device))
        loss.backward()
        if args.clip > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip)
        opt.step()
        sched.step()
        print("step {:3d}  loss {:.4f}".format(it, loss.item()))


if __name__ == "__main__":
    main()
