#!/usr/bin/env python3
"""One speaker-verification training step on synthetic data with the semantics of the reference's
experiments/speaker_verification/main.py:65-78,257-290 (cfg4): utterances [speakers*utterances, 160, 40] -> 3-layer
TT-LSTM -> TTLinear -> ReLU -> L2 norm -> GE2E loss on [speakers, utterances, 256] -> backward -> gradient scaling /
clipping -> Adam(1e-3).  Everything, the loss included, stays on the GPU (the reference moves the embeddings to the CPU
for its per-speaker loop).  Example:  python examples/speaker_step.py --speakers 16 --utterances 32 -n 10"""
import argparse
import contextlib
import io
import time

import torch

from models import SpeakerEncoder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--speakers", type=int, default=16)
    ap.add_argument("--utterances", type=int, default=32)
    ap.add_argument("--frames", type=int, default=160)
    ap.add_argument("--mels", type=int, default=40)
    ap.add_argument("--hidden_size", type=int, default=256)
    ap.add_argument("--n_layers", type=int, default=3)
    ap.add_argument("--emb_size", type=int, default=256)
    ap.add_argument("--ncores", type=int, default=3)
    ap.add_argument("--ttrank", type=int, default=16)
    ap.add_argument("--gru", action="store_true")
    ap.add_argument("-n", "--nruns", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        model = SpeakerEncoder(args.mels, args.hidden_size, args.n_layers, args.emb_size, dev, n_cores=args.ncores,
                               rank=args.ttrank, use_gru=args.gru).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    S, U = args.speakers, args.utterances
    x = torch.rand(S * U, args.frames, args.mels, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        embeds = model(x).view(S, U, -1)
        from ttrnn_hip import ge2e
        loss, _ = ge2e.ge2e_loss(embeds, model.similarity_weight, model.similarity_bias, None, with_eer=False)
        loss.backward()
        model.do_gradient_ops()
        opt.step()
        return loss

    loss = step()
    torch.cuda.synchronize()
    times = []
    for _ in range(args.nruns):
        t0 = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    with torch.no_grad():
        embeds = model(x).view(S, U, -1)
        _, eer = model.loss(embeds)
    ms = 1e3 * sum(times) / len(times)
    print("speaker-encoder train step: %.2f ms (min %.2f) for %d utterances x %d frames; loss %.4f, EER %.3f" % (
        ms, 1e3 * min(times), S * U, args.frames, float(loss), eer))


if __name__ == "__main__":
    main()
