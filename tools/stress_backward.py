#!/usr/bin/env python3
"""Repeat-launch stress of the TRAINING step (forward + BPTT) under poisoned allocations: every buffer the binding allocates without
initialising it (outputs, workspaces, reserves, gradient buffers) is filled with random bytes before each launch, so a kernel that reads
memory nobody wrote, leaves part of an output unwritten or races on an LDS hand-off changes its results from step to step.  Every
gradient of every repetition must equal the first repetition's bit for bit (the routes exercised here have no atomics in them:
DESIGN.md section 9).      python tools/stress_backward.py [--reps 40] [--cases h512,cfg4,gru512,h768,cfg2,cfg1,naive]
Exit code 1 when any repetition differed."""
import argparse, contextlib, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
import torch
from tensorized_rnn.tt_lstm import TTLSTM
from tensorized_rnn.gru import TTGRU
from ttrnn_hip import functional as F

CASES = {   # name: (cell, in, H, layers, d, r, naive, B, T)
    "h512": ("lstm", 256, 512, 1, 3, 8, False, 512, 160),      # fused core both ways, eight gate waves (k_lstm_bwd_f10l)
    "cfg4": ("lstm", 40, 256, 3, 3, 16, False, 512, 160),      # four-wave wave-local reverse kernel, two workgroups per CU
    "cfg2": ("lstm", 1, 256, 1, 3, 8, False, 64, 784),         # eight-wave reverse kernel
    "cfg1": ("lstm", 1, 128, 1, 2, 4, False, 32, 784),         # two-core kernels
    "gru512": ("gru", 256, 512, 1, 3, 8, False, 512, 160),     # runtime tier, resident fragments, eight-wave plan
    "h768": ("lstm", 40, 768, 1, 4, 8, False, 512, 160),       # runtime tier, streamed fragments
    "naive": ("lstm", 40, 256, 1, 3, 8, True, 300, 40),        # runtime tier, block-diagonal head
}
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--cases", default=",".join(CASES))
ap.add_argument("--grid", action="store_true", help="instead of the named cases: bench.py's grid of shapes (in = 40, batch 64, --grid-steps steps)")
ap.add_argument("--grid-steps", type=int, default=64)
ap.add_argument("--no-poison", action="store_true", help="plain allocations: a difference that disappears here is an uninitialised read")
ap.add_argument("--only", default="", help="--grid: substring filter on the case names")
a = ap.parse_args()
if a.grid:
    CASES = {"%s-H%d-d%d-r%d" % (c, H, d, r): (c, 40, H, 1, d, r, False, 64, a.grid_steps)
             for c in ("lstm", "gru") for H in (64, 128, 256, 384, 512, 768, 1024) for d in (2, 3, 4) for r in (2, 4, 8, 16)}
    a.cases = ",".join(k for k in CASES if a.only in k)
dev = torch.device("cuda:0")
F.POISON_ALLOCATIONS = not a.no_poison
bad = 0
for name in a.cases.split(","):
    cell, inp, H, L, d, r, naive, B, T = CASES[name]
    torch.manual_seed(5)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            m = (TTGRU if cell == "gru" else TTLSTM)(inp, H, L, dev, n_cores=d, tt_rank=r, is_naive=naive)
    except Exception as e:      # (a shape the reference's auto_shape does not factor)
        print(json.dumps({"case": name, "skipped": str(e)[:80]}))
        continue
    x = torch.randn(B, T, inp, device=dev)
    w = torch.randn(B, T, H, device=dev)
    first, differing = None, set()
    for rep in range(a.reps):
        m.zero_grad(set_to_none=True)
        out = m(x)[0]
        (out * w).sum().backward()
        grads = [out.detach().clone()] + [p.grad.clone() for p in m.parameters()]
        if first is None:
            first = grads
            finite = all(bool(torch.isfinite(g).all()) for g in grads)
        else:
            for i, (g0, g1) in enumerate(zip(first, grads)):
                if not torch.equal(g0, g1):
                    differing.add(i)
    torch.cuda.synchronize()
    names = ["out"] + [n for n, _ in m.named_parameters()]
    if not a.grid or differing or not finite:
        print(json.dumps({"case": name, "reps": a.reps, "finite": finite, "differing": [names[i] for i in sorted(differing)],
                          "bwd_route": F.rnn_backward_route(m._all_layers[0]._layer_spec(), B, T)}))
    bad += 1 if (differing or not finite) else 0
print(json.dumps({"cases": len(a.cases.split(",")), "cases_that_differed_or_were_not_finite": bad}))
sys.exit(1 if bad else 0)
