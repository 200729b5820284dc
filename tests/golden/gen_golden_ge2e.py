#!/usr/bin/env python3
"""Generate tests/golden/g9_ge2e_*.npz by RUNNING THE REFERENCE's GE2E similarity matrix and loss on CPU
(experiments/speaker_verification/encoder/speaker_encoder.py:93-170).  Build container only (needs /root/reference,
scipy, scikit-learn); only data is written.  The reference's `np.int` (removed in numpy 2) is aliased for the run."""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

np.int = int            # noqa: the reference predates numpy 1.24
REF = '/root/reference'
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, 'experiments', 'speaker_verification'))
from encoder.speaker_encoder import SpeakerEncoder  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CPU = torch.device('cpu')


def case(name, S, U, D, seed, enroll):
    torch.manual_seed(seed)
    with redirect_stdout(io.StringIO()):
        enc = SpeakerEncoder(40, 64, 1, D, CPU, CPU, compression=None)
    raw = torch.relu(torch.randn(S, U, D) + 0.3 * torch.randn(S, 1, D))       # speaker-dependent offset
    emb = (raw / raw.norm(dim=2, keepdim=True)).requires_grad_(True)
    enr = None
    if enroll:
        r2 = torch.relu(torch.randn(S, U, D) + 0.3 * torch.randn(S, 1, D))
        enr = r2 / r2.norm(dim=2, keepdim=True)
    sim = enc.similarity_matrix(emb, enr)
    loss, eer = enc.loss(emb, enr)
    loss.backward()
    np.savez(os.path.join(OUT, name + '.npz'), embeds=emb.detach().numpy(),
             enroll=(enr.numpy() if enr is not None else np.zeros(0, dtype=np.float32)),
             sim=sim.detach().numpy(), loss=np.float64(loss.item()), eer=np.float64(eer),
             d_embeds=emb.grad.numpy(), weight=np.float32(enc.similarity_weight.item()),
             bias=np.float32(enc.similarity_bias.item()))
    print(name, 'loss %.6f eer %.4f' % (loss.item(), eer))


if __name__ == '__main__':
    case('g9_ge2e_train_s6u5', 6, 5, 32, 11, False)
    case('g9_ge2e_train_s16u8', 16, 8, 256, 12, False)
    case('g9_ge2e_enroll_s5u4', 5, 4, 32, 13, True)
