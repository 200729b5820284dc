"""CPU oracle of the GE2E similarity matrix / softmax loss / EER — TEST INFRASTRUCTURE ONLY.

Loop-for-loop restatement of experiments/speaker_verification/encoder/speaker_encoder.py:93-170 of the reference
(per-speaker Python loop, torch CPU ops; EER from sklearn's roc_curve + scipy's brentq like the reference's snippet).
Pinned by tests/golden/g9_ge2e_*.npz, which tests/golden/gen_golden_ge2e.py produced by running the reference itself.
Imported by tests/ only; the product (ttrnn_hip/ge2e.py) never imports it."""
import numpy as np
import torch


def similarity_matrix(verification, weight, bias, enrollment=None):
    """speaker_encoder.py:93-140: [S, U, D] (L2-normalised) -> [S, U, S]."""
    S, U = verification.shape[:2]
    sim = torch.zeros(S, U, S, dtype=verification.dtype)
    incl = torch.mean(verification, dim=1, keepdim=True)
    incl = incl / torch.norm(incl, dim=2, keepdim=True)
    if enrollment is None:
        excl = (torch.sum(verification, dim=1, keepdim=True) - verification) / (U - 1)
        excl = excl / torch.norm(excl, dim=2, keepdim=True)
        for j in range(S):
            others = [s for s in range(S) if s != j]
            sim[others, :, j] = (verification[others] * incl[j]).sum(dim=2)      # :121
            sim[j, :, j] = (verification[j] * excl[j]).sum(dim=1)                # :122
    else:
        cen = torch.mean(enrollment, dim=1, keepdim=True)
        cen = cen / torch.norm(cen, dim=2, keepdim=True)
        for j in range(S):
            sim[:, :, j] = (verification * cen[j, :, :]).sum(dim=2)              # :127
    return sim * weight + bias                                                   # :139


def eer_of(sim2d, S, U):
    """speaker_encoder.py:160-168 (not back-propagated)."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import roc_curve
    truth = np.repeat(np.arange(S), U)
    labels = np.eye(S, dtype=np.int64)[truth]
    fpr, tpr, _ = roc_curve(labels.flatten(), sim2d.flatten())
    return float(brentq(lambda v: 1. - v - interp1d(fpr, tpr)(v), 0., 1.))


def loss(verification, weight, bias, enrollment=None):
    """speaker_encoder.py:142-170: (cross-entropy loss, EER)."""
    S, U = verification.shape[:2]
    sim = similarity_matrix(verification, weight, bias, enrollment).reshape(S * U, S)
    target = torch.from_numpy(np.repeat(np.arange(S), U)).long()
    l = torch.nn.functional.cross_entropy(sim, target)
    return l, eer_of(sim.detach().numpy(), S, U)
