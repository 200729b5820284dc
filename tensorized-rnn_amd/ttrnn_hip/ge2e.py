"""GE2E similarity matrix and softmax loss for the speaker-verification caller of the hot path.

The reference builds the [S, U, S] similarity matrix on the CPU with a Python loop over speakers
(experiments/speaker_verification/encoder/speaker_encoder.py:109-127; main.py moves the embeddings to `loss_device`
first).  Here the whole thing is a handful of device-side tensor ops — inclusive centroids, leave-one-out centroids for the
diagonal, one [S*U, D] x [D, S] product — so it stays on the GPU next to the encoder, and under data parallelism every
rank contributes its utterances through one all-gather of the [S_local, U, D] embeddings (gradients flow back to the
local slice only; the loss is the mean over ALL utterances, as on a single device)."""
import numpy as np
import torch
import torch.nn.functional as F


def similarity_matrix(verification, weight, bias, enrollment=None):
    """verification [S, U, D] L2-normalised embeddings -> scaled cosine similarities [S, U, S] (speaker_encoder.py:93-140)."""
    S, U, _ = verification.shape
    if enrollment is not None:
        cen = F.normalize(enrollment.mean(dim=1), dim=1)                           # [S, D]
        sim = torch.einsum("sud,jd->suj", verification, cen)
        return sim * weight + bias
    incl = F.normalize(verification.mean(dim=1), dim=1)                            # [S, D]
    excl = F.normalize((verification.sum(dim=1, keepdim=True) - verification) / (U - 1), dim=2)   # [S, U, D]
    sim = torch.einsum("sud,jd->suj", verification, incl)
    own = (verification * excl).sum(dim=2)                                         # [S, U]: speaker against its own rest
    eye = torch.eye(S, dtype=torch.bool, device=verification.device).unsqueeze(1)  # [S, 1, S]
    sim = torch.where(eye, own.unsqueeze(2), sim)
    return sim * weight + bias


def eer(sim2d, S, U):
    """Equal error rate of a [S*U, S] similarity matrix (speaker_encoder.py:160-168); host side, not back-propagated."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import roc_curve
    truth = np.repeat(np.arange(S), U)
    labels = np.eye(S, dtype=np.int64)[truth]
    fpr, tpr, _ = roc_curve(labels.flatten(), np.asarray(sim2d, dtype=np.float64).flatten())
    return float(brentq(lambda v: 1. - v - interp1d(fpr, tpr)(v), 0., 1.))


def ge2e_loss(verification, weight, bias, enrollment=None, with_eer=True):
    """(softmax loss over all S*U utterances, EER or None) — speaker_encoder.py:142-170."""
    S, U, _ = verification.shape
    sim = similarity_matrix(verification, weight, bias, enrollment).reshape(S * U, S)
    target = torch.arange(S, device=sim.device).repeat_interleave(U)
    loss = F.cross_entropy(sim, target)
    e = eer(sim.detach().float().cpu().numpy(), S, U) if with_eer else None
    return loss, e


class _GatherSpeakers(torch.autograd.Function):
    """all_gather along the speaker axis; backward hands every rank the gradient slice of its own speakers."""

    @staticmethod
    def forward(ctx, local, group):
        import torch.distributed as dist
        world = dist.get_world_size(group)
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous(), group=group)
        ctx.rank, ctx.n, ctx.world = dist.get_rank(group), local.shape[0], world
        return torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, grad):
        # Every rank evaluates the FULL loss, so rank r's slice of d(loss)/d(embeds) is already the single-device
        # gradient of its own utterances; the data-parallel step then takes the MEAN over ranks of every parameter
        # gradient.  The encoder's gradient flows through this slice only (the other ranks contribute zero for these
        # utterances), so it is scaled by `world` here to survive that mean.  The replicated similarity weight / bias
        # receive their full gradient on every rank, the mean leaves them exact, and the loss value stays unscaled.
        return grad[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous() * ctx.world, None


def ge2e_loss_data_parallel(local_embeds, weight, bias, group=None, with_eer=False, force_gather=False):
    """Every rank holds [S_local, U, D] embeddings of its own speakers.  The similarity matrix needs all centroids: one
    all-gather (cfg4: 512 x 256 floats), then each rank evaluates the FULL loss (identical on every rank, returned
    unscaled) and back-propagates into its slice.  After the gradient all-reduce (MEAN) of the data-parallel step
    (ttrnn_hip.dist.FlatGradAllReduce) every parameter holds its single-device gradient: the slice gradient is scaled by
    the world size in _GatherSpeakers.backward, the replicated similarity weight / bias are left alone.
    (The reference computes the loss on one CPU, main.py:280.)"""
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return ge2e_loss(local_embeds, weight, bias, None, with_eer)
    # force_gather / TTRNN_FORCE_COLLECTIVES=1: take the all-gather even in a one-rank group (drives RCCL on a 1-GPU box)
    if dist.get_world_size(group) == 1 and not (force_gather or os.environ.get("TTRNN_FORCE_COLLECTIVES") == "1"):
        return ge2e_loss(local_embeds, weight, bias, None, with_eer)
    full = _GatherSpeakers.apply(local_embeds, group)
    return ge2e_loss(full, weight, bias, None, with_eer)
