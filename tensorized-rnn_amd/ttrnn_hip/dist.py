"""Batch-data-parallel helpers for the TT-RNN path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The reference is single-device (no torch.distributed anywhere).  The path shards over batch — samples
never interact inside the RNN (t3nsor/ops.py:78-93 keeps the batch as a row index; gates are
pointwise) — so:
  * forward / inference: contiguous batch shards, NO collective;
  * training: one all-reduce of the gradients per step.  TT models are tiny (cfg2 38 KB, cfg4 514 KB
    of gradients), so the collective is latency-bound: every gradient is packed into ONE flat fp32
    bucket and reduced with a single call instead of one small ring per tensor; the mean over ranks
    keeps single-GPU semantics for mean-reduced losses.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) slice of n items for `rank` (first n % world ranks get one extra)."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x, rank=None, world=None, dim=0):
    """This rank's contiguous shard of a batch-first tensor."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(x.shape[dim], rank, world)
    return x.narrow(dim, lo, hi - lo)


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (strided TT cores keep their layout)."""
    for p in module.parameters():
        buf = p.detach().contiguous()
        dist.broadcast(buf, src=src, group=group)
        with torch.no_grad():
            p.copy_(buf)


class FlatGradAllReduce(object):
    """All-reduce (mean) of every gradient of `module` through one persistent flat fp32 bucket."""

    def __init__(self, module, group=None, force=False):
        """force=True (or TTRNN_FORCE_COLLECTIVES=1): run the bucket copies and the all-reduce even when the group has ONE
        rank, so that a 1-GPU box drives the exact RCCL call sequence of the multi-GPU step (tests, bench.py)."""
        import os
        self.group = group
        self.force = bool(force) or os.environ.get("TTRNN_FORCE_COLLECTIVES") == "1"
        self.params = [p for p in module.parameters() if p.requires_grad]
        # nn.Module.parameters() de-duplicates shared tensors (e.g. TTLinearSet's gate{i} / gates.{i})
        self.sizes = [p.numel() for p in self.params]
        total = sum(self.sizes)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # optional device timing of the collective (bench.py: so that the first multi-GPU run explains its own efficiency):
        # set `timer` to an object with start(name) / stop(name) recording events on the current stream
        self.timer = None

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def _views(self):
        """One view of the flat bucket per parameter, laid out like that parameter's gradient (TT cores are dense but
        permuted: t3nsor/tensor_train.py:104-114), so that bucket <-> gradient copies are plain element-wise copies."""
        views, grads, off = [], [], 0
        for p, n in zip(self.params, self.sizes):
            if p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)      # missing gradients count as zero
            g = p.grad
            dense = g.is_contiguous() or g.permute(*sorted(range(g.dim()), key=lambda d: -g.stride(d))).is_contiguous()
            if not dense:
                p.grad = g = g.contiguous()
            views.append(torch.as_strided(self.flat, g.shape, g.stride(), off))
            grads.append(g)
            off += n
        return views, grads

    def sync(self):
        """Call after backward(): grads become the mean over ranks.  Missing grads count as zero.  Two multi-tensor
        copies (torch._foreach_copy_: one launch each, whatever the number of parameters) around ONE all-reduce."""
        if self.world == 1 and not (self.force and dist.is_initialized()):
            return
        views, grads = self._views()
        if self.timer is not None:
            self.timer.start("grad_sync")
        torch._foreach_copy_(views, grads)
        if self.timer is not None:
            self.timer.start("all_reduce")
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.timer is not None:
            self.timer.stop("all_reduce")
        self.flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, views)
        if self.timer is not None:
            self.timer.stop("grad_sync")
