"""Worker of tests/test_dist_gloo.py::test_rccl_one_rank_group_drives_every_collective (run as a fresh process on the GPU
box): a ONE-rank process group on the `nccl` backend (= RCCL on ROCm) through which every collective of the data-parallel
training step runs on device tensors — broadcast of the strided TT parameters, the flat-bucket gradient all-reduce (forced
past its world == 1 early-out), the all-gather of the GE2E embeddings, barrier — checked against the no-collective results."""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorized-rnn_amd"), os.path.join(ROOT, "examples")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    calls = {"all_reduce": 0, "all_gather": 0, "broadcast": 0}
    for name in list(calls):
        orig = getattr(dist, name)

        def counted(*a, _o=orig, _n=name, **k):
            calls[_n] += 1
            return _o(*a, **k)
        setattr(dist, name, counted)
    try:
        from models import MNISTClassifier
        from ttrnn_hip import ge2e
        from ttrnn_hip.dist import FlatGradAllReduce, broadcast_parameters
        torch.manual_seed(5)
        with contextlib.redirect_stdout(io.StringIO()):
            model = MNISTClassifier(1, 10, 256, 1, dev, gru=False, n_cores=3, tt_rank=8).to(dev)
        before = [p.detach().clone() for p in model.parameters()]
        broadcast_parameters(model, src=0)
        assert all(torch.equal(a, p.detach()) and a.stride() == p.stride() for a, p in zip(before, model.parameters()))
        x = torch.rand(8, 24, 1, device=dev)
        target = torch.randint(0, 10, (8,), device=dev)
        loss = torch.nn.functional.nll_loss(model(x).float(), target)
        loss.backward()
        grads = [p.grad.detach().clone() for p in model.parameters()]
        red = FlatGradAllReduce(model, force=True)
        assert red.world == 1
        red.sync()                               # copies -> RCCL all-reduce of the flat bucket -> mean -> copies
        assert calls["all_reduce"] == 1
        for g, p in zip(grads, model.parameters()):
            assert torch.equal(g, p.grad) and p.grad.stride() == p.stride()
        # GE2E: all-gather of the [S, U, D] embeddings along the speaker axis, gradient back to the local slice
        torch.manual_seed(6)
        emb = torch.nn.functional.normalize(torch.randn(6, 5, 32, device=dev), dim=2)
        w = torch.tensor(10.0, device=dev, requires_grad=True)
        b = torch.tensor(-5.0, device=dev, requires_grad=True)
        e1 = emb.clone().requires_grad_(True)
        l1, _ = ge2e.ge2e_loss(e1, w, b, None, with_eer=False)
        l1.backward()
        gw, gb = w.grad.clone(), b.grad.clone()
        w.grad = b.grad = None
        e2 = emb.clone().requires_grad_(True)
        l2, _ = ge2e.ge2e_loss_data_parallel(e2, w, b, with_eer=False, force_gather=True)
        l2.backward()
        assert calls["all_gather"] == 1
        assert torch.equal(l1, l2) and torch.equal(e1.grad, e2.grad) and torch.equal(gw, w.grad) and torch.equal(gb, b.grad)
        t = torch.ones(3, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        print("RCCL_ONE_RANK_OK", calls)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
