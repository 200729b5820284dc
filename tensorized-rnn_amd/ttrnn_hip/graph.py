"""hipGraph capture of a whole training step (opt-in).

The reference's training step (experiments/digit_classification/benchmarking.py:41-70: zero_grad, forward, nll_loss,
backward, Adam) is ~30 ... 110 library launches plus torch's own, issued through ctypes and autograd from one Python
thread: for short sequences / small shapes the step is bound by the HOST (about 1.3 ms per step whatever the shape,
profiles/r3/bench_grid_train.json), not by the kernels.  Every entry point of libttrnn is capture safe by contract
(include/ttrnn.h: no allocation, no synchronisation, no host <-> device copy inside a call; launches go to the caller's
stream), so the step can be recorded ONCE into a hipGraph and replayed with one host call.

    step = CapturedTrainStep(model, optimizer, loss_fn, (x, target))
    for x, target in batches:
        loss = step(x, target)          # copies the batch into the static inputs, replays the graph

What capture requires from the caller:
  * fixed shapes and dtypes (one CapturedTrainStep per input shape);
  * an optimizer whose step stays on the device: torch.optim.Adam(..., capturable=True) (optionally fused=True);
  * nothing in `loss_fn` that synchronises (`.item()`, printing a tensor, boolean tests on device values).
The gradients (`p.grad`) and the returned loss are STATIC tensors that every replay overwrites; clone what must survive
the next step.  Parameters are updated in place, so `model` is a normal module before, between and after replays.
Not captured: ttrnn_hip.device_status() (it synchronises by design) — call it between replays.
"""
import torch


def adam_for_capture(params, lr=1e-3, **kw):
    """torch.optim.Adam in the form a captured step needs (device-side step counter); fused where torch offers it."""
    try:
        return torch.optim.Adam(params, lr=lr, capturable=True, fused=True, **kw)
    except (RuntimeError, TypeError, ValueError):
        return torch.optim.Adam(params, lr=lr, capturable=True, **kw)


class CapturedTrainStep(object):
    def __init__(self, model, optimizer, loss_fn, example_inputs, warmup=3, after_backward=None):
        """optimizer: a capturable optimizer (adam_for_capture), or None to capture forward + backward only.
        loss_fn(model, *inputs) -> scalar loss tensor.  example_inputs: device tensors of the step's shapes (their values
        are consumed by `warmup` real optimizer steps: the warm-up steps TRAIN, exactly as eager steps would).
        after_backward: optional callable run between backward() and optimizer.step() inside the graph (gradient
        clipping / scaling, speaker_encoder.py:60-66, or a gradient all-reduce)."""
        if not torch.cuda.is_available():
            raise RuntimeError("CapturedTrainStep needs a GPU")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.after_backward = after_backward
        self.static_inputs = [t.clone() for t in example_inputs]
        dev = self.static_inputs[0].device
        # warm-up on a side stream (torch's capture recipe): lazy state — optimizer moments, cached descriptors, the library's
        # per-kernel attributes and occupancy answers, torch's autograd threads — is created before the capture begins
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                self._eager_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # gradients are allocated INSIDE the capture (private pool): replays then overwrite the same buffers
        self._zero_grad()
        with torch.cuda.graph(self.graph):
            self.static_loss = self._eager_step(zero=False)
        self.replays = 0
        # the graph holds the kernels the library's options selected AT CAPTURE TIME: a later set_option / set_fp32_math would
        # silently keep replaying the old route (an A/B under --graph would measure nothing) — remembered and checked per replay
        from . import _lib
        self._options_epoch = _lib.OPTIONS_EPOCH

    def _eager_step(self, zero=True):
        if zero:
            self._zero_grad()
        loss = self.loss_fn(self.model, *self.static_inputs)
        loss.backward()
        if self.after_backward is not None:
            self.after_backward()
        if self.optimizer is not None:
            self.optimizer.step()
        return loss

    def _zero_grad(self):
        if self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=True)
        else:
            self.model.zero_grad(set_to_none=True)

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_inputs):
            raise ValueError("expected {} inputs, got {}".format(len(self.static_inputs), len(inputs)))
        for dst, src in zip(self.static_inputs, inputs):
            if src is not dst:
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError("a captured step is bound to its input shapes: expected {} {}, got {} {}".format(
                        tuple(dst.shape), dst.dtype, tuple(src.shape), src.dtype))
                dst.copy_(src, non_blocking=True)
        from . import _lib
        if _lib.OPTIONS_EPOCH != self._options_epoch:
            raise _lib.TtrnnError("a library option or the fp32 math mode changed after this step was captured: the graph still "
                                  "holds the kernels of the old setting — capture a new CapturedTrainStep")
        self.graph.replay()
        self.replays += 1
        return self.static_loss
