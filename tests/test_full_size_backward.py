"""Gradient parity at the sizes `bench.py --mode train` times ("Next round" 1).

At >= 4 * in rows the weight gradients leave the per-row kernels for the dense split-K GEMMs with per-XCD row ranges
(ttrnn_fast_gemm.hip), dx runs as a GEMM under the reverse kernel's row maxima, cfg5 runs its pair kernels: code that only
executes at the full batch.  The oracle cannot run a full batch through float64 autograd for every configuration, so the
loss is MASKED to a few batch rows:

    loss = sum_{b in rows} (out[b] * w[b]).sum() + (hT[rows] * v_h).sum() + (cT[rows] * v_c).sum()

Every other sample receives a zero output gradient: every parameter gradient then depends on `rows` only — while every
workgroup, row range, split-K partial and workgroup pair of the full-size launch still runs, and anything they add for the
other samples must be exactly nothing.  Compared with the oracle's float64 autograd on those rows alone
(tests/bptt_oracle.py): every core / bias gradient, dx[rows], dh0 / dc0[rows]; dx, dh0, dc0 of the masked-out samples must
be exactly 0.  cfg2 additionally runs UNMASKED on its full batch (one layer: the oracle's BPTT is replayed in segments).

Reference: experiments/digit_classification/benchmarking.py:41-70 (the step that is timed), tensorized_rnn/lstm.py:101-135.
Tolerance: 1e-4 of each tensor's max magnitude; bf16 storage: 5e-2.
"""
import pytest
import torch

from bptt_oracle import masked_loss_grads
from golden_io import build_module

pytestmark = pytest.mark.gpu

CONFIGS = {
    # kind, in, H, L, d, r, B, T, storage dtype, rows (first / last sample, both sides of the batch's middle — the split-K
    # row ranges of the dense gradients cut the B*T rows into 8 XCD ranges — and an odd one), segment length of the oracle replay
    "cfg2": ("ttlstm", 1, 256, 1, 3, 8, 64, 784, torch.float32, [0, 31, 32, 63], 112),
    "cfg3": ("ttgru", 1, 256, 1, 3, 8, 256, 784, torch.bfloat16, [0, 127, 255], 112),
    "cfg3_fp32": ("ttgru", 1, 256, 1, 3, 8, 256, 784, torch.float32, [0, 128, 255], 112),
    "cfg4": ("ttlstm", 40, 256, 3, 3, 16, 512, 160, torch.float32, [0, 255, 256, 511], 160),
    "cfg5": ("ttlstm", 1024, 1024, 1, 4, 32, 128, 1024, torch.float32, [0, 127], 32),
}
CASES = [("cfg2", "split"), ("cfg2", "exact"), ("cfg3", None), ("cfg3_fp32", "split"), ("cfg4", "split"), ("cfg4", "exact"),
         ("cfg5", "split")]


def dev():
    return torch.device("cuda:0")


_ORACLE_CACHE = {}


def _oracle(key, *args, **kwargs):
    """The float64 oracle gradients depend on the seeded inputs only, not on the library's math mode: computed once per
    (configuration, variant) and shared by the split / exact cases."""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = masked_loss_grads(*args, **kwargs)
    return _ORACLE_CACHE[key]


def _rel(got, ref):
    ref = ref.double()
    return float((got.detach().double().cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


def _model(name):
    kind, inp, H, L, d, r, B, T, dtype, rows, seg = CONFIGS[name]
    torch.manual_seed(1111)
    m = build_module(dict(kind=kind, input_size=inp, hidden_size=H, num_layers=L, n_cores=d, tt_rank=r), dev()).to(dtype)
    return m.train()


def _product_step(m, lstm, x, W, VH, VC, h0=None, c0=None, x_grad=True):
    """forward + backward of the masked loss on the full batch.  W [B,T,H] / VH / VC [B,H] are zero outside `rows`."""
    m.zero_grad(set_to_none=True)
    xg = x.clone().requires_grad_(x_grad)
    hg = h0.clone().requires_grad_(True) if h0 is not None else None
    cg = c0.clone().requires_grad_(True) if (c0 is not None and lstm) else None
    if lstm:
        out, (hT, cT) = m(xg, None if hg is None else (hg, cg))
        loss = (out.float() * W).sum() + (hT.float() * VH).sum() + (cT.float() * VC).sum()
    else:
        out, hT = m(xg, hg)
        loss = (out.float() * W).sum() + (hT.float() * VH).sum()
    loss.backward()
    torch.cuda.synchronize()
    return dict(dx=xg.grad, dh0=None if hg is None else hg.grad, dc0=None if cg is None else cg.grad, out=out.detach(),
                params={k: p.grad for k, p in m.named_parameters()})


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name,math", CASES)
def test_masked_loss_gradients_at_bench_size(name, math):
    import ttrnn_hip
    kind, inp, H, L, d, r, B, T, dtype, rows, seg = CONFIGS[name]
    lstm = kind == "ttlstm"
    tol = 5e-2 if dtype == torch.bfloat16 else 1e-4
    m = _model(name)
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(4242)
    x = torch.rand(B, T, inp, generator=g).to(dtype)
    n = len(rows)
    w = torch.randn(n, T, H, generator=g)
    vh, vc = torch.randn(n, H, generator=g), torch.randn(n, H, generator=g)
    h0 = (0.5 * torch.tanh(torch.randn(B, H, generator=g))).to(dtype)
    c0 = (0.5 * torch.randn(B, H, generator=g)).to(dtype)
    W = torch.zeros(B, T, H)
    W[rows] = w
    VH, VC = torch.zeros(B, H), torch.zeros(B, H)
    VH[rows], VC[rows] = vh, vc
    W, VH, VC = W.to(dev()), VH.to(dev()), VC.to(dev())
    others = torch.ones(B, dtype=torch.bool)
    others[rows] = False

    with (ttrnn_hip.fp32_math(math) if math else _null()):
        # (A) exactly the call of the training benchmark: zero initial state, the input needs no gradient
        got = _product_step(m, lstm, x.to(dev()), W, VH, VC, x_grad=False)
        ref = _oracle((name, "A"), kind, sd, L, x[rows].float(), w, vh, vc if lstm else None, seg=seg)
        assert _rel(got["out"][rows].float(), ref["out"]) <= (2e-2 if dtype == torch.bfloat16 else 1e-4)
        worst = {}
        for k, p in got["params"].items():
            assert p is not None and torch.isfinite(p.float()).all(), k
            worst[k] = _rel(p.float(), ref["params"][k])
        print("%s/%s (A) worst relative gradient error: %.3g (%s)" % (name, math, max(worst.values()), max(worst, key=worst.get)))
        assert max(worst.values()) <= tol, worst
        # (B) everything differentiable: input, caller's h0 / c0 (shared by all layers, lstm.py:120)
        got = _product_step(m, lstm, x.to(dev()), W, VH, VC, h0.to(dev()), c0.to(dev()), x_grad=True)
        ref = _oracle((name, "B"), kind, sd, L, x[rows].float(), w, vh, vc if lstm else None, h0=h0[rows].float(),
                      c0=c0[rows].float() if lstm else None, seg=seg)
        worst = {k: _rel(p.float(), ref["params"][k]) for k, p in got["params"].items()}
        worst["dx"] = _rel(got["dx"][rows].float(), ref["dx"])
        worst["dh0"] = _rel(got["dh0"][rows].float(), ref["dh0"])
        if lstm:
            worst["dc0"] = _rel(got["dc0"][rows].float(), ref["dc0"])
        print("%s/%s (B) worst relative gradient error: %.3g (%s)" % (name, math, max(worst.values()), max(worst, key=worst.get)))
        assert max(worst.values()) <= tol, worst
        # the masked-out samples contribute exactly nothing
        assert float(got["dx"][others].float().abs().max()) == 0.0
        assert float(got["dh0"][others].float().abs().max()) == 0.0
        if lstm:
            assert float(got["dc0"][others].float().abs().max()) == 0.0
    st = ttrnn_hip.device_status()
    assert st["pair_timeouts"] == 0


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("math", ["split", "exact"])
def test_cfg2_unmasked_full_batch_gradients(math):
    """The headline configuration's training arithmetic on ALL 64 x 784 rows against the float64 oracle (segmented BPTT):
    the reverse-time kernel's by-products (column maxima, input_size == 1 sums), the dense hidden-matrix gradient over
    50 176 rows in eight split-K ranges, the bias sums."""
    import ttrnn_hip
    kind, inp, H, L, d, r, B, T, dtype, rows, seg = CONFIGS["cfg2"]
    m = _model("cfg2")
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(77)
    x = torch.rand(B, T, inp, generator=g)
    w = torch.randn(B, T, H, generator=g)
    vh, vc = torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
    with ttrnn_hip.fp32_math(math):
        got = _product_step(m, True, x.to(dev()), w.to(dev()), vh.to(dev()), vc.to(dev()), x_grad=False)
        ref = _oracle(("cfg2", "full"), kind, sd, L, x, w, vh, vc, seg=56)
        worst = {k: _rel(p, ref["params"][k]) for k, p in got["params"].items()}
        print("cfg2 unmasked/%s (x without gradient): %.3g (%s)" % (math, max(worst.values()), max(worst, key=worst.get)))
        assert max(worst.values()) <= 1e-4, worst
        got = _product_step(m, True, x.to(dev()), w.to(dev()), vh.to(dev()), vc.to(dev()), x_grad=True)
        worst = {k: _rel(p, ref["params"][k]) for k, p in got["params"].items()}
        worst["dx"] = _rel(got["dx"], ref["dx"])
        assert max(worst.values()) <= 1e-4, worst


class _null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
