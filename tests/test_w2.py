"""The first-tier kernels of the reference's own published encoder shape — TT-LSTM hidden 768, two cores, rank 2, 40 inputs
(experiments/speaker_verification/encoder/params_model.py:2-4,14-16; tensorized-rnn_amd/csrc/ttrnn_fast_w2.hip): forward with the
input projection inside the recurrent kernel, reverse-time kernel with wave-local transposed stages.  Through the module API and
the C ABI, against the oracle (float64 where operand ranges are stretched), against the runtime tier they replace (option dev2
bits 4 / 5), and against themselves bit for bit."""
import pytest
import torch

from test_gpu_parity import _maxabs, _oracle_forward, build_module, dev

pytestmark = pytest.mark.gpu

META = dict(kind="ttlstm", input_size=40, hidden_size=768, num_layers=1, n_cores=2, tt_rank=2)


def _routes(m, B, T):
    from ttrnn_hip import functional as F
    spec = m._all_layers[0]._layer_spec()
    return F.rnn_route(spec, B, T), F.rnn_backward_route(spec, B, T)


def test_routes():
    import ttrnn_hip
    torch.manual_seed(1)
    m = build_module(META, dev())
    assert _routes(m, 8, 12) == ("fused_core", "fused_core")
    with ttrnn_hip.option("dev2", 16 | 32):
        assert _routes(m, 8, 12) == ("runtime_mfma", "runtime_mfma")
    with ttrnn_hip.fp32_math("exact"):
        assert _routes(m, 8, 12)[0] != "fused_core"
    # ranks up to 4 are this kernel's (the rank-4 instantiation; rank 3 fills its fourth slot with zeros); rank 8, another cell: not
    for r in (1, 3, 4):
        assert _routes(build_module(dict(META, tt_rank=r), dev()), 8, 12) == ("fused_core", "fused_core"), r
    m8 = build_module(dict(META, tt_rank=8), dev())
    assert _routes(m8, 8, 12) == ("runtime_mfma", "runtime_mfma")
    # the four-core models of the reference's result tables at this size: cores contracted pairwise onto the same kernels
    for r in (2, 3, 4):
        assert _routes(build_module(dict(META, n_cores=4, tt_rank=r), dev()), 8, 12) == ("fused_core", "fused_core"), r
    assert _routes(build_module(dict(META, n_cores=4, tt_rank=8), dev()), 8, 12) == ("runtime_mfma", "runtime_mfma")
    assert _routes(build_module(dict(META, n_cores=3, tt_rank=2), dev()), 8, 12)[0] == "runtime_mfma"
    # the TT-GRU of the same layer (speaker_encoder.py `use_gru`): k_gru_fwd_w2 / k_gru_bwd_w2 (ranks up to 4)
    mg = build_module(dict(META, kind="ttgru"), dev())
    assert _routes(mg, 8, 12) == ("fused_core", "fused_core")
    assert _routes(build_module(dict(META, kind="ttgru", tt_rank=4), dev()), 8, 12) == ("fused_core", "fused_core")
    assert _routes(build_module(dict(META, kind="ttgru", tt_rank=8), dev()), 8, 12) == ("runtime_mfma", "runtime_mfma")
    with ttrnn_hip.option("dev2", 16 | 32):
        assert _routes(mg, 8, 12) == ("runtime_mfma", "runtime_mfma")
    # ... and its four-core variants ((6, 6, 8, 8) contracted pairwise: (36, 64))
    for r in (2, 3, 4):
        assert _routes(build_module(dict(META, kind="ttgru", n_cores=4, tt_rank=r), dev()), 8, 12) == ("fused_core", "fused_core"), r
    assert _routes(build_module(dict(META, kind="ttgru", n_cores=4, tt_rank=8), dev()), 8, 12) == ("runtime_mfma", "runtime_mfma")


CASES = ["fresh", "tiny_weights", "huge_weights", "huge_h0", "zero_core", "mixed_magnitudes", "x_ranges", "no_bias", "rank4", "rank3", "rank1",
         "d4rank2", "d4rank4", "d4rank3_x_ranges", "d4rank2_huge_h0"]


@pytest.mark.parametrize("case", CASES)
def test_forward_operand_ranges(case):
    """k_lstm_fwd_w2 on two fp16 pieces per operand: power-of-two scales per launch for the cores, per sample for a caller's h_0,
    per STEP for x_t (its own maximum) and for the stage hand-off.  160 steps for the fresh model, short runs for the stretched
    operands; float64 oracle; the runtime tier (dev2 bit 4) as the second opinion; repeat launches and batch splits bit for bit."""
    import ttrnn_hip
    torch.manual_seed(29)
    meta = dict(META, bias=False) if case == "no_bias" else META
    if case.startswith("rank"):
        meta = dict(META, tt_rank=int(case[4:]))
    if case.startswith("d4rank"):
        meta = dict(META, n_cores=4, tt_rank=int(case[6]))
        case = case[8:] or "d4"
    m = build_module(meta, dev())
    T = 160 if case in ("fresh", "rank4", "d4") else 7
    B = 5
    g = torch.Generator().manual_seed(31)
    x = torch.rand(B, T, 40, generator=g) if case == "fresh" else torch.randn(B, T, 40, generator=g)
    h0, c0 = torch.randn(B, 768, generator=g) * 0.3, torch.randn(B, 768, generator=g) * 0.3
    with torch.no_grad():
        hid = [p for n, p in m.named_parameters() if "hidden_weights.parameters" in n]
        inp = [p for n, p in m.named_parameters() if "input_weights.parameters" in n]
        assert len(hid) == meta["n_cores"] and len(inp) == meta["n_cores"]
        if case == "tiny_weights":
            for p in hid + inp:
                p.mul_(1e-5)
        elif case == "huge_weights":
            for p in hid + inp:
                p.mul_(6.0)
        elif case == "huge_h0":
            h0 = torch.randn(B, 768, generator=g) * torch.tensor([0.1, 3.0, 40.0, 500.0, 6000.0]).view(B, 1)
        elif case == "zero_core":
            hid[0].zero_()
        elif case == "mixed_magnitudes":
            for p, step, f in ((hid[1], 3, 1e-6), (hid[0], 2, 1e-5), (inp[1], 5, 1e-4)):
                w = p.detach().clone().reshape(-1)
                w[::step] *= f
                p.copy_(w.reshape(p.shape))
        elif case == "x_ranges":
            # every step has its own scale: frames of 1e4, of 1e-6, all-zero frames, one outlier channel
            x = x * torch.tensor([1e4, 1.0, 1e-6, 0.0, 30.0, 1e-3, 1.0]).view(1, T, 1)
            x[:, 4, 7] = 3e3
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    r64, _, c64 = _oracle_forward("ttlstm", sd, 1, x.double(), (h0.double(), c0.double()))
    scale = max(1e-30, float(c64.abs().max()), float(r64.abs().max()))
    xd, hd, cd = x.to(dev()), h0.to(dev()), c0.to(dev())
    # saturated gates amplify one fp32 ulp of a pre-activation into the state: huge operands get the tolerance every kernel gets there
    tol = 2e-3 if case in ("huge_weights", "huge_h0", "x_ranges") else 2e-6
    with torch.no_grad():
        assert _routes(m, B, T)[0] == "fused_core"
        out, (hT, cT) = m(xd, (hd, cd))
        again, _ = m(xd, (hd, cd))
        assert torch.equal(out, again)                               # repeat launch
        part, _ = m(xd[1:3], (hd[1:3], cd[1:3]))
        assert torch.equal(out[1:3], part)                           # samples never interact
        assert torch.equal(out[:, -1], hT)
        nost = m(xd)[0]                                              # zero initial state
        r0, _, _ = _oracle_forward("ttlstm", sd, 1, x.double())
        assert _maxabs(nost, r0) <= tol * max(1.0, float(r0.abs().max()))
        with ttrnn_hip.option("dev2", 16):
            assert _routes(m, B, T)[0] == "runtime_mfma"
            tier, (_, tier_c) = m(xd, (hd, cd))
    assert torch.isfinite(out).all() and torch.isfinite(cT).all(), case
    err = max(_maxabs(out, r64), _maxabs(cT, c64))
    err_tier = max(_maxabs(tier, r64), _maxabs(tier_c, c64))
    print(case, "encoder-shape kernel: max abs error vs float64 (state scale %.3g): %.3g; runtime tier: %.3g" % (scale, err, err_tier))
    assert err <= tol * max(1.0, scale)
    if tol > 1e-5:      # saturated regime: one ulp of a pre-activation of size 1e3 ... 1e4 decides a unit; same class as the tier, not 3 x it
        assert err <= 10.0 * err_tier + 1e-4 * max(1.0, scale)
    else:
        assert err <= 3.0 * err_tier + 3e-7 * max(1.0, scale)


@pytest.mark.parametrize("B,T,init,dout,rank,d", [(3, 1, True, True, 2, 2), (2, 2, False, True, 2, 2), (5, 9, True, True, 2, 2), (4, 6, True, False, 2, 2),
                                                   (3, 5, False, False, 2, 2), (4, 7, True, True, 4, 2), (3, 6, False, True, 3, 2), (2, 3, True, False, 1, 2),
                                                   (3, 6, True, True, 2, 4), (4, 5, False, True, 4, 4), (2, 2, True, False, 3, 4)])
def test_training_step_vs_oracle(B, T, init, dout, rank, d):
    """forward (reserve records) + k_lstm_bwd_w2 + the chain weight gradients: every gradient against the oracle's autograd
    (1e-4 of each tensor's maximum), and against the runtime tier's reverse kernel reading the SAME records"""
    import ttrnn_hip
    from oracle import ttrnn_oracle as O
    torch.manual_seed(41)
    m = build_module(dict(META, tt_rank=rank, n_cores=d), dev())
    g = torch.Generator().manual_seed(43)
    x = torch.randn(B, T, 40, generator=g)
    h0 = torch.randn(B, 768, generator=g) * 0.5 if init else None
    c0 = torch.randn(B, 768, generator=g) * 0.5 if init else None
    w = torch.randn(B, T, 768, generator=g)
    wh, wc = torch.randn(B, 768, generator=g), torch.randn(B, 768, generator=g)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    layers, leaves = O.layers_from_state_dict(sd, 1, requires_grad=True)
    xr = x.clone().requires_grad_(True)
    st = None
    if init:
        h0r, c0r = h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
        st = (h0r, c0r)
    ro, (rh, rc) = O.lstm_forward(layers, xr, st)
    loss = (rh * wh).sum() + (rc * wc).sum()
    if dout:
        loss = loss + (ro * w).sum()
    loss.backward()

    def run():
        m.zero_grad()
        xg = x.to(dev()).requires_grad_(True)
        stg = None
        if init:
            stg = (h0.to(dev()).requires_grad_(True), c0.to(dev()).requires_grad_(True))
        out, (hT, cT) = m(xg, stg)
        ls = (hT * wh.to(dev())).sum() + (cT * wc.to(dev())).sum()
        if dout:
            ls = ls + (out * w.to(dev())).sum()
        ls.backward()
        grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
        grads["x"] = xg.grad.cpu()
        if init:
            grads["h0"], grads["c0"] = stg[0].grad.cpu(), stg[1].grad.cpu()
        return out.detach().cpu(), grads

    assert _routes(m, B, T) == ("fused_core", "fused_core")
    out, got = run()
    assert _maxabs(out, ro.detach()) <= 1e-5
    ref = {n: leaves[n].grad for n in leaves}
    ref["x"] = xr.grad
    if init:
        ref["h0"], ref["c0"] = h0r.grad, c0r.grad
    for n, r in ref.items():
        assert _maxabs(got[n], r) <= 1e-4 * max(1e-3, float(r.abs().max())), n
    with ttrnn_hip.option("dev2", 32):
        _, tier = run()
    for n, r in ref.items():
        assert _maxabs(got[n], tier[n]) <= 2e-5 * max(1e-3, float(r.abs().max())), n
    _, again = run()
    for n in got:
        # bitwise repeatable: everything the recurrent kernels and the chain weight-gradient kernel produce.  (The input matrix of
        # a layer whose INPUT is differentiated over fewer than 4 x in rows goes through the per-row kernels, atomics by design:
        # DESIGN.md section 9.)
        # (ranks 1 and 3 have no chain weight-gradient plan: at these few rows their gradients take the per-row kernels as well)
        if "input_weights" not in n and ((rank in (2, 4) and d == 2) or "weights" not in n):
            assert torch.equal(got[n], again[n]), n


@pytest.mark.parametrize("scale", [1e-12, 1e-4, 1e6, 1e14])
def test_reverse_kernel_gradient_ranges(scale):
    """the gate gradients are split under each WAVE's own maximum per step: the same relative error over 26 decades of output
    gradients (float64 oracle)"""
    from oracle import ttrnn_oracle as O
    torch.manual_seed(47)
    m = build_module(META, dev())
    g = torch.Generator().manual_seed(53)
    B, T = 3, 8
    x = torch.randn(B, T, 40, generator=g)
    # output gradients that grow by a decade per step on top of the overall scale
    w = torch.randn(B, T, 768, generator=g) * (10.0 ** torch.arange(T, dtype=torch.float32)).view(1, T, 1) * scale
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    layers, leaves = O.layers_from_state_dict(sd, 1, requires_grad=True, dtype=torch.float64)
    ro, _ = O.lstm_forward(layers, x.double())
    (ro * w.double()).sum().backward()
    out, _ = m(x.to(dev()))
    (out * w.to(dev())).sum().backward()
    for n, p in m.named_parameters():
        r = leaves[n].grad
        assert _maxabs(p.grad, r) <= 3e-6 * float(r.abs().max()), (scale, n)


GRU_CASES = ["fresh", "tiny_weights", "huge_weights", "huge_h0", "zero_core", "x_ranges", "no_bias", "rank4", "rank3", "rank1",
             "d4rank2", "d4rank4", "d4rank3_x_ranges", "d4rank2_huge_h0"]


@pytest.mark.parametrize("case", GRU_CASES)
def test_gru_forward_operand_ranges(case):
    """k_gru_fwd_w2 (round 6): the encoder layer as a TT-GRU (speaker_encoder.py:33-47 `use_gru`; gru.py:33-44) — three waves per sample,
    four accumulator rows per unit (r, z, the hidden and the input part of n kept apart), the state image under the exponent of the
    previous step's exact maximum (a GRU's large h_0 decays only as fast as z lets it).  Same cases and yardsticks as the LSTM
    kernel's test: float64 oracle, the runtime tier (dev2 bit 4) as the second opinion, repeat launches and batch splits bit for bit."""
    import ttrnn_hip
    torch.manual_seed(37)
    base = dict(META, kind="ttgru")
    meta = dict(base, bias=False) if case == "no_bias" else base
    if case.startswith("rank"):
        meta = dict(base, tt_rank=int(case[4:]))
    if case.startswith("d4rank"):
        meta = dict(base, n_cores=4, tt_rank=int(case[6]))
        case = case[8:] or "d4"
    m = build_module(meta, dev())
    T = 160 if case in ("fresh", "rank4", "d4") else 9
    B = 5
    g = torch.Generator().manual_seed(41)
    x = torch.rand(B, T, 40, generator=g) if case == "fresh" else torch.randn(B, T, 40, generator=g)
    h0 = torch.randn(B, 768, generator=g) * 0.3
    with torch.no_grad():
        hid = [p for n, p in m.named_parameters() if "hidden_weights.parameters" in n]
        inp = [p for n, p in m.named_parameters() if "input_weights.parameters" in n]
        if case == "tiny_weights":
            for p in hid + inp:
                p.mul_(1e-5)
        elif case == "huge_weights":
            for p in hid + inp:      # (x 6 as for the LSTM makes a GRU chaotic: the tier itself then misses float64 by 0.17)
                p.mul_(2.5)
        elif case == "huge_h0":
            h0 = torch.randn(B, 768, generator=g) * torch.tensor([0.1, 3.0, 40.0, 500.0, 6000.0]).view(B, 1)
        elif case == "zero_core":
            hid[0].zero_()
        elif case == "x_ranges":
            x = x * torch.tensor([1e4, 1.0, 1e-6, 0.0, 30.0, 1e-3, 1.0, 1.0, 1e2]).view(1, T, 1)
            x[:, 4, 7] = 3e3
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    r64 = _oracle_forward("ttgru", sd, 1, x.double(), h0.double())[0]
    scale = max(1e-30, float(r64.abs().max()))
    xd, hd = x.to(dev()), h0.to(dev())
    tol = 2e-3 if case in ("huge_weights", "huge_h0", "x_ranges") else 2e-6
    with torch.no_grad():
        assert _routes(m, B, T)[0] == "fused_core"
        out, hT = m(xd, hd)
        again, _ = m(xd, hd)
        assert torch.equal(out, again)
        part, _ = m(xd[1:3], hd[1:3])
        assert torch.equal(out[1:3], part)
        assert torch.equal(out[:, -1], hT.reshape(B, 768))
        nost = m(xd)[0]
        r0 = _oracle_forward("ttgru", sd, 1, x.double())[0]
        fin = m(xd, hd, need_outputs=False) if "need_outputs" in m.forward.__code__.co_varnames else None
        with ttrnn_hip.option("dev2", 16):
            assert _routes(m, B, T)[0] == "runtime_mfma"
            tier, _ = m(xd, hd)
            tier0 = m(xd)[0]
        e0, e0t = _maxabs(nost, r0), _maxabs(tier0, r0)
        print(case, "zero initial state: %.3g (tier %.3g)" % (e0, e0t))
        assert e0 <= tol * max(1.0, float(r0.abs().max())) or e0 <= 10.0 * e0t + 1e-4
    assert torch.isfinite(out).all(), case
    if fin is not None:
        assert torch.equal(fin[1].reshape(B, 768), hT.reshape(B, 768))
    err, err_tier = _maxabs(out, r64), _maxabs(tier, r64)
    print(case, "TT-GRU encoder-shape kernel: max abs error vs float64 (state scale %.3g): %.3g; runtime tier: %.3g" % (scale, err, err_tier))
    assert err <= tol * max(1.0, scale)
    if tol > 1e-5:
        assert err <= 10.0 * err_tier + 1e-4 * max(1.0, scale)
    else:
        assert err <= 3.0 * err_tier + 3e-7 * max(1.0, scale)


@pytest.mark.parametrize("B,T,h0_scale,dout,rank,d", [(4, 12, 0.3, "plain", 2, 2), (3, 1, 0.3, "plain", 2, 2), (2, 33, 50.0, "plain", 2, 2), (5, 8, None, "plain", 2, 2),
                                                       (4, 20, 0.3, "decades", 2, 2), (4, 24, 0.3, "last_step_only", 2, 2), (5, 9, 0.3, "sparse", 2, 2),
                                                       (3, 10, 0.3, "plain", 4, 2), (3, 7, None, "decades", 3, 2), (300, 3, 0.3, "plain", 2, 2),
                                                       (4, 12, 0.3, "plain", 2, 4), (3, 9, 0.3, "decades", 4, 4), (5, 9, 0.3, "sparse", 3, 4), (2, 20, 50.0, "plain", 2, 4)])
def test_gru_training_step(B, T, h0_scale, dout, rank, d):
    """k_gru_fwd_w2 + k_gru_bwd_w2 (wave-local transposed stages, three gates' hidden-chain gradients in the forward's four slots, the
    direct path dh z in registers): every gradient of a training step against the float64 oracle's autograd, and against the step
    that runs on the tier in both directions (dev2 bits 4, 5); output gradients over ten decades, a loss on the last step only,
    steps and samples without gradient; the reverse kernel's own results bit for bit on a repeat."""
    import ttrnn_hip
    from oracle import ttrnn_oracle as O
    torch.manual_seed(43)
    m = build_module(dict(META, kind="ttgru", tt_rank=rank, n_cores=d), dev())
    x = torch.randn(B, T, 40)
    h0 = None if h0_scale is None else torch.randn(B, 768) * h0_scale
    w = torch.randn(B, T, 768)
    if dout == "decades":
        w = w * (10.0 ** (torch.rand(B, T, 1) * 10 - 6))
    elif dout == "last_step_only":
        w[:, :-1] = 0.0
    elif dout == "sparse":
        w[:, 1:5] = 0.0
        w[2] = 0.0
    assert _routes(m, B, T) == ("fused_core", "fused_core")
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    layers, leaves = O.layers_from_state_dict(sd, 1, requires_grad=True, dtype=torch.float64)
    xr = x.double().clone().requires_grad_(True)
    h0r = None if h0 is None else h0.double().clone().requires_grad_(True)
    ro, rh = O.gru_forward(layers, xr, h0r)
    wsum = 0.0 if dout in ("sparse", "last_step_only") else 0.5
    ((ro * w.double()).sum() + wsum * rh.sum()).backward()

    def run():
        m.zero_grad()
        xg = x.to(dev()).requires_grad_(True)
        h0g = None if h0 is None else h0.to(dev()).requires_grad_(True)
        out, hT = m(xg, h0g)
        ((out * w.to(dev())).sum() + wsum * hT.sum()).backward()
        g = {"x": xg.grad.clone(), **{n: p.grad.detach().clone() for n, p in m.named_parameters()}}
        if h0g is not None:
            g["h0"] = h0g.grad.clone()
        return out.detach().clone(), g

    out, got = run()
    _, again = run()
    for n in ("x",) + (("h0",) if h0 is not None else ()):
        assert torch.equal(got[n], again[n]), n
    with ttrnn_hip.option("dev2", 16 | 32):
        assert _routes(m, B, T) == ("runtime_mfma", "runtime_mfma")
        out_t, old = run()
    assert _maxabs(out, ro.detach()) <= (2e-3 if (h0_scale or 0) > 1 else 2e-6) * max(1.0, float(ro.abs().max()))
    refs = {"x": xr.grad, **{n: leaves[n].grad for n, _ in m.named_parameters()}}
    if h0r is not None:
        refs["h0"] = h0r.grad
    worst, worst_t = 0.0, 0.0
    for n, ref in refs.items():
        sc = max(float(ref.abs().max()), 1e-30)
        assert torch.isfinite(got[n]).all(), n
        worst = max(worst, _maxabs(got[n].double(), ref) / sc)
        worst_t = max(worst_t, _maxabs(old[n].double(), ref) / sc)
    print("TT-GRU training step", B, T, h0_scale, dout, rank, d, "max gradient error relative to each tensor's maximum: %.3g (tier: %.3g)" % (worst, worst_t))
    tol = 2e-3 if (h0_scale or 0) > 1 else 1e-4
    assert worst <= tol and worst <= 3.0 * worst_t + 1e-5
    if dout == "sparse" and h0 is not None:
        assert float(got["h0"][2].abs().max()) == 0.0
