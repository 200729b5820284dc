"""The chain weight-gradient kernel (ttrnn_rnn_wgrad, ABI 7; tensorized-rnn_amd/csrc/ttrnn_fast_c2w.hip) called directly through the
C ABI on seeded operands, against a float64 evaluation of what it replaces: the gradients of a cell's two TTLinear calls over all
B*T rows (reference: tensorized_rnn/lstm.py:23-26, gru.py:33-36 -> t3nsor/layers.py:121-127 -> t3nsor/ops.py:78-93 under torch
autograd).  The module-level parity of the same route against reference-generated fixtures is tests/test_gpu_parity.py
(g6_bwd_spk*)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def dense_from_cores(cores):
    """cores[k]: (R_k, I_k, J_k, R_{k+1}) -> W[in][out] (t3nsor/ops.py:78-93 evaluates x -> x W core by core)."""
    t = cores[0][0]                                    # (I0, J0, R1)
    for c in cores[1:]:
        t = torch.einsum('...a,aijb->...ijb', t, c)
    t = t[..., 0]                                      # (i0, j0, i1, j1, ...)
    d = len(cores)
    perm = [2 * k + 1 for k in range(d)] + [2 * k for k in range(d)]
    t = t.permute(*perm)
    n_in = int(np.prod([c.shape[2] for c in cores]))
    return t.reshape(n_in, -1)


def make_cores(in_modes, out_modes, ranks, gen, scale=0.3):
    cores = []
    for k in range(len(in_modes)):
        # the reference's physical layout: (R, J, I, R') storage viewed as (R, I, J, R')  (t3nsor/ops.py:47-51)
        phys = torch.randn(ranks[k], in_modes[k], out_modes[k], ranks[k + 1], generator=gen) * scale
        cores.append(phys.transpose(1, 2))
    return cores


CASES = {
    # the reference's speaker encoder (params_model.py:2-4,14-16): in = 40 mel channels, H = 768, d = 2, r = 2
    "spk_lstm": dict(cell="lstm", inp=([5, 8], [48, 64], [1, 2, 1]), hid=([24, 32], [48, 64], [1, 2, 1]), H=768, B=5, T=7, h0=False, mats=3),
    "spk_lstm_h0": dict(cell="lstm", inp=([5, 8], [48, 64], [1, 2, 1]), hid=([24, 32], [48, 64], [1, 2, 1]), H=768, B=3, T=9, h0=True, mats=3),
    "spk_lstm_hid_only": dict(cell="lstm", inp=([5, 8], [48, 64], [1, 2, 1]), hid=([24, 32], [48, 64], [1, 2, 1]), H=768, B=4, T=6, h0=False, mats=2),
    "spk_gru": dict(cell="gru", inp=([5, 8], [48, 48], [1, 2, 1]), hid=([24, 32], [48, 48], [1, 2, 1]), H=768, B=4, T=5, h0=True, mats=2),
    "spk_r4": dict(cell="lstm", inp=([5, 8], [48, 64], [1, 4, 1]), hid=([24, 32], [48, 64], [1, 4, 1]), H=768, B=3, T=8, h0=False, mats=3),
    # d = 4 at H = 768 (the reference's result tables: d in {2, 4}): merged two by two, pulled back onto the four cores; in = 40 as
    # (2, 2, 2, 5): J_t = 10 is not a multiple of 4 — the input rows are staged element by element
    "spk_d4r4": dict(cell="lstm", inp=([2, 2, 2, 5], [6, 8, 8, 8], [1, 4, 4, 4, 1]), hid=([4, 4, 6, 8], [6, 8, 8, 8], [1, 4, 4, 4, 1]),
                     H=768, B=3, T=6, h0=True, mats=3),
    "spk_d4r4_hid_only": dict(cell="lstm", inp=([2, 2, 2, 5], [6, 8, 8, 8], [1, 4, 4, 4, 1]), hid=([4, 4, 6, 8], [6, 8, 8, 8], [1, 4, 4, 4, 1]),
                              H=768, B=5, T=4, h0=False, mats=2),
    "spk_d4r2": dict(cell="lstm", inp=([2, 2, 2, 5], [6, 8, 8, 8], [1, 2, 2, 2, 1]), hid=([4, 4, 6, 8], [6, 8, 8, 8], [1, 2, 2, 2, 1]),
                     H=768, B=4, T=7, h0=False, mats=3),
    "spk_d4r2_hid_only": dict(cell="lstm", inp=([2, 2, 2, 5], [6, 8, 8, 8], [1, 2, 2, 2, 1]), hid=([4, 4, 6, 8], [6, 8, 8, 8], [1, 2, 2, 2, 1]),
                              H=768, B=3, T=5, h0=True, mats=2),
    "spk_d4r3": dict(cell="lstm", inp=([2, 2, 2, 5], [6, 8, 8, 8], [1, 3, 3, 3, 1]), hid=([4, 4, 6, 8], [6, 8, 8, 8], [1, 3, 3, 3, 1]),
                     H=768, B=3, T=5, h0=True, mats=3),       # (no compile-time plan: the large variant with its plan at run time)
    # the four-core TT-GRU's hidden matrix ((6, 6, 8, 8) contracted pairwise: I_h = 36)
    "spk_gru_d4r2": dict(cell="gru", inp=([2, 2, 2, 5], [6, 6, 8, 8], [1, 2, 2, 2, 1]), hid=([4, 4, 6, 8], [6, 6, 8, 8], [1, 2, 2, 2, 1]),
                         H=768, B=3, T=7, h0=True, mats=2),
    "spk_gru_d4r4": dict(cell="gru", inp=([2, 2, 2, 5], [6, 6, 8, 8], [1, 4, 4, 4, 1]), hid=([4, 4, 6, 8], [6, 6, 8, 8], [1, 4, 4, 4, 1]),
                         H=768, B=4, T=5, h0=False, mats=2),
    # edge cases of the row walk: one step per sample (every row is a head row), a single row
    "spk_T1": dict(cell="lstm", inp=([5, 8], [48, 64], [1, 2, 1]), hid=([24, 32], [48, 64], [1, 2, 1]), H=768, B=7, T=1, h0=True, mats=3),
    "spk_one_row": dict(cell="lstm", inp=([5, 8], [48, 64], [1, 2, 1]), hid=([24, 32], [48, 64], [1, 2, 1]), H=768, B=1, T=1, h0=True, mats=2),
    # three cores, low rank: sides of one and two cores
    "h256_d3r2": dict(cell="lstm", inp=([2, 4, 5], [8, 8, 16], [1, 2, 2, 1]), hid=([4, 8, 8], [8, 8, 16], [1, 2, 2, 1]), H=256, B=4, T=10, h0=False, mats=2),
    "h256_d3r2_in64": dict(cell="lstm", inp=([2, 4, 8], [8, 8, 16], [1, 2, 2, 1]), hid=([4, 8, 8], [8, 8, 16], [1, 2, 2, 1]), H=256, B=4, T=11, h0=True, mats=3),
}


# cases only the kernel's LARGE variant with a RUN-TIME plan takes: not routed to by default (spills; slower than the dense gradient)
BIG = {"spk_d4r3"}


def _run(case, scale_dy=1.0, seed=3, poison=False, dev2=None):
    from ttrnn_hip import _lib
    if dev2 is None:
        dev2 = 8 if case in BIG else 0
    with _lib.option("dev2", dev2):
        return _run_inner(case, scale_dy, seed, poison)


def _run_inner(case, scale_dy, seed, poison):
    from ttrnn_hip import _lib, functional as F
    lib = _lib.load()
    c = CASES[case]
    gen = torch.Generator().manual_seed(seed)
    H, B, T = c["H"], c["B"], c["T"]
    G = 4 if c["cell"] == "lstm" else 3
    cores_in = make_cores(*c["inp"], gen)
    cores_hid = make_cores(*c["hid"], gen)
    n_in = int(np.prod(c["inp"][0]))
    x = torch.rand(B, T, n_in, generator=gen) * 2 - 0.5
    out = torch.tanh(torch.randn(B, T, H, generator=gen))
    h0 = torch.randn(B, H, generator=gen) * 1.5 if c["h0"] else None
    # gate gradients spanning many decades over the steps (vanishing gradients: early steps are tiny)
    decay = torch.exp(-1.5 * torch.arange(T - 1, -1, -1, dtype=torch.float32)).view(1, T, 1)
    dg_hid = torch.randn(B, T, G * H, generator=gen) * decay * scale_dy
    dg_in = dg_hid if c["cell"] == "lstm" else dg_hid * (1.0 + 0.5 * torch.rand(B, T, G * H, generator=gen))
    # float64 reference
    ref = {}
    hprev = torch.cat([(h0 if h0 is not None else torch.zeros(B, H)).unsqueeze(1), out[:, :-1]], dim=1)
    for name, cores, rows, dy in (("in", cores_in, x, dg_in), ("hid", cores_hid, hprev, dg_hid)):
        c64 = [t.double().clone().requires_grad_(True) for t in cores]
        W = dense_from_cores(c64)
        ((rows.reshape(B * T, -1).double() @ W) * dy.reshape(B * T, -1).double()).sum().backward()
        ref[name] = [t.grad for t in c64]
        ref[name + "_b"] = dy.reshape(B * T, -1).double().sum(0)
    # device call
    d = dev()
    spec_in = F.TTSpec(*c["inp"])
    spec_hid = F.TTSpec(*c["hid"])
    spec = F.RnnLayerSpec(c["cell"], n_in, H, spec_in, spec_hid, True, True)
    desc = spec.desc(B, T, _lib.TTRNN_F32)
    mats = c["mats"]
    wsb = lib.ttrnn_rnn_wgrad_workspace(ctypes.byref(desc), mats)
    assert wsb > 0, "the chain kernel does not take this case"
    ci = [t.to(d) for t in cores_in]
    ch = [t.to(d) for t in cores_hid]
    assert all(not t.is_contiguous() or min(t.shape[1:3]) == 1 for t in ci + ch) or True
    pk_in, pk_hid = spec_in.pack(ci), spec_hid.pack(ch)
    xd, od = x.to(d), out.to(d)
    h0d = h0.to(d) if h0 is not None else None
    dgi = dg_in.to(d)
    dgh = dgi if c["cell"] == "lstm" else dg_hid.to(d)
    if c["cell"] == "gru":
        dgi = dg_in.to(d)
    F.POISON_ALLOCATIONS = poison
    try:
        ws = F._workspace(wsb, d)
        dpi = torch.zeros(spec_in.packed_elems, device=d)
        dph = torch.zeros(spec_hid.packed_elems, device=d)
        dbi = torch.zeros(G * H, device=d)
        dbh = torch.zeros(G * H, device=d)
        hb = torch.ones(H, device=d) if h0 is None else torch.maximum(h0d.abs().amax(0), torch.ones(H, device=d))
        wa = _lib.WgradArgs(F._ptr(xd), F._ptr(od), F._ptr(h0d), F._ptr(dgi), F._ptr(dgh), F._ptr(pk_in), F._ptr(pk_hid),
                            F._ptr(dpi) if mats & 1 else None, F._ptr(dph), F._ptr(dbi) if mats & 1 else None, F._ptr(dbh),
                            None, F._ptr(hb), None, None)
        _lib.check(lib.ttrnn_rnn_wgrad(ctypes.byref(desc), mats, ctypes.byref(wa), F._ptr(ws), wsb, F._stream(xd)), "ttrnn_rnn_wgrad")
        torch.cuda.synchronize()
        got = {"hid": [g.cpu() for g in spec_hid.unpack_grads(dph, ch)], "hid_b": dbh.cpu()}
        if mats & 1:
            got["in"] = [g.cpu() for g in spec_in.unpack_grads(dpi, ci)]
            got["in_b"] = dbi.cpu()
    finally:
        F.POISON_ALLOCATIONS = False
    return ref, got


@pytest.mark.parametrize("case", sorted(CASES))
def test_chain_wgrad_vs_float64(case):
    ref, got = _run(case)
    worst = 0.0
    for name in ("in", "hid"):
        if name not in got:
            continue
        for k, (r, g) in enumerate(zip(ref[name], got[name])):
            scale = float(r.abs().max())
            err = float((g.double() - r).abs().max()) / scale
            worst = max(worst, err)
            assert err <= 3e-6, (case, name, k, err)       # fp32-class: two fp16 pieces per operand, fp32 accumulation
        rb, gb = ref[name + "_b"], got[name + "_b"]
        assert float((gb.double() - rb).abs().max()) <= 3e-6 * float(rb.abs().max()), (case, name, "bias")
    print(case, "worst relative error %.2e" % worst)


@pytest.mark.parametrize("scale", [1e-12, 1e-4, 1e6, 1e15])
def test_chain_wgrad_gate_gradient_ranges(scale):
    """the launch's power-of-two scales follow the operands: the same relative error over 27 decades of gate gradients"""
    ref, got = _run("spk_lstm_h0", scale_dy=scale)
    for name in ("in", "hid"):
        for r, g in zip(ref[name], got[name]):
            assert float((g.double() - r).abs().max()) <= 3e-6 * float(r.abs().max()), (scale, name)


@pytest.mark.parametrize("case", ["spk_lstm", "spk_lstm_h0", "spk_lstm_hid_only", "spk_gru", "spk_r4", "spk_d4r4", "spk_d4r2", "spk_d4r2_hid_only"])
def test_image_kernel_still_takes_the_encoder_shapes(case):
    """option dev2 bit 6 turns the register hand-off kernel (ttrnn_c2r_dev.h) off: k_c2w, the kernel with the C1 / dC1 images in
    LDS, takes the same calls (the A/B of DESIGN.md 4c) to the same accuracy"""
    dev2 = 64 | (8 if case in BIG else 0)
    ref, got = _run(case, dev2=dev2)
    for name in ("in", "hid"):
        if name not in got:
            continue
        for r, g in zip(ref[name], got[name]):
            assert float((g.double() - r).abs().max()) <= 3e-6 * float(r.abs().max()), (case, name)


def test_run_time_plan_kernel_is_bit_identical_to_the_compile_time_instantiation():
    """the speaker encoder's shapes run a kernel whose plan is a compile-time constant (c2_const_plan); option dev2 bit 2 selects
    the same kernel with the plan at run time: same arithmetic, same order, same bits"""
    for case in ("spk_lstm_h0", "spk_lstm_hid_only", "spk_d4r2", "spk_d4r4_hid_only"):
        _, a = _run(case, seed=11, dev2=64)
        _, b = _run(case, seed=11, dev2=64 | 4)
        for name in ("in", "hid"):
            if name in a:
                for x, y in zip(a[name], b[name]):
                    assert torch.equal(x, y), (case, name)


def test_chain_wgrad_is_bitwise_repeatable():
    """fixed-order sums: two launches on poisoned workspaces give the same bits"""
    _, a = _run("spk_lstm", seed=5, poison=True)
    _, b = _run("spk_lstm", seed=5, poison=True)
    for name in ("in", "hid"):
        for x, y in zip(a[name], b[name]):
            assert torch.equal(x, y)
        assert torch.equal(a[name + "_b"], b[name + "_b"])


def test_chain_route_is_offered_only_where_the_chain_is_cheaper():
    """2 in out > 1.5 x chain FLOPs decides: the speaker encoder's shapes take the chain, cfg2 / cfg4 stay dense"""
    from ttrnn_hip import _lib, functional as F
    lib = _lib.load()

    def offered(cell, inp, hid, H, mats=2):
        n_in = int(np.prod(inp[0]))
        spec = F.RnnLayerSpec(cell, n_in, H, F.TTSpec(*inp), F.TTSpec(*hid), True, True)
        d = spec.desc(64, 32, _lib.TTRNN_F32)
        return lib.ttrnn_rnn_wgrad_workspace(ctypes.byref(d), mats) > 0

    spk = CASES["spk_lstm"]
    assert offered("lstm", spk["inp"], spk["hid"], 768, 3) and offered("lstm", spk["inp"], spk["hid"], 768, 2)
    assert not offered("lstm", ([1, 1, 1], [8, 8, 16], [1, 8, 8, 1]), ([4, 8, 8], [8, 8, 16], [1, 8, 8, 1]), 256)         # cfg2
    assert not offered("lstm", ([2, 4, 5], [8, 8, 16], [1, 16, 16, 1]), ([4, 8, 8], [8, 8, 16], [1, 16, 16, 1]), 256)     # cfg4
    with _lib.option("dev2", 1):
        assert not offered("lstm", spk["inp"], spk["hid"], 768, 3)
    # rank 4 at H = 768 and the four-core shapes need the kernel's large variant: offered where the plan has a compile-time
    # instantiation (ranks 2 and 4); other ranks (run-time plan of the large variant: slower than the dense gradient) are not
    r4 = CASES["spk_r4"]
    assert offered("lstm", r4["inp"], r4["hid"], 768, 3) and offered("lstm", r4["inp"], r4["hid"], 768, 2)
    for name in ("spk_d4r4", "spk_d4r2"):
        d4 = CASES[name]
        assert offered("lstm", d4["inp"], d4["hid"], 768, 2) and offered("lstm", d4["inp"], d4["hid"], 768, 3), name
    d3 = CASES["spk_d4r3"]
    assert not offered("lstm", d3["inp"], d3["hid"], 768, 2)
