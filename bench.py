#!/usr/bin/env python3
"""Benchmark of the TT-LSTM / TT-GRU hot path on MI355X (contract: see the task brief).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward pass (`no_grad`) of the workload's module over one synthetic batch
[B, T, in] that is already resident in HBM.  Workload at every N: BASELINE.json configs[1]
("cfg2": TT-LSTM in=1 H=256 ncores=3 ttrank=8, seq_len=784, batch=64 PER GPU, fp32) — batch-sharded,
no data-path collective in the forward (weak scaling).  `value` = whole-job timesteps/s =
N * T / t_step, t_step = max over ranks of the mean step time.

Extra objects on the JSON line:
  roofline      compute roofline of the dominant kernel (ttrnn_rnn_forward): algorithmic FLOP per
                launch (691 712 FLOP per sample-timestep for cfg2) / the kernel's
                mean duration measured with events on its launch stream; peak = 157.3 TFLOP/s fp32.
  cpu_baseline  the oracle's op-for-op torch-CPU restatement of the reference path ("port") timed
                on this host's cores on the same workload (rank 0, N=1 only).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tensorized-rnn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

FP32_MATH_DESC = {
    None: None,
    "split": "split (default where a split kernel exists): fp32 operands as error-compensated 16-bit pieces with fp32 "
             "accumulate - two fp16 pieces / 3 MFMA terms under per-launch power-of-two scales in the LSTM forward "
             "kernels (cfg2, cfg4, cfg5), three bf16 pieces / 6 terms elsewhere; error vs float64 equal to the fp32-MFMA "
             "mode - 1.2e-7 abs over 784 steps in BOTH modes, set by the 1-ulp hardware exp / rcp of the gate non-linearities, "
             "not by the matrix arithmetic (the reference's CPU fp32: 1.6e-8) "
             "(tests/test_gpu_parity.py::test_split_math_error_vs_fp64_is_fp32_class, ::test_split_math_operand_ranges, "
             "::test_split_math_outlier_up, ::test_big_shape_half_piece_operand_ranges); operands whose magnitude spread would "
             "cost the fp16 pieces bits are detected on the device and run on three bf16 pieces (DESIGN.md 4a); "
             "TTRNN_FP32_MATH=exact selects the fp32 MFMA",
    "exact": "exact: v_mfma_f32_16x16x4_f32 on fp32 operands",
}

WORKLOADS = {
    # name: kind, in, H, layers, ncores, rank, B (per GPU), T, dtype, FLOP per sample-timestep
    "cfg2": dict(kind="ttlstm", inp=1, H=256, L=1, d=3, r=8, B=64, T=784, dtype="f32", flop=691712, flop_in=33024,
                 desc="TT-LSTM in=1 H=256 ncores=3 ttrank=8 seq_len=784 batch=64/GPU fp32 (BASELINE.json configs[1])"),
    "cfg3": dict(kind="ttgru", inp=1, H=256, L=1, d=3, r=8, B=256, T=784, dtype="bf16", flop=519360, flop_in=24768,
                 desc="TT-GRU in=1 H=256 ncores=3 ttrank=8 seq_len=784 batch=256/GPU bf16 storage (configs[2])"),
    "cfg3_fp32": dict(kind="ttgru", inp=1, H=256, L=1, d=3, r=8, B=256, T=784, dtype="f32", flop=519360, flop_in=24768,
                      desc="TT-GRU in=1 H=256 ncores=3 ttrank=8 seq_len=784 batch=256/GPU fp32 (configs[2] in the reference's own dtype)"),
    "gru64": dict(kind="ttgru", inp=1, H=256, L=1, d=3, r=8, B=64, T=784, dtype="f32", flop=519360, flop_in=24768,
                  desc="TT-GRU in=1 H=256 ncores=3 ttrank=8 seq_len=784 batch=64/GPU fp32 (cfg2's sizes with the GRU cell)"),
    "cfg4": dict(kind="ttlstm", inp=40, H=256, L=3, d=3, r=16, B=512, T=160, dtype="f32", flop=12416768,
                 desc="3-layer TT-LSTM in=40 H=256 ncores=3 ttrank=16 seq_len=160 batch=512/GPU fp32 (configs[3] per-GPU batch)"),
    "cfg5": dict(kind="ttlstm", inp=1024, H=1024, L=1, d=4, r=32, B=128, T=1024, dtype="f32", flop=70267904,
                 desc="TT-LSTM in=1024 (assumed) H=1024 ncores=4 ttrank=32 seq_len=1024 batch=128/GPU fp32 (configs[4] global batch on one GPU)"),
    "cfg1": dict(kind="ttlstm", inp=1, H=128, L=1, d=2, r=4, B=32, T=784, dtype="f32", flop=71552, flop_in=4352,
                 desc="TT-LSTM in=1 H=128 ncores=2 ttrank=4 seq_len=784 batch=32 fp32 (configs[0] shapes)"),
    # the reference harness' defaults (benchmarking.py: in=256 H=512 batch 512, 160 steps) with --naive_tt / --ttrank 16: the runtime tier's
    # paired forward kernel (k_g2_fwd_p).  FLOP per sample-timestep = sum over both TT-matrices (per gate for the naive set) of
    # 2 rows_k K_k M_k over the chain's stages + 13 per hidden unit
    "naive512": dict(kind="ttlstm", inp=256, H=512, L=1, d=3, r=8, B=512, T=160, dtype="f32", naive=True, flop=3938816,
                     desc="naive per-gate TT-LSTM (tt_linearset.py) in=256 H=512 ncores=3 ttrank=8 seq_len=160 batch=512/GPU fp32 (benchmarking.py --naive_tt)"),
    "r16_512": dict(kind="ttlstm", inp=256, H=512, L=1, d=3, r=16, B=512, T=160, dtype="f32", flop=13769216,
                    desc="TT-LSTM in=256 H=512 ncores=3 ttrank=16 seq_len=160 batch=512/GPU fp32 (benchmarking.py --ttrank 16)"),
    # the reference's own published encoder layer (experiments/speaker_verification/encoder/params_model.py:2-4,14-16: 40 mel channels,
    # hidden 768, ONE layer, n_cores 2, rank 2; 160-frame partial utterances, speaker_encoder.py:42-48) at the cfg4 batch.  FLOP per
    # sample-timestep (SURVEY 8(d) rule): hidden 2*24*128*32 + 2*64*48*48 = 491 520, input 2*5*128*8 + 2*64*48*10 = 71 680, + 13 H
    "spk": dict(kind="ttlstm", inp=40, H=768, L=1, d=2, r=2, B=512, T=160, dtype="f32", flop=573184, flop_in=71680,
                desc="TT-LSTM in=40 H=768 ncores=2 ttrank=2 seq_len=160 batch=512/GPU fp32 (the reference's speaker-encoder layer, params_model.py)"),
}
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 vector = f32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0
GPU_CLOCK_HZ = 2.4e9          # MI355X_MICROARCH.md: peak engine clock; MFMA busy cycles are priced against it

# What the default kernels EXECUTE per sample-timestep on the matrix pipes (MFMA instructions issued, padding and the
# split terms included; DESIGN.md section 4 derives each count):
#   bf16_mfma = 16-bit MFMAs, v_mfma_f32_16x16x32_{bf16,f16} (16 384 FLOP, 16 pipe cycles on one SIMD; `pipe16` names the
#   operand type), fp32 = v_mfma_f32_16x16x4_f32 (2 048 FLOP,
#   32 pipe cycles).  `rec_simds` = SIMDs the recurrent kernel's MFMAs of ONE sample are spread over (a workgroup owns a CU).
EXECUTED = {
    # cfg2: S2 16 tiles x 1 term-packed fp16 MFMA + S10 4 tiles x 8 k-blocks x 3 terms (k_lstm_fwd_f10)
    "cfg2": dict(bf16_mfma=16 + 96, fp32_mfma=0, kin_bf16_flop=0, rec_simds=4, pipe16="f16_mfma", terms=3,
                 note="fused core on two-piece fp16 operands: S2 (K=8, four terms packed into one MFMA per tile) + S10 "
                      "(64 x 16 x 256, three terms x0w0 + x0w1 + x1w0)"),
    # cfg1 (round 4, ttrnn_fast_f2.hip): stage 1 8 m-tiles x 2 chained MFMAs (the three split terms, K = 16 packed twice along the
    # 32-wide k) + stage 0 2 column tiles x 3 terms, on the two waves (two SIMDs) of a sample's workgroup
    "cfg1": dict(bf16_mfma=16 + 6, fp32_mfma=0, kin_bf16_flop=0, rec_simds=2, pipe16="f16_mfma", terms=3,
                 note="two-core kernel on two-piece fp16 operands (k_lstm_fwd_f2): one barrier per step, gates on the accumulators"),
    # cfg3: bf16 storage, plain bf16 MFMAs: S2 4 waves x 6 m-tiles (each wave its own eight chain rows: half of a tile's
    # columns are padding) + S10 4 tiles x 8 k-blocks (k_gru_fwd_f10v)
    "cfg3": dict(bf16_mfma=24 + 32, fp32_mfma=0, kin_bf16_flop=0, rec_simds=4,
                 note="bf16 fused core, no splitting (storage precision is bf16); S2 inside the gate waves"),
    # fp32 GRU (round 5, k_gru_fwd_f10vh): S2 on tile pairs, 4 waves x 6 MFMAs, + S10 4 tiles x 8 k-blocks x 3 terms
    "cfg3_fp32": dict(bf16_mfma=24 + 96, fp32_mfma=0, kin_bf16_flop=0, rec_simds=4, pipe16="f16_mfma", terms=3,
                      note="fused core on two-piece fp16 operands, S2 (tile pairs, four terms in two MFMAs) inside the gate waves"),
    "gru64": dict(bf16_mfma=24 + 96, fp32_mfma=0, kin_bf16_flop=0, rec_simds=4, pipe16="f16_mfma", terms=3,
                  note="fused core on two-piece fp16 operands, S2 (tile pairs, four terms in two MFMAs) inside the gate waves"),
    # cfg4, per layer: S2 32 tiles x 1 + S10 4 tiles x 16 k-blocks x 3 terms (fp16 pieces); K-in: dense GEMM on fp16 pieces,
    # 3 terms, contraction padded to 64 (layer 0, in = 40) / 256 (layers 1, 2)
    "cfg4": dict(bf16_mfma=3 * (32 + 192), fp32_mfma=0, kin_bf16_flop=3 * 2 * 1024 * (64 + 256 + 256), rec_simds=4, rec_wgs_per_cu=2,
                 pipe16="f16_mfma", terms=3,
                 note="per layer: fused core (r = 16) on two-piece fp16 operands (four-wave workgroups, two per CU) + K-in as one "
                      "dense GEMM over B*T rows on two-piece fp16 operands (three terms)"),
    # cfg5: merged two-core matrix on two-piece fp16 operands (k_lstm_fwd_big2h): stage 1 128 m-tiles x 2 k-blocks x 3 terms,
    # stage 0 4 feature tiles x 4 row tiles x 16 k-blocks x 3 terms, over the 8 SIMDs of a workgroup pair; K-in: dense GEMM on
    # fp16 pieces
    "cfg5": dict(bf16_mfma=128 * 2 * 3 + 4 * 4 * 16 * 3, fp32_mfma=0,
                 kin_bf16_flop=3 * 2 * 1024 * 4096, rec_simds=8, rec_wgs_per_sample=2, pipe16="f16_mfma", terms=3,
                 note="K-rec: merged 2-core chain on two-piece fp16 operands (three terms per product), two workgroups per "
                      "sample, every core fragment resident (registers + a quarter of stage 1's in LDS); K-in: dense GEMM on two-piece fp16 operands (three terms)"),
    # runtime tier, two samples per workgroup (k_g2_fwd_p), per SAMPLE-timestep: stage 1 = 64 term-packed tiles (16 m tiles x 4 column
    # tiles); stage 2 = the streamed head, 256 blocks x 3 terms per PAIR (naive: one column tile holds both samples) / x 6 (r = 16:
    # two column tiles); K-in: dense GEMM on fp16 pieces
    "naive512": dict(bf16_mfma=64 + 256 * 3 // 2, fp32_mfma=0, kin_bf16_flop=3 * 2 * 2048 * 256, rec_simds=4, rec_wgs_per_sample=0.5,
                     pipe16="f16_mfma", terms=3,
                     note="runtime tier, two samples per eight-wave workgroup: the block-diagonal head (512 KB of fp16 pieces) streamed from L2 "
                          "once per step for both; bound by the L2 -> CU path (55 of 64 B/clk in stage 2), not by the matrix pipe"),
    # k_lstm_fwd_w2<2> (round 6): per wave stage 1 (2 ranks x 2 row blocks + 2 input tiles) x 3 terms = 18, stage 2 3 tiles x 2 k-blocks
    # x 3 terms = 18; four waves per sample, two workgroups per CU; no K-in at all (the input chain rides in the hidden chain's padding)
    "spk": dict(bf16_mfma=4 * 36, fp32_mfma=0, kin_bf16_flop=0, rec_simds=4, rec_wgs_per_cu=2, pipe16="f16_mfma", terms=3,
                note="two-core kernel with wave-local stages on two-piece fp16 operands (k_lstm_fwd_w2): register-local hand-off, the "
                     "input projection inside the recurrent kernel, one barrier per step"),
    "r16_512": dict(bf16_mfma=64 + 256 * 6 // 2, fp32_mfma=0, kin_bf16_flop=3 * 2 * 2048 * 256, rec_simds=4, rec_wgs_per_sample=0.5,
                    pipe16="f16_mfma", terms=3,
                    note="runtime tier, two samples per eight-wave workgroup in two column tiles: the head (512 KB of fp16 pieces) streamed "
                         "from L2 once per step for both; bound by the L2 -> CU path"),
}


class EventTimer(object):
    """Records (start, end) event pairs on the current stream around named kernel launches."""

    def __init__(self, sample=1):
        self.pairs = {}
        self._open = {}
        self._seen = {}
        self.enabled = False
        # sample = k: only every k-th start / stop pair of a name records events.  An event record is a barrier packet in the
        # queue on this stack (~4.5 us of GPU timeline each, tools/gap_report.sh): with a pair around EVERY call the harness is
        # ~2 % of a 0.5 ms step; sampled, the launch durations are still measured live inside the timed region
        self.sample = max(1, int(sample))

    def start(self, name):
        if self.enabled:
            n = self._seen.get(name, 0)
            self._seen[name] = n + 1
            if n % self.sample:
                return
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._open[name] = ev

    def stop(self, name):
        if self.enabled and name in self._open:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.pairs.setdefault(name, []).append((self._open.pop(name), ev))

    def calls(self, name):
        """start() calls seen while enabled (sampled or not)."""
        return self._seen.get(name, 0)

    def times_ms(self, name):
        return [a.elapsed_time(b) for a, b in self.pairs.get(name, [])]

    def mean_ms(self, name):
        p = self.times_ms(name)
        return sum(p) / len(p) if p else None

    def median_ms(self, name):
        p = sorted(self.times_ms(name))
        return p[len(p) // 2] if p else None

    def count(self, name):
        return len(self.pairs.get(name, []))


def build_model(w, device):
    from tensorized_rnn.gru import TTGRU
    from tensorized_rnn.tt_lstm import TTLSTM
    torch.manual_seed(1111)
    cls = TTLSTM if w["kind"] == "ttlstm" else TTGRU
    with contextlib.redirect_stdout(io.StringIO()):
        m = cls(w["inp"], w["H"], w["L"], device, n_cores=w["d"], tt_rank=w["r"], is_naive=bool(w.get("naive")))
    if w["dtype"] == "bf16":
        m = m.to(torch.bfloat16)
    return m.eval()


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(w, state_dict, x, budget_s=20.0):
    """Times the oracle (op-for-op torch-CPU restatement of the reference path; tools/validate_oracle_speed.py checks in
    the build container that it runs within 10 % of the reference itself) on the host: torch.set_num_threads(1) and
    (all physical cores), median of 5 runs each on a bounded T-slice of the same workload — the SAME
    weights (the GPU module's state_dict, as fp32) and the SAME input tensor the GPU steps were timed on."""
    from oracle import ttrnn_oracle as O
    logical = os.cpu_count() or 1
    phys = min(physical_cores(), logical)
    layers, _ = O.layers_from_state_dict({k: v.detach().float().cpu() for k, v in state_dict.items()}, w["L"])
    fwd = O.lstm_forward if w["kind"] == "ttlstm" else O.gru_forward
    x = x.detach().float().cpu()

    def run(T):
        t0 = time.perf_counter()
        with torch.no_grad():
            fwd(layers, x[:, :T])
        return time.perf_counter() - t0

    res = {}
    sample_T = {}
    mid = min(8, phys)                      # BASELINE.md section 3 quotes the reference at 8 threads
    counts = sorted({1, mid, phys})
    for n in counts:
        torch.set_num_threads(n)
        run(2)
        per_step = min(run(4), run(4)) / 4.0
        # five runs inside this thread count's share of the budget, at least 4 and at most all T steps
        T = int(max(4, min(w["T"], budget_s / len(counts) / 5.0 / per_step)))
        times = sorted(run(T) for _ in range(5))
        res[n] = T / times[2]
        sample_T[n] = T
    best = max(res, key=res.get)
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    return {"value": res[best], "unit": "timesteps/s", "cores": best, "kind": "port",
            "threads_1": res[1], "threads_8": res[mid] if mid == 8 else None, "threads_all_physical": res[phys],
            "by_threads": {str(n): res[n] for n in counts}, "physical_cores": phys, "logical_cpus": logical,
            "cpu_model": model,
            "sample": "oracle/ttrnn_oracle.py (torch-CPU, op-for-op restatement of the reference loop; within 10 % of the "
                      "reference's own speed: profiles/r2/oracle_speed_validation.json), batch {}, first {} of {} timesteps at {} "
                      "thread(s), fp32, no_grad, median of 5 runs each, on the weights and the input tensor of the GPU run; "
                      "`value` = the fastest thread count (all physical "
                      "cores oversubscribe a path of ~40 tiny ATen ops per step)".format(
                          w["B"], "/".join(str(sample_T[n]) for n in counts), w["T"], "/".join(str(n) for n in counts))}


def run_grid(args, device):
    """H x ncores x ttrank x {LSTM, GRU} (in = 40, B = 64, T = 64): which kernel family each shape runs on and its forward
    time against the any-shape VALU kernels (force_generic) — VERDICT r1 item 4: no experiment flag combination of the
    reference (pmnist_test.py:47-56, params_model.py) should land on ttrnn_generic.hip."""
    import ttrnn_hip
    from tensorized_rnn.gru import TTGRU
    from tensorized_rnn.tt_lstm import TTLSTM
    from ttrnn_hip import functional as F
    B, T, inp = 64, 64, 40
    rows = []

    train = args.mode == "train"      # forward + BPTT (loss = sum of the outputs) instead of the no_grad forward

    def one(m, x):
        if not train:
            with torch.no_grad():
                m(x)
            return
        m.zero_grad(set_to_none=True)
        m(x)[0].sum().backward()

    def timed_graph(m, x, n):
        """forward + BPTT of sum(outputs), captured once (ttrnn_hip.CapturedTrainStep, no optimizer) and replayed"""
        cap = ttrnn_hip.CapturedTrainStep(m, None, lambda mm, xx: mm(xx)[0].sum(), (x,), warmup=2)
        cap(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            cap(x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    def timed(m, x, n):
        one(m, x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            one(m, x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    for cell, cls in (("lstm", TTLSTM), ("gru", TTGRU)):
        for H in (64, 128, 256, 512, 768, 1024):
            for d in (2, 3, 4):
                for r in (2, 4, 8, 16):
                    torch.manual_seed(1111)
                    with contextlib.redirect_stdout(io.StringIO()):
                        m = cls(inp, H, 1, device, n_cores=d, tt_rank=r).eval()
                    x = torch.rand(B, T, inp, device=device)
                    route = F.rnn_route(m._all_layers[0]._layer_spec(), B, T)
                    bwd_route = F.rnn_backward_route(m._all_layers[0]._layer_spec(), B, T)
                    ms = timed(m, x, max(3, args.steps // 4))
                    with ttrnn_hip.option("force_generic", 1):
                        valu_ms = timed(m, x, 3)
                    graph_ms = round(timed_graph(m, x, max(3, args.steps // 4)), 4) if (train and args.graph) else None
                    rows.append({"cell": cell, "H": H, "ncores": d, "ttrank": r, "route": route, "bwd_route": bwd_route, "ms": round(ms, 4),
                                 "graph_ms": graph_ms,
                                 "valu_ms": round(valu_ms, 4), "speedup": round(valu_ms / ms, 2)})
    on_valu = [r for r in rows if r["route"] == "valu"]
    geo = 1.0
    for r in rows:
        geo *= r["speedup"] ** (1.0 / len(rows))
    line = {"metric": ("training step (forward + BPTT)" if train else "forward") +
                      " time per shape, MFMA route vs any-shape VALU kernels (geometric-mean speed-up)", "value": geo,
            "unit": "x", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "grid: TT-LSTM / TT-GRU in=40, H in {64..1024}, ncores in {2,3,4}, ttrank in {2,4,8,16}, "
                                   "batch 64, seq_len 64, " + ("forward + backward of sum(outputs)" if train else "forward (no_grad)")},
            "shapes": len(rows), "shapes_on_valu_route": len(on_valu),
            "eager_ms": {"min": min(r["ms"] for r in rows), "median": sorted(r["ms"] for r in rows)[len(rows) // 2]},
            "graph_replay_ms": (None if not (train and args.graph) else
                                {"min": min(r["graph_ms"] for r in rows),
                                 "median": sorted(r["graph_ms"] for r in rows)[len(rows) // 2],
                                 "what": "the same forward + BPTT recorded once into a hipGraph and replayed "
                                         "(ttrnn_hip.CapturedTrainStep): host floor of the eager step removed"}),
            "bwd_routes": {k: sum(1 for r in rows if r["bwd_route"] == k) for k in sorted({r["bwd_route"] for r in rows})},
            "routes": {k: sum(1 for r in rows if r["route"] == k) for k in sorted({r["route"] for r in rows})},
            "grid": rows}
    print(json.dumps(line))


def _free_port():
    import socket
    with contextlib.closing(socket.socket(socket.AF_INET, socket.SOCK_STREAM)) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a torchrun environment: start `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same args>` as a child process, pass its output
    through, print rank 0's JSON line last and return the child's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
    payload = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            payload = ln.rstrip("\n")          # held back: it must be the LAST line
        else:
            sys.stdout.write(ln)
    rc = proc.wait()
    sys.stdout.flush()
    if payload is not None:
        print(payload, flush=True)
    elif rc == 0:
        rc = 1
        sys.stderr.write("bench.py: the ranks exited 0 without a JSON line\n")
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS) + ["grid"],
                    help="cfgN: one BASELINE.json configuration (cfg2 = the headline metric).  grid: sweep over the "
                         "(hidden_size, ncores, ttrank, cell) combinations of the reference's experiment flags, route and "
                         "time per shape against the any-shape VALU kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prepared", action="store_true",
                    help="forward mode: time the steps on prepare_for_inference() modules (weight-only work kept across the "
                         "no_grad forwards).  Default: every timed forward starts from the TT cores, as the reference's eval "
                         "loop does (benchmarking.py:16-38); the prepared figure is then reported beside it as `prepared`")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the configuration's batch PER GPU (the headline metric, 'timesteps/sec/GPU (batch=64)'). "
                         "strong: the configuration's batch is the GLOBAL batch, sharded over the ranks (SURVEY.md 8(d): "
                         "cfg4 512 -> 64 per GPU, cfg5 128 -> 16 per GPU at 8 GPUs)")
    ap.add_argument("--shard-of", type=int, default=0, metavar="N",
                    help="diagnostic, single GPU only: time the shard ONE rank of an N-GPU `--scaling strong` job would get "
                         "(rank 0's slice of the configuration's batch).  The line is labelled `shard_of`; it is not the "
                         "headline metric and no scaling number")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="forward: the headline metric (no_grad forward). train: the reference's training benchmark step "
                         "(benchmarking.py:41-70: classifier forward + nll_loss + BPTT + Adam) + flat-bucket gradient "
                         "all-reduce (RCCL) for N > 1, reported in the same unit")
    ap.add_argument("--seq-len", type=int, default=0, metavar="T",
                    help="diagnostic: override the configuration's sequence length (the workload description says so; not the "
                         "headline metric)")
    ap.add_argument("--graph", action="store_true",
                    help="train mode / grid --mode train: record the step once into a hipGraph (ttrnn_hip.CapturedTrainStep) and "
                         "time REPLAYS — one host call per step instead of 30 ... 110 launches through ctypes and autograd.  "
                         "Opt-in; the line says so in config.mode.  Single process only")
    args = ap.parse_args()
    if args.graph and args.mode != "train":
        raise SystemExit("--graph captures a TRAINING step: use it with --mode train")
    if args.graph and args.gpus > 1:
        raise SystemExit("--graph is single-process (the gradient all-reduce is not captured)")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  Nothing here has touched the GPU (no
        # torch.cuda call, libttrnn not loaded), and the ranks are CHILD processes — never an exec of this one.
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE={}: under torch.distributed.run pass --nproc-per-node {} (or run plain "
                         "`python bench.py --gpus {}`, which starts its own ranks)".format(args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libttrnn has no CPU path)")
    # TTRNN_BENCH_SINGLE_DEVICE=1 (with TTRNN_BENCH_BACKEND=gloo) lets a 1-GPU box exercise the N>1 code path
    if os.environ.get("TTRNN_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    backend = None
    # TTRNN_BENCH_FORCE_DIST=1: take the distributed branch (process group on RCCL, barriers, MAX all-reduce of the time,
    # the flat-bucket gradient all-reduce in train mode) with ONE rank, so that a 1-GPU box runs the code path of --gpus N
    force_dist = os.environ.get("TTRNN_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", str(20000 + os.getpid() % 20000))   # one-rank forced group: no fixed port to collide on
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("TTRNN_BENCH_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
        dist.init_process_group(backend, rank=rank, world_size=world,
                                **({"device_id": device} if backend == "nccl" else {}))

    if args.workload == "grid":
        if world != 1:
            raise SystemExit("--workload grid is a single-GPU sweep")
        return run_grid(args, device)
    w = dict(WORKLOADS[args.workload])
    if args.seq_len > 0:
        w["T"] = args.seq_len
        w["desc"] += " — DIAGNOSTIC: seq_len overridden to {}".format(args.seq_len)
    global_batch = w["B"] * world
    if args.scaling == "strong":
        from ttrnn_hip.dist import shard_bounds
        global_batch = w["B"]
        if global_batch < world:
            raise SystemExit("--scaling strong: global batch {} < {} ranks".format(global_batch, world))
        lo, hi = shard_bounds(global_batch, rank, world)
        w["B"] = hi - lo                              # this rank's contiguous shard of the global batch
    if args.shard_of > 1:
        if world != 1 or args.scaling == "strong":
            raise SystemExit("--shard-of is a single-GPU diagnostic (use --scaling strong under torch.distributed.run for a real job)")
        from ttrnn_hip.dist import shard_bounds
        lo, hi = shard_bounds(w["B"], 0, args.shard_of)
        w["B"] = hi - lo
        global_batch = w["B"]
    from ttrnn_hip import functional as F
    model = build_model(w, device)
    torch.manual_seed(1111 + rank)
    x = torch.rand(w["B"], w["T"], w["inp"], device=device)
    if w["dtype"] == "bf16":
        x = x.to(torch.bfloat16)

    timer = EventTimer(sample=1 if (args.graph or args.steps < 8) else 4)      # every 4th library call of the timed loop carries events
    F.KERNEL_TIMER = timer
    prepared = args.mode == "forward" and args.prepared
    if prepared:
        model.prepare_for_inference()       # include/ttrnn.h: ttrnn_rnn_forward_phase (opt-in; weights are fixed in this loop)

    reducer = None
    optimizer_impl = None
    if args.mode == "train":
        # the reference's own training benchmark step (experiments/digit_classification/benchmarking.py:41-70):
        # MNIST_Classifier (TT-RNN -> TTLinear head on the last timestep -> log_softmax), random integer targets,
        # zero_grad + forward + nll_loss + backward + Adam(lr 1e-3); plus, for N > 1, the flat-bucket gradient all-reduce
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        from models import MNISTClassifier
        from ttrnn_hip.dist import FlatGradAllReduce
        n_cls = 10 if w["inp"] == 1 else 256           # pMNIST classes / benchmarking.py's default emb_size
        torch.manual_seed(1111)
        with contextlib.redirect_stdout(io.StringIO()):
            model = MNISTClassifier(w["inp"], n_cls, w["H"], w["L"], device, gru=(w["kind"] == "ttgru"), n_cores=w["d"],
                                    tt_rank=w["r"]).to(device)
        if w["dtype"] == "bf16":
            model = model.to(torch.bfloat16)
        model.train()
        torch.manual_seed(2222 + rank)
        target = torch.randint(0, n_cls, (w["B"],), device=device)
        reducer = FlatGradAllReduce(model, force=force_dist) if dist is not None else None
        if reducer is not None and backend != "gloo" and not args.graph:
            reducer.timer = timer      # events around the bucket copies and the all-reduce (reported under "collectives_measured")
        if args.graph:
            import ttrnn_hip
            opt = ttrnn_hip.adam_for_capture(model.parameters(), lr=1e-3)
            optimizer_impl = "torch.optim.Adam(capturable=True, fused where available), step replayed from a hipGraph"
        else:
            try:        # one fused multi-tensor kernel instead of eight foreach launches (same update rule)
                opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
                optimizer_impl = "torch.optim.Adam(fused=True)"
            except (RuntimeError, TypeError, ValueError):
                opt = torch.optim.Adam(model.parameters(), lr=1e-3)
                optimizer_impl = "torch.optim.Adam"

    def step():
        if args.mode == "forward":
            with torch.no_grad():
                return model(x)
        opt.zero_grad()
        out = model(x)
        loss = torch.nn.functional.nll_loss(out.float(), target)
        loss.backward()
        if reducer is not None:
            reducer.sync()
        opt.step()
        return loss

    eager_kern_ms = None
    if args.graph:
        # two eager steps with the kernel timer on (the events of the recurrent launches cannot be recorded inside a replay),
        # then the capture; from here on step() is one graph replay
        import ttrnn_hip
        step()
        torch.cuda.synchronize()
        timer.enabled = True
        step()
        step()
        torch.cuda.synchronize()
        timer.enabled = False
        eager_kern_ms = (timer.mean_ms("ttrnn_rnn_forward"), timer.median_ms("ttrnn_rnn_forward"),
                         timer.count("ttrnn_rnn_forward") / 2.0)
        timer.pairs.clear()
        captured = ttrnn_hip.CapturedTrainStep(
            model, opt, lambda m, xx, tt: torch.nn.functional.nll_loss(m(xx).float(), tt), (x, target), warmup=1)

        def step():                                   # noqa: F811
            return captured(x, target)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    last = None
    # Events inside the timed region: ONE pair per library call (ttrnn_hip.functional's KERNEL_TIMER hooks around
    # ttrnn_rnn_forward_cores / _backward) — the roofline's launch durations.  Until round 5 a second pair bracketed every step
    # ("step" events): on this stack an event record is a barrier packet in the queue, the two extra records cost 9 us of GPU
    # timeline per 0.5 ms step (tools/gap_report.sh: 18.8 us between one step's recurrent kernel and the next step's first
    # launch with four records per step, back to back without), i.e. the harness was 2 % of the figure it reported.
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False

    # Dispersion of the headline WITHOUT events in the queue: the same step again in 5 blocks of `steps`,
    # a device synchronisation around each block — outside the timed region above; min / median / max of the block means.
    blocks_ms = None
    if world == 1 and args.steps >= 4:
        bm = []
        for _ in range(5):
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            bm.append((time.perf_counter() - tb) * 1e3 / args.steps)
        bm.sort()
        blocks_ms = {"min": bm[0], "median": bm[2], "max": bm[4], "blocks": 5, "steps_per_block": args.steps}

    # ---- what was timed, checked OUTSIDE the timed region ------------------------------------------------------------
    # (1) the last step's results are finite (train: the loss and every gradient);  (2) the library's device-side event
    # counters — a pair kernel that timed out NaN-poisons its samples, a guard trip means a slower kernel ran than the
    # one the line names;  (3) the kernel families the library routes this descriptor to;  (4) every rank was there.
    import ttrnn_hip
    def _tensors(o):
        if torch.is_tensor(o):
            yield o
        elif isinstance(o, (tuple, list)):
            for v in o:
                for t in _tensors(v):
                    yield t
    checked = list(_tensors(last))
    if args.mode == "train":
        checked += [p.grad for p in model.parameters() if p.grad is not None]
    finite = bool(checked) and all(bool(torch.isfinite(t.float()).all().item()) for t in checked)
    status = ttrnn_hip.device_status()
    rnn_module = model.rnn if args.mode == "train" else model
    spec0 = rnn_module._all_layers[0]._layer_spec()
    tdt = torch.bfloat16 if w["dtype"] == "bf16" else torch.float32
    routes = {"forward": F.rnn_route(spec0, w["B"], w["T"], tdt),
              "backward": F.rnn_backward_route(spec0, w["B"], w["T"], tdt) if args.mode == "train" else None}
    seen = torch.ones(1, dtype=torch.float64, device=device if backend != "gloo" else "cpu")
    if dist is not None:
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    ranks_seen = int(round(float(seen.item())))
    bad = torch.tensor([0.0 if (finite and status["pair_timeouts"] == 0) else 1.0], dtype=torch.float64,
                       device=device if backend != "gloo" else "cpu")
    if dist is not None:
        dist.all_reduce(bad, op=dist.ReduceOp.SUM)
    ranks_bad = int(round(float(bad.item())))
    kern_ms = timer.mean_ms("ttrnn_rnn_forward")
    kern_ms_median = timer.median_ms("ttrnn_rnn_forward")
    step_ms_median = blocks_ms["median"] if blocks_ms else None      # (median of the block means: see above)
    launches_per_step = timer.calls("ttrnn_rnn_forward") / float(max(args.steps, 1))
    if eager_kern_ms is not None:                     # --graph: the recurrent launches' duration from the eager steps before capture
        kern_ms, kern_ms_median, launches_per_step = eager_kern_ms

    # fp32 workloads: which matrix arithmetic ran (include/ttrnn.h TTRNN_MATH_*) and, at N=1, the same steps again
    # in the OTHER mode, so that both figures come from one process on one device (outside the timed region above)
    math_mode = ttrnn_hip.get_fp32_math() if w["dtype"] == "f32" else None
    other = None
    if math_mode is not None and args.mode == "forward" and world == 1:
        alt = "exact" if math_mode == "split" else "split"
        with ttrnn_hip.fp32_math(alt):
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            alt_ms = (time.perf_counter() - ta) * 1e3 / max(args.steps, 1)
        other = {"fp32_math": alt, "ms_per_step": alt_ms, "value": w["T"] / (alt_ms * 1e-3), "unit": "timesteps/s"}

    # the same steps on prepared modules (weight-only work kept across forwards) — beside the headline, never the headline
    prepared_extra = None
    rec_only_ms = None
    if args.mode == "forward" and not prepared and world == 1:
        model.prepare_for_inference()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        # on prepared modules a ttrnn_rnn_forward call is the recurrent kernel(s) alone where the route separates its weight-only
        # launches (input_size == 1 fused-core routes): events around it time the DOMINANT KERNEL by itself — the figure the
        # rocprofv3 kernel statistics under profiles/ must agree with
        # two passes: the wall-clock figure WITHOUT events in the queue (an event record is a barrier packet: two per step cost
        # 9 us of a 0.49 ms step), then the same steps again with a pair around every call for the kernel's duration
        F.KERNEL_TIMER = None
        tu = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        pr_ms = (time.perf_counter() - tu) * 1e3 / max(args.steps, 1)
        ptimer = EventTimer()
        ptimer.enabled = True
        F.KERNEL_TIMER = ptimer
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        F.KERNEL_TIMER = timer
        import ctypes as _ct
        from ttrnn_hip import _lib as _L
        _d = spec0.desc(w["B"], w["T"], _L.TTRNN_BF16 if w["dtype"] == "bf16" else _L.TTRNN_F32)
        if _L.load().ttrnn_rnn_prepare_supported(_ct.byref(_d)):
            rec_only_ms = (ptimer.mean_ms("ttrnn_rnn_forward"), ptimer.median_ms("ttrnn_rnn_forward"))
        prepared_extra = {"ms_per_step": pr_ms, "value": w["T"] / (pr_ms * 1e-3), "unit": "timesteps/s",
                          "what": "prepare_for_inference(): packed cores, scale header, fused-core fragments and (input_size == 1) "
                                  "the unit-row input projection built once and kept across the forwards (ttrnn_rnn_forward_phase)"}
        model.release_prepared()

    el = torch.tensor([elapsed], dtype=torch.float64, device=device if backend != "gloo" else "cpu")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    t_step = elapsed / args.steps

    if rank == 0:
        flop_per_launch = float(w["flop"]) * w["B"] * w["T"] / max(launches_per_step, 1e-9)
        # `dtype` on the JSON line is the ARITHMETIC type: the bf16-storage workload still runs its chain on the
        # exact-fp32 MFMA (bf16 only in HBM), so it is priced against the fp32 peak
        arith = w.get("arith", w["dtype"])
        peak = PEAK_FP32_TFLOPS if arith == "f32" else PEAK_BF16_TFLOPS
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                tr = json.load(fh).get(args.workload)
            if tr and args.mode == "forward":
                traffic = tr["hbm_bytes_per_launch"]       # rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE), see profiles/
        except (OSError, ValueError, KeyError):
            traffic = None
        call_ms = kern_ms
        if rec_only_ms is not None:                    # the recurrent kernel alone (see above); call_ms keeps the whole call
            kern_ms, kern_ms_median = rec_only_ms
        achieved = flop_per_launch / (kern_ms * 1e-3) / 1e12
        # the instruction mix that actually ran, priced on the pipe it ran on: a fraction that cannot exceed 1
        executed = None
        ex = EXECUTED.get(args.workload)
        # (the training forward launches the same kernels with the reserve records switched on: same matrix instructions)
        if ex is not None and math_mode in (None, "split"):
            n_st = float(w["B"]) * w["T"]                      # sample-timesteps per step (all launches of the step)
            bf16_flop = (ex["bf16_mfma"] * 16384.0 + ex["kin_bf16_flop"]) * n_st
            fp32_flop = ex["fp32_mfma"] * 2048.0 * n_st
            step_kernel_s = kern_ms * 1e-3 * launches_per_step
            floor_s = bf16_flop / (PEAK_BF16_TFLOPS * 1e12) + fp32_flop / (PEAK_FP32_TFLOPS * 1e12)
            # matrix-pipe busy share of the RECURRENT kernel on the CUs it occupies: MFMA pipe cycles per SIMD and step
            rec_cycles = (ex["bf16_mfma"] * 16.0 + ex["fp32_mfma"] * 32.0) / ex["rec_simds"]
            pipes = [n for n, f in ((ex.get("pipe16", "bf16_mfma"), bf16_flop), ("fp32_mfma", fp32_flop)) if f]
            per_cu = max(1, -(-w["B"] // 256))                 # samples one CU works through per launch
            executed = {"flop": bf16_flop + fp32_flop, "pipe": " + ".join(pipes),
                        "bf16_mfma_flop": bf16_flop, "fp32_mfma_flop": fp32_flop,
                        "split_terms": ex.get("terms", 6) if (w["dtype"] == "f32" and bf16_flop) else 1,
                        "peak": {"bf16_mfma": PEAK_BF16_TFLOPS, "f16_mfma": PEAK_BF16_TFLOPS, "fp32_mfma": PEAK_FP32_TFLOPS,
                                 "unit": "TFLOP/s"},
                        "pipe_time_at_peak_ms": floor_s * 1e3,
                        "frac": floor_s / step_kernel_s,
                        "mfma_busy_frac": rec_cycles * w["T"] * per_cu / (step_kernel_s * GPU_CLOCK_HZ),
                        "mfma_busy_note": "recurrent kernel(s) only: {:.0f} matrix-pipe cycles per SIMD and sample-timestep x T "
                                          "x {} sample(s) per CU / (measured time of the whole call x 2.4 GHz) = share of "
                                          "the time the matrix pipe of an OCCUPIED CU is busy (B < 256 leaves CUs idle on "
                                          "top of that); PMC-measured counterpart: profiles/".format(rec_cycles, per_cu),
                        "note": ex["note"]}
        frac_algo = achieved / peak
        dense_kin = bool(ex and ex.get("kin_bf16_flop"))
        if executed is not None and (dense_kin or frac_algo > 1.0):
            roof_frac, roof_basis = executed["frac"], "executed MFMA instruction mix / peak of the pipe it ran on (" + executed["pipe"] + ")"
        elif executed is None and frac_algo > 1.0:
            # (exact mode / a workload without a tabulated mix: priced against the FASTEST matrix pipe, which bounds every
            # contraction order on any pipe — a lower bound of the true fraction, never above 1)
            roof_frac = achieved / PEAK_BF16_TFLOPS
            roof_basis = "algorithmic FLOPs / peak of the fastest matrix pipe (16-bit MFMA, {} TFLOP/s): the algorithmic figure exceeds 1 " \
                         "on the arithmetic dtype's own peak and no executed mix is tabulated for this mode — a lower bound".format(PEAK_BF16_TFLOPS)
        else:
            roof_frac = frac_algo
            roof_basis = "algorithmic FLOPs of the reference's chain (SURVEY.md 8(d)) / " + ("fp32" if arith == "f32" else "bf16") + " MFMA peak"
        wg_per_sample = (ex or {}).get("rec_wgs_per_sample", 1)
        wg_per_cu = (ex or {}).get("rec_wgs_per_cu", 1)
        chip_occ = min(1.0, w["B"] * wg_per_sample / float(wg_per_cu) / 256.0)
        line = {
            "metric": ("timesteps/sec/GPU (batch={}) {} h={} ncores={} rank={}" if args.scaling == "weak" else
                       "timesteps/sec of the GLOBAL batch {} sharded over the GPUs, {} h={} ncores={} rank={}").format(
                w["B"] if args.scaling == "weak" else global_batch,
                "TT-LSTM" if w["kind"] == "ttlstm" else "TT-GRU", w["H"], w["d"], w["r"]) +
                      (" (prepared weights)" if prepared else ""),
            # weak: every GPU advances its own batch by T timesteps per step (whole job = N x T); strong: the job is ONE
            # global batch advanced by T timesteps per step
            "value": (world if args.scaling == "weak" else 1) * w["T"] / t_step,
            "unit": "timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_step * 1e3,
            "ms_per_step_median": step_ms_median,
            "ms_per_step_blocks": blocks_ms,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": ("f32 (2xfp16 operands, fp32 accumulate)" if (arith == "f32" and math_mode == "split") else
                      arith if arith == w["dtype"] else "{} ({} storage)".format(arith, w["dtype"])), "data": "synthetic",
            "config": {"workload": w["desc"], "per_gpu_batch": w["B"], "seq_len": w["T"],
                       "global_batch": global_batch, "parallelism": "batch-sharded x{} (no forward collective)".format(world),
                       "mode": ("forward (no_grad), inputs resident in HBM" if args.mode == "forward" else
                                "train step of the reference's benchmarking.py:41-70 (zero_grad + classifier forward + nll_loss "
                                "+ BPTT + Adam; gradient all-reduce for N > 1), inputs resident in HBM; " + optimizer_impl +
                                ("; the whole step captured once and REPLAYED as a hipGraph (--graph)" if args.graph else "")),
                       "fp32_math": FP32_MATH_DESC.get(math_mode),
                       "prepared_weights": (
                           "--prepared: prepare_for_inference() modules — packed cores, scale header, fused-core fragments and "
                           "(input_size == 1) the unit-row input projection are built once and kept across the forwards of the "
                           "timed loop (ttrnn_rnn_forward_phase)" if prepared else
                           "no: every timed forward starts from the TT cores (packs them, rebuilds scales / fragments), as the "
                           "reference's eval loop evaluates the chain from its cores on every call (benchmarking.py:16-38)")},
            "sample_timesteps_per_s": global_batch * w["T"] / t_step,
            "shard_of": (None if args.shard_of <= 1 else
                         "ONE rank's shard of the configuration's batch under --scaling strong on {} GPUs ({} samples), "
                         "timed alone on one GPU: a per-rank cost, not a scaling measurement".format(args.shard_of, w["B"])),
            "collectives": (None if dist is None else
                            "{} process group, world {}{}: barrier + MAX all-reduce of the time{}".format(
                                backend, world, " (forced one-rank group, TTRNN_BENCH_FORCE_DIST=1)" if force_dist and world == 1 else "",
                                "; flat-bucket gradient all-reduce per step" if args.mode == "train" else "")),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         # `frac` is a fraction of a roof and never exceeds 1: the algorithmic figure
                         # (SURVEY 8(d) FLOPs of the reference's chain / the arithmetic dtype's MFMA peak) where the route
                         # evaluates that chain on the TT cores; where it runs a cheaper contraction order (K-in as a dense GEMM of
                         # the materialised matrix: cfg4, cfg5) or the algorithmic figure exceeds 1, the EXECUTED instruction mix
                         # over the peak of the pipe it ran on.  Both are always reported beside it.
                         "frac": roof_frac, "frac_basis": roof_basis,
                         "frac_algorithmic": achieved / peak,
                         "frac_executed": (executed or {}).get("frac"),
                         "pipe": (executed or {}).get("pipe"),
                         "chip_occupancy": chip_occ,
                         "chip_occupancy_note": "CUs holding a workgroup of the recurrent kernel / 256 (one sample per workgroup; "
                                                "batch < 256 leaves CUs idle by construction)",
                         "traffic": traffic,
                         "launch": ("the forward call inside the training step (reserve records on); the executed mix is the "
                                    "forward's" if args.mode == "train" else "forward"),
                         "kernel": ("the persistent recurrent kernel alone (events around ttrnn_rnn_forward_phase(RUN) on prepared "
                                    "modules, same process)" if rec_only_ms is not None else
                                    "ttrnn_rnn_forward (K-in batched input projection + K-rec persistent recurrent kernel)"),
                         "kernel_ms": kern_ms, "kernel_ms_median": kern_ms_median,
                         "call_ms": call_ms,
                         "events": "HIP events on the launch stream around every {} library call of the timed loop (call_ms, "
                                   "{} pairs){}".format("" if timer.sample == 1 else "%dth" % timer.sample, timer.count("ttrnn_rnn_forward"),
                                                        "; kernel_ms: around EVERY call of the prepared re-run in the same process"
                                                        if rec_only_ms is not None else ""),
                         "basis": "algorithmic FLOPs of the reference's stage-by-stage chain (SURVEY.md 8(d)) over the "
                                  "MFMA peak of the arithmetic dtype; the fused-core / split-math kernels execute "
                                  "fewer FLOPs, on the bf16 MFMA (DESIGN.md 4a, 8)",
                         # input_size == 1 workloads evaluate the input chain on two unit rows and scale by x_t
                         # (W_in x is linear in a scalar): `achieved` prices the reference's ALGORITHMIC FLOPs
                         # (SURVEY.md 8(d)); this is the same figure with the input chain's share left out
                         "achieved_hidden_chain_only": achieved * (1.0 - w.get("flop_in", 0) / float(w["flop"])),
                         "flop_per_launch": flop_per_launch,
                         "frac_note": "algorithmic basis: can exceed 1 where the fused contraction order executes fewer "
                                      "FLOPs than the reference's chain on a faster pipe; `executed` is the bounded figure",
                         "executed": executed},
        }
        if reducer is not None:
            # what the gradient exchange of a step costs on the device: events on the launch stream around
            # the whole sync (bucket copy-in, all-reduce, scale, copy-out) and around the all-reduce call alone, rank 0
            line["collectives_measured"] = {
                "bucket_bytes": reducer.nbytes, "world": world,
                "grad_sync_ms_mean": timer.mean_ms("grad_sync"), "grad_sync_ms_median": timer.median_ms("grad_sync"),
                "all_reduce_ms_mean": timer.mean_ms("all_reduce"), "all_reduce_ms_median": timer.median_ms("all_reduce"),
                "calls": timer.count("all_reduce"),
                "note": "one flat fp32 bucket, one all-reduce (SUM) per step + 1/world scale; device time on rank 0 between "
                        "events recorded on the launch stream (the RCCL kernel runs on its own stream: the pair brackets the "
                        "wait for it); no multi-GPU scaling curve has been measured so far (no 8-GPU node in rounds 1 - 5)"}
        line["ranks_seen"] = ranks_seen
        line["checked"] = {"finite": finite, "tensors": len(checked), "ranks_bad": ranks_bad,
                           "what": ("loss + every parameter gradient of the last timed step" if args.mode == "train" else
                                    "outputs and final state of the last timed step") + ", torch.isfinite on rank 0; "
                                   "ranks_bad = ranks whose results were not finite or that counted a pair time-out",
                           "device_status": status, "routes": routes}
        if other is not None:
            line["other_fp32_math"] = other
        if prepared_extra is not None:
            line["prepared"] = prepared_extra
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, rnn_module.state_dict(), x)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        payload = json.dumps(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line is the last thing on stdout: RCCL writes its version banner through C stdio, which — on a pipe —
        # would otherwise come out at process exit, behind the line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass
        print(payload, flush=True)
    if ranks_seen != world:
        sys.stderr.write("bench.py: {} of {} ranks took part\n".format(ranks_seen, world))
        return 3
    if ranks_bad:
        sys.stderr.write("bench.py: the timed steps are not valid on {} rank(s): finite={} device_status={}\n".format(
            ranks_bad, finite, status))
        return 4
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
