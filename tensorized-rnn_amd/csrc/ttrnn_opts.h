// ttrnn_opts.h — process-wide library options (kernel-route A/B switches and the fp32 math mode).
// Read ONCE from the TTRNN_* environment variables at first use, afterwards only through ttrnn_set_option /
// ttrnn_get_option (include/ttrnn.h): no getenv on any launch path, and every field is an atomic, so concurrent
// host threads may launch while another thread flips a switch.
#pragma once
#include <atomic>

namespace ttrnn {

enum OptId {
  OPT_FP32_MATH = 0,     // TTRNN_FP32_MATH = exact | split   (TTRNN_MATH_EXACT / TTRNN_MATH_SPLIT)
  OPT_FORCE_GENERIC,     // TTRNN_FORCE_GENERIC=1   any-shape VALU kernels for everything
  OPT_NO_GEMM,           // TTRNN_NO_GEMM=1         batched projections / gradients through the TT chain kernels
  OPT_NO_IN1,            // TTRNN_NO_IN1=1          no input_size == 1 shortcut
  OPT_NO_F10,            // TTRNN_NO_F10=1          no fused-core kernels (stage-wise MFMA kernels instead)
  OPT_NO_G2,             // TTRNN_NO_G2=1           no runtime-shape two-stage MFMA kernels (any-shape VALU kernels instead)
  OPT_FORCE_G2,          // TTRNN_FORCE_G2=1        runtime-shape kernels even where a shape-specialised kernel exists (A/B)
  OPT_DIAG,              // TTRNN_DIAG=1            diagnostic builds with s_memtime stamps
  OPT_BF16_FP32_MFMA,    // TTRNN_BF16_FP32_MFMA=1  bf16 storage on the fp32 MFMA kernels
  OPT_BIG_MERGE,         // TTRNN_BIG_MERGE=0|1|2   pairs of cores contracted per launch (big shape)
  OPT_BIG_NO_GEMM,       // TTRNN_BIG_NO_GEMM=1
  OPT_BIG_NO_PAIR,       // TTRNN_BIG_NO_PAIR=1     one workgroup per sample (big shape)
  OPT_NO_BIGB,           // TTRNN_NO_BIGB=1         any-shape backward for the big shape
  OPT_BIGW_SLICES,       // TTRNN_BIGW_SLICES=1
  OPT_F10_NB1,           // TTRNN_F10_NB1=1         one sample per workgroup even when B > #CUs
  OPT_DENSE_FP32,        // TTRNN_DENSE_FP32=1      dense weight gradient on the fp32 MFMA
  OPT_F10_NB2,           // TTRNN_F10_NB2=1         B > #CUs: two samples per eight-wave workgroup instead of four-wave workgroups (A/B)
  OPT_GEMM_PIECES,       // TTRNN_GEMM_PIECES=0|2|3 forward input-projection GEMMs: 0 by size, 2 two fp16 pieces, 3 three bf16 pieces (A/B)
  OPT_BIG_FP32_MFMA,     // TTRNN_BIG_FP32_MFMA=1   big-shape pair kernel on the fp32 MFMA even in split mode (A/B)
  OPT_PAIR_FAULT,        // TTRNN_PAIR_FAULT=1      tests only: the forward pair kernels are launched one workgroup short, so that
                         //                         the time-out path (NaN poison + device status counter) can be exercised
  OPT_NO_GEMM3,          // TTRNN_NO_GEMM3=1        two-piece fp16 K-in GEMM with x split on the fly (round 2's kernel) instead of the
                         //                         pre-split LDS-DMA GEMM (A/B)
  OPT_DEV,               // TTRNN_DEV=0..1073741823     developer bit mask: A/B ROUTE switches between kernels that compute the same result
                         //                         (1: gemm3 ping-pong schedule, 4: f10gq for bf16 GRU, 32: eight-wave GRU kernel,
                         //                         64: gemm3 one workgroup per tile, 1024: fused-core wgrad on unit rows, 4096: g2 four-wave streamed plans where the eight-wave plan would keep the head resident (forward: sixteen slots; reverse: twelve), 8192 / 16384: H = 512 r = 8 forward / reverse recurrence on the runtime tier, 32768 / 128: four-barrier / wave-local fused-core reverse LSTM kernel everywhere, 2048: g2 head
                         //                         256: fp32 GRU H = 256 r = 8 forward on the runtime tier instead of k_gru_fwd_f10vh, 512: the single-barrier LSTM forward kernel k_lstm_fwd_f10s instead of k_lstm_fwd_f10q, 65536: no fused set-up launch (ttrnn_rnn_forward_cores = pack + forward), 262144: head forward as chain + separate epilogue launch, 536870912: the tier's reverse kernel resident for one column tile only, 268435456: the dense weight gradient's row ranges by the one-round rule everywhere, 134217728: the tier's forward plans without column tiles inside a stage-2 unit (G2Mat::cin), 67108864: TT-GRU r = 16 reverse recurrence on the stage-wise kernel instead of the runtime tier's, 33554432: naive TT-LSTM / TT-GRU H = 256 on the runtime tier instead of k_rnn_fwd_f10n, 16777216: the tier's K-in dense matrix from the merged cores (k_g2_dense) instead of the chain kernel on identity rows (measured: no difference), 8388608: d = 4 matrices' dense weight gradient pulled back onto the cores by the any-shape chain kernel instead of ttrnn_fast_proj.hip's launches, 2097152: runtime-tier reverse kernel on four-wave workgroups even where only one fits a CU, 4194304: its streamed T2 on the general per-position loop, 524288 / 1048576: the runtime tier's two-samples-per-workgroup forward kernel k_g2_fwd_p never / for every batch of two or more (default: more samples than CUs and a streamed head), 131072: H = 256 reverse LSTM kernel with the record-dependent gate factors precomputed by the helper waves (k_lstm_bwd_f10p instead of k_lstm_bwd_f10h; measured slower),
                         //                         fragments not resident).  Bits 2 / 8 / 16 (and 32 / 64 inside the dense weight
                         //                         gradient) are result-destroying ablations that exist ONLY in -DTTRNN_ABLATIONS
                         //                         builds (`make ablation` -> tools/bin/libttrnn_abl.so), never in libttrnn.so
  OPT_DEV2,              // TTRNN_DEV2=0..1073741823    second developer bit mask (round 6; same kind of A/B route switches as `dev`):
                         //                         1: no chain weight gradient (ttrnn_rnn_wgrad_workspace answers 0: dense gradients everywhere),
                         //                         2: chain weight gradient for the hidden matrix only (the input matrix keeps its dense pass over dy),
                         //                         4: the chain kernel with the run-time plan even for the shapes that have a compile-time instantiation,
                         //                         8: offer the chain kernel's LARGE variant too (measured slower than the dense gradient: tests only),
                         //                         16: the speaker encoder's shape on the runtime tier instead of k_lstm_fwd_w2 (forward), 32: the same for the reverse-time kernel,
                         //                         64: k_c2w (C1 / dC1 images in LDS) for the calls k_c2r (register hand-offs) would take,
                         //                         128: TT-GRU H = 256 r = 16 reverse recurrence on the tier's kernel, 256: the same for the TT-GRU of H = 512 r = 8,
                         //                         512: naive per-gate sets of H = 256 on the tier's reverse-time kernel (instead of k_rnn_bwd_f10n),
                         //                         1024: the four-barrier fused-core LSTM reverse kernel instead of its wave-local form (k_lstm_bwd_f10h<.., WL>),
                         //                         (-DTTRNN_ABLATIONS builds only, tools/c2w_bench.py: bits 16 and up are the chain kernels' ablation switches)
  OPT_COUNT
};

int opt(OptId id);                       // current value
const char* opt_name(OptId id);          // "fp32_math", "force_generic", ...
int opt_set(const char* name, int value);   // 0 or -1 (unknown name / bad value)
int opt_get(const char* name, int* value);

}  // namespace ttrnn
