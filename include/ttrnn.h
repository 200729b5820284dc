/*
 * ttrnn.h — C ABI of libttrnn.so: tensor-train LSTM / GRU hot path for AMD MI355X (gfx950).
 *
 * The reference (onucharles/tensorized-rnn) has no FFI / operator registry: its boundary for this
 * path is the Python nn.Module API of `tensorized_rnn` + `t3nsor`.  Every entry point below names
 * the reference code (file:line under the reference repo) whose arithmetic it replaces; the host
 * side in `tensorized-rnn_amd/` re-creates that module API on top of these calls (INTEGRATION.md
 * shows the ctypes binding a maintainer of the reference would add).
 *
 * Conventions
 *   - plain C types only; all tensor arguments are raw DEVICE pointers owned by the caller;
 *   - every launch goes to the caller's stream (`void* stream` is a hipStream_t, NULL = default);
 *   - no allocation, no synchronisation, no host<->device copy inside any call except the
 *     documented ttrnn_*_workspace queries (pure host arithmetic) — graph-capture safe;
 *   - return value: TTRNN_OK (0) or a negative ttrnn_status; nothing is launched on error;
 *   - batch-first contiguous layouts: x[B][T][in], out[B][T][H], h/c[B][H]  (lstm.py:117, gru.py:118);
 *   - storage dtype `dtype` applies to x / out / h0 / c0 / hT / cT / bias / cores; packed cores,
 *     recurrent state, gate arithmetic and accumulation are always fp32.
 *
 * TT-matrix convention (t3nsor/ops.py:54-93, t3nsor/layers.py:121-127):
 *   y[n][o] = sum G_0[0,i_0,j_0,a_1] * G_1[a_1,i_1,j_1,a_2] ... G_{d-1}[a_{d-1},i_{d-1},j_{d-1},0] * x[n][i] + bias[o]
 *   o = ((i_0*I_1+i_1)*I_2+i_2)...,  i = ((j_0*J_1+j_1)*J_2+j_2)...   (index 0 most significant)
 *   core k is the reference's Parameter `weight_t.tt_cores[k]`: logical shape (R_k, I_k, J_k, R_{k+1}),
 *   arbitrary element strides (the reference stores it as a transposed view, tensor_train.py:104-114).
 */
#ifndef TTRNN_H_
#define TTRNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTRNN_ABI_VERSION 7
#define TTRNN_MAX_D 6          /* n_cores (+1 for new_core='first'/'last', rnn_utils.py:29-34) */

typedef enum ttrnn_status {
  TTRNN_OK = 0,
  TTRNN_ERR_BAD_DESC = -1,     /* inconsistent modes / ranks / sizes                          */
  TTRNN_ERR_NULL = -2,         /* a required pointer is NULL                                  */
  TTRNN_ERR_UNSUPPORTED = -3,  /* legal but not implemented (dtype / d / size combination)    */
  TTRNN_ERR_WORKSPACE = -4,    /* workspace smaller than ttrnn_*_workspace() reported          */
  TTRNN_ERR_LAUNCH = -5        /* HIP refused the launch (hipGetLastError != hipSuccess)      */
} ttrnn_status;

typedef enum ttrnn_dtype { TTRNN_F32 = 0, TTRNN_BF16 = 1 } ttrnn_dtype;
typedef enum ttrnn_cell { TTRNN_LSTM = 0, TTRNN_GRU = 1 } ttrnn_cell;

/* One TT-matrix = the weight of one TTLinear (t3nsor/layers.py:83-127). */
typedef struct ttrnn_ttm {
  int32_t d;                          /* number of cores                                      */
  int32_t in_modes[TTRNN_MAX_D];      /* J_k, prod = in features                              */
  int32_t out_modes[TTRNN_MAX_D];     /* I_k, prod = out features                             */
  int32_t ranks[TTRNN_MAX_D + 1];     /* R_0 .. R_d, R_0 = R_d = 1                            */
} ttrnn_ttm;

/* One recurrent layer (one TTLSTMCell / TTGRUCell unrolled over the sequence):
 * tensorized_rnn/lstm.py:23-41,101-135; gru.py:25-50,104-136; tt_lstm.py:16-40; gru.py:148-172. */
typedef struct ttrnn_rnn_desc {
  int32_t cell;          /* ttrnn_cell: LSTM gate order i,f,g,o (lstm.py:26-29); GRU r,z,n (gru.py:38-44) */
  int32_t dtype;         /* ttrnn_dtype of x/out/state tensors                                 */
  int32_t batch;         /* B                                                                  */
  int32_t seq_len;       /* T (1 = single cell step, LSTMCell.forward / GRUCell.forward)       */
  int32_t input_size;    /* in  = prod(in_w.in_modes)                                          */
  int32_t hidden_size;   /* H   ; prod(out_modes) = n_gates*H (4 LSTM, 3 GRU)                  */
  int32_t has_bias_in;   /* TTLSTMCell/TTGRUCell put a bias on BOTH TTLinears (tt_lstm.py:26,39) */
  int32_t has_bias_hid;
  ttrnn_ttm in_w;        /* cell.input_weights                                                 */
  ttrnn_ttm hid_w;       /* cell.hidden_weights                                                */
  int32_t hid_blocks;    /* 0 / 1: nothing known about hid_w's cores.  G > 1: a PROMISE that hid_w is the direct sum of G
                          * TT-matrices written as one train — a leading (1 -> G) selector core followed by cores that are
                          * block-diagonal in their rank indices (rank = G * r, block g couples only to block g): the
                          * naive per-gate TTLinearSet (tt_linearset.py:5-38) presented as one matrix.  The kernels may
                          * then skip the zero blocks; results equal those of hid_blocks = 0 on such cores. */
} ttrnn_rnn_desc;

/* ---- library ------------------------------------------------------------------------------- */
int ttrnn_abi_version(void);
const char* ttrnn_status_string(int status);
/* 1 when a usable HIP device is visible to this process, else 0 (never launches). */
int ttrnn_device_available(void);

/* Device-side event counters.  No call above synchronises, so a kernel cannot hand a status back through its launch; the
 * events a caller must be able to see are counted on the device instead and read HERE (this call synchronises the device):
 *   counters[TTRNN_STAT_PAIR_TIMEOUTS]  threads of the two-workgroups-per-sample kernels (H = 1024 class, 2B <= #CUs) that gave
 *                                       up waiting for their partner workgroup (~0.1 s: the partner was not resident — CUs held
 *                                       by another stream or process).  The affected samples' outputs are NaN by construction.
 *   counters[TTRNN_STAT_GUARD_TRIPS]    launches that left the two-piece fp16 kernel for the fp32-MFMA one because a few large
 *                                       entries had pushed the bulk of a weight operand into fp16's subnormal range.
 *   counters[TTRNN_STAT_BLOCK_VIOLATIONS]  head rows of a hidden matrix passed with hid_blocks > 1 that had a non-zero entry
 *                                       outside their gate's rank block: the hid_blocks promise was false and the
 *                                       runtime-shape kernels ignored those entries (outputs and gradients are those of the
 *                                       block-diagonal part).  Checked on every launch that uses the promise.
 * n = number of counters wanted (<= TTRNN_STAT_COUNT); reset != 0 zeroes them after reading.  The reference has no
 * counterpart (its failures are Python exceptions, lstm.py / ops.py raise nothing on this path). */
#define TTRNN_STAT_PAIR_TIMEOUTS 0
#define TTRNN_STAT_GUARD_TRIPS 1
#define TTRNN_STAT_BLOCK_VIOLATIONS 2
#define TTRNN_STAT_COUNT 4
int ttrnn_device_status(unsigned int* counters, int n, int reset);

/* How fp32 tensors are multiplied inside the shape-specialised recurrent kernels (process-wide; the analogue of
 * torch.backends.*.matmul precision switches — the reference itself has none, its fp32 GEMMs are whatever the
 * BLAS under torch.einsum does, t3nsor/ops.py:85-91):
 *   TTRNN_MATH_EXACT  v_mfma_f32_16x16x4_f32 on the fp32 operands;
 *   TTRNN_MATH_SPLIT  every fp32 operand is split into 16-bit pieces and the product is rebuilt from 16-bit MFMA terms
 *                     with fp32 accumulation (error-compensated products).  Two flavours, chosen by the kernel:
 *                     (a) three bf16 pieces (x = x0+x1+x2 exactly), the six terms of weight >= 2^-18: per-product
 *                         error < 2^-24 relative, 2.67x less matrix-pipe time than TTRNN_MATH_EXACT;
 *                     (b) the fused-core / merged-core kernels and the dense GEMMs: two fp16 pieces under DIAGONAL
 *                         power-of-two scales — one per row and one per column of every weight operand (two-sided: an
 *                         outlier entry costs its own row and column range, not the whole matrix), one per hidden unit /
 *                         per GEMM row for the activations (no overflow for any finite input) — terms
 *                         x0w0 + x0w1 + x1w0: operands carry >= 22 significand bits, per-product error <= 2^-21.4
 *                         relative — below the rounding an fp32 sum of 256 such products accumulates anyway; 5.3x
 *                         less matrix-pipe time.  A weight operand whose bulk would still land in fp16's subnormal
 *                         range leaves for the fp32-MFMA kernel on the device (TTRNN_STAT_GUARD_TRIPS counts it).
 *                     Measured against a float64 evaluation both flavours sit where TTRNN_MATH_EXACT sits (DESIGN.md
 *                     section 4a; tests/test_gpu_parity.py::test_split_math_*).
 * Default: TTRNN_MATH_SPLIT where a split kernel exists for the descriptor (DESIGN.md section 5 lists them and gives
 * the measured error of both modes against a float64 evaluation); the environment variable TTRNN_FP32_MATH =
 * "exact" | "split", read at first use, overrides the default; ttrnn_set_fp32_math overrides both.
 * Storage dtype, accumulators, gate math and state are unaffected; TTRNN_BF16 descriptors ignore the switch. */
#define TTRNN_MATH_EXACT 0
#define TTRNN_MATH_SPLIT 1
int ttrnn_set_fp32_math(int mode);      /* TTRNN_OK or TTRNN_ERR_UNSUPPORTED */
int ttrnn_get_fp32_math(void);

/* Library options: the kernel-route switches used for A/B measurements and by the parity tests that compare two routes
 * on one input (the reference has no counterpart; its only "route" is ATen's dispatcher).  All options live in ONE
 * table that is filled once, at first use, from the environment variables TTRNN_<NAME> (upper case) and is afterwards
 * changed only through ttrnn_set_option — no getenv on any launch path; the fields are atomics, so a host thread may
 * flip a switch while others launch (a launch reads each switch once).  Names (value 0 / 1 unless noted):
 *   "fp32_math" (TTRNN_MATH_*), "force_generic", "no_gemm", "no_in1", "no_f10", "no_g2", "force_g2", "diag",
 *   "bf16_fp32_mfma",
 *   "big_merge" (0..2), "big_no_gemm", "big_no_pair", "no_bigb", "bigw_slices", "f10_nb1", "dense_fp32", "f10_nb2",
 *   "gemm_pieces" (0 | 2 | 3), "big_fp32_mfma", "pair_fault" (tests only: exercises the pair kernels' time-out path),
 *   "no_gemm3", "dev" (0..1073741823: developer bit mask, A/B route switches between kernels that compute the same result;
 *   csrc/ttrnn_opts.h lists the bits).
 * Workspace sizes must be queried under the same options the launch will run with.
 * Returns TTRNN_OK, or TTRNN_ERR_UNSUPPORTED for an unknown name / value out of range. */
int ttrnn_set_option(const char* name, int value);
int ttrnn_get_option(const char* name, int* value);

/* ---- weights: strided reference Parameters <-> packed fp32 cores ----------------------------
 * Packed layout (fp32): for k = 0..d-1   W_k [K_k = J_k*R_{k+1}][M_k = I_k*R_k],
 *   W_k[(j*R_{k+1}+b)*M_k + (i*R_k+a)] = G_k[a,i,j,b]          (the stage-k GEMM operand),
 * followed by the same matrices transposed, Wt_k[M_k][K_k] (operand of the backward chain).
 * ttrnn_packed_elems() = 2 * sum_k K_k*M_k floats. */
int64_t ttrnn_packed_elems(const ttrnn_ttm* w);
/* replaces: t3nsor/ops.py:47-51 `transpose` + tensor_train.py:104-114 (parameter views) as seen by
 * tt_dense_matmul.  cores[k]: device pointer of core k; strides[4*k..4*k+3]: its element strides
 * for the logical (R_k, I_k, J_k, R_{k+1}) axes. */
int ttrnn_pack_cores(const ttrnn_ttm* w, const void* const* cores, const int64_t* strides,
                     int dtype, float* packed, void* stream);
/* the same for TWO TT-matrices (one recurrent layer's input_weights and hidden_weights, tt_lstm.py:16-40) in one launch:
 * the per-call launch count matters at sub-millisecond sequences */
int ttrnn_pack_cores2(const ttrnn_ttm* wa, const void* const* cores_a, const int64_t* strides_a, float* packed_a,
                      const ttrnn_ttm* wb, const void* const* cores_b, const int64_t* strides_b, float* packed_b,
                      int dtype, void* stream);
/* inverse for gradients: scatters the first half (W_k grads) of `packed_grad` into strided
 * per-core gradient tensors of the given dtype (overwrite, not accumulate). */
int ttrnn_unpack_core_grads(const ttrnn_ttm* w, const float* packed_grad, void* const* core_grads,
                            const int64_t* strides, int dtype, void* stream);

/* ---- TTLinear ------------------------------------------------------------------------------
 * replaces: TTLinear.forward t3nsor/layers.py:121-127 -> tt_dense_matmul t3nsor/ops.py:54-93
 * y[n_rows][out] = TT(packed) x[n_rows][in] (+ bias).  bias may be NULL. */
size_t ttrnn_ttlinear_workspace(const ttrnn_ttm* w, int64_t n_rows);
int ttrnn_ttlinear_forward(const ttrnn_ttm* w, int dtype, int64_t n_rows, const float* packed,
                           const void* bias, const void* x, void* y,
                           void* workspace, size_t workspace_bytes, void* stream);
/* autograd of the above (reference: torch autograd through ops.py:81-90).  Any of dx /
 * d_packed / d_bias may be NULL to skip it.  d_packed (fp32, ttrnn_packed_elems floats, first
 * half used) and d_bias (fp32[out]) are ACCUMULATED into: zero them first.
 * `dtype` is the storage type of x and dx; `dy_dtype` that of dy (the recurrent backward hands
 * fp32 gate gradients to bf16 layers). */
int ttrnn_ttlinear_backward(const ttrnn_ttm* w, int dtype, int dy_dtype, int64_t n_rows, const float* packed,
                            const void* x, const void* dy, void* dx, float* d_packed,
                            float* d_bias, void* workspace, size_t workspace_bytes, void* stream);

/* The same with hints from the producer of the operands, which save this call its own passes over the rows (all device pointers,
 * any may be NULL, hints == NULL: exactly ttrnn_ttlinear_backward):
 *   x_colmax  fp32[in]  UPPER BOUNDS of max_n |x[n][j]|  (rows that are LSTM hidden states: all 1.0)
 *   dy_colmax fp32[out] UPPER BOUNDS of max_n |dy[n][o]| (ttrnn_rnn_backward_ex: stats rows 0 / 1)
 *             With dy's bounds the dense weight gradient runs on two fp16 pieces under per-column scales
 *             at every size; a bound BELOW the true maximum overflows the fp16 pieces (inf / NaN in d_packed).
 *   xdy_sum   fp32[out] in_size == 1, dx == NULL, d_bias == NULL: sum_n x[n] dy[n][:] (stats row 2) — the weight gradient then
 *             needs no pass over dy at all (the bias gradient is stats row 3).
 *   x_period  > 0: `x` is the output out[n_rows / x_period][x_period][in] of a recurrent layer and the operand's row n is the
 *             PREVIOUS step's state: row n - 1 of out, or row n / x_period of x_first (storage dtype; NULL = zeros) where
 *             n % x_period == 0 — the h_{t-1} rows of the hidden matrix's weight gradient (lstm.py:123-133) read in place
 *             instead of being materialised by the caller.  Only the dense-gradient routes read rows this way: ask
 *             ttrnn_ttlinear_backward_shift_ok first; TTRNN_ERR_UNSUPPORTED elsewhere.  (x_colmax then bounds these rows.) */
typedef struct ttrnn_lin_hints {
  const float* x_colmax;
  const float* dy_colmax;
  const float* xdy_sum;
  int64_t x_period;
  const void* x_first;
  const float* dy_rowmax;   /* ABI 5.  fp32[n_rows] or NULL: an UPPER BOUND of max_c |dy[n][c]| per row — with it the input
                             * gradient dx = dy W^T of a stacked layer runs as a GEMM on two fp16 pieces (three MFMA terms, one
                             * power-of-two scale per row of dy and per column of W^T) instead of three bf16 pieces (six terms)
                             * without a pass over dy for its row maxima; the reverse-time kernel knows them
                             * (TTRNN_BWD_STATS_ROWMAX: the exact maximum of every step's gate gradients is what it scales
                             * its own operands by). */
} ttrnn_lin_hints;
/* 1 if a ttrnn_ttlinear_backward_hinted call with these arguments (d_packed wanted, dx as said, a workspace of
 * ttrnn_ttlinear_workspace bytes) takes a route that honours hints->x_period under the current options. */
int ttrnn_ttlinear_backward_shift_ok(const ttrnn_ttm* w, int dtype, int dy_dtype, int64_t n_rows, int64_t x_period,
                                     int want_dx);
int ttrnn_ttlinear_backward_hinted(const ttrnn_ttm* w, int dtype, int dy_dtype, int64_t n_rows, const float* packed,
                                   const void* x, const void* dy, void* dx, float* d_packed, float* d_bias,
                                   const ttrnn_lin_hints* hints, void* workspace, size_t workspace_bytes, void* stream);

/* ---- TTLinear heads with fused row-wise epilogues ---------------------------------------------------------------
 * The step right after the path in both callers:
 *   TTRNN_EPI_LOG_SOFTMAX   y = log_softmax(TT(x) + b, dim=1)      experiments/digit_classification/mnist_classifier.py:55-57
 *   TTRNN_EPI_RELU_L2NORM   u = relu(TT(x) + b);  y = u / ||u||_2  experiments/speaker_verification/encoder/speaker_encoder.py:86-89
 * y has the storage dtype; aux (fp32[n_rows], may be NULL for inference with LOG_SOFTMAX) receives the row's log-sum-exp /
 * L2 norm.  ttrnn_head_backward takes the SAVED outputs y (+ aux), turns dy into the pre-activation gradient and runs
 * ttrnn_ttlinear_backward on it (same accumulate / NULL conventions).  Workspace: ttrnn_head_workspace(w, n_rows). */
#define TTRNN_EPI_NONE 0
#define TTRNN_EPI_LOG_SOFTMAX 1
#define TTRNN_EPI_RELU_L2NORM 2
size_t ttrnn_head_workspace(const ttrnn_ttm* w, int64_t n_rows);
int ttrnn_head_forward(const ttrnn_ttm* w, int dtype, int epilogue, int64_t n_rows, const float* packed, const void* bias,
                       const void* x, void* y, float* aux, void* workspace, size_t workspace_bytes, void* stream);
int ttrnn_head_backward(const ttrnn_ttm* w, int dtype, int epilogue, int64_t n_rows, const float* packed, const void* x,
                        const void* y, const float* aux, const void* dy, void* dx, float* d_packed, float* d_bias,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- recurrent layer -----------------------------------------------------------------------
 * replaces: the Python time loop LSTM.forward lstm.py:123-133 / GRU.forward gru.py:124-134 for ONE
 * layer (layer l+1 consumes layer l's `out`; same result as the reference's step-major loop),
 * LSTMCell.forward lstm.py:23-32 / GRUCell.forward gru.py:25-44 and both TTLinear chains.
 *   x[B][T][in], h0/c0[B][H] (NULL = zeros, lstm.py:88-91) -> out[B][T][H], hT/cT[B][H] (may be NULL).
 *   c0 / cT are ignored for GRU.
 *   reserve: NULL for inference; else fp32, ttrnn_rnn_reserve_bytes long, saved for ttrnn_rnn_backward: LSTM
 *            [B][T][H][4] (per hidden unit the activated gates i,g,f,o) followed by [B][T][H] (the cell states c_t);
 *            GRU [B][T][H][4] (r,z,n, hidden_part_n). */
/* out == NULL: the caller consumes only the final state (experiments/speaker_verification/encoder/speaker_encoder.py:80-86 keeps
 * `hidden[-1]` of the last layer and drops its outputs; experiments/digit_classification/mnist_classifier.py:52-55 classifies
 * the last step): accepted where ttrnn_rnn_out_optional(desc) returns 1 — on every route for an inference call (reserve ==
 * NULL; every forward kernel guards its [B][T][H] store) — and TTRNN_ERR_NULL for a training forward (reserve != NULL: the
 * weight-gradient step reads the rows of `out` as h_{t-1}). */
int ttrnn_rnn_out_optional(const ttrnn_rnn_desc* desc);
size_t ttrnn_rnn_workspace(const ttrnn_rnn_desc* desc);
size_t ttrnn_rnn_reserve_bytes(const ttrnn_rnn_desc* desc);
int ttrnn_rnn_forward(const ttrnn_rnn_desc* desc, const void* x, const void* h0, const void* c0,
                      const float* packed_in, const void* bias_in,
                      const float* packed_hid, const void* bias_hid,
                      void* out, void* hT, void* cT, float* reserve,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Inference with unchanged weights (the reference's eval loop, experiments/digit_classification/benchmarking.py:16-38, calls
 * the model again and again on fixed parameters): part of what ttrnn_rnn_forward launches depends on the WEIGHTS only — the
 * fused core's scale header and fragments, and for input_size == 1 the input projection of the two unit rows — and lands in
 * `workspace`.  A caller that keeps packed cores, biases and workspace alive may split the call:
 *   TTRNN_PHASE_ALL      = ttrnn_rnn_forward;
 *   TTRNN_PHASE_PREPARE  only the weight-dependent launches (x, h0, c0, out, hT, cT, reserve are ignored and may be NULL);
 *   TTRNN_PHASE_RUN      skips them: `workspace` must still hold what a PREPARE (or ALL) call with the SAME descriptor, options,
 *                        packed cores and biases left there — the caller's responsibility; stale contents are a silent wrong
 *                        answer, which is why this is an explicit entry point and not a cache inside the library.
 * ttrnn_rnn_prepare_supported: 1 when the descriptor's route splits this way under the current options (input_size == 1 on the
 * fused-core kernels: 1 launch per forward instead of 4); otherwise PREPARE is a no-op and RUN does everything. */
#define TTRNN_PHASE_ALL 0
#define TTRNN_PHASE_PREPARE 1
#define TTRNN_PHASE_RUN 2
int ttrnn_rnn_prepare_supported(const ttrnn_rnn_desc* desc);
int ttrnn_rnn_forward_phase(const ttrnn_rnn_desc* desc, int phase, const void* x, const void* h0, const void* c0,
                            const float* packed_in, const void* bias_in,
                            const float* packed_hid, const void* bias_hid,
                            void* out, void* hT, void* cT, float* reserve,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ABI 6.  ttrnn_pack_cores2 + ttrnn_rnn_forward in ONE call, for callers whose weights may have changed since the last call (a
 * training loop, the module API of tensorized_rnn/lstm.py:101-135 whose Parameters an optimizer updates in place): `cores_*` /
 * `strides_*` as ttrnn_pack_cores (t3nsor/ops.py:47-51 parameter views, storage dtype = desc->dtype), `packed_in` / `packed_hid`
 * (ttrnn_packed_elems floats each) are WRITTEN — the backward entry points read them.  Results are bit-identical to the two calls.
 * ttrnn_rnn_forward_cores_fused: 1 when the descriptor's route does all of its weight-only work (packing, the unit-row input
 * projection of input_size == 1, scale header, MFMA fragments) in one set-up launch under the current options — two launches per
 * forward instead of five (BASELINE configs[1]: 0.518 -> 0.49 ms per call); 0: the call is exactly pack + forward. */
int ttrnn_rnn_forward_cores_fused(const ttrnn_rnn_desc* desc);
int ttrnn_rnn_forward_cores(const ttrnn_rnn_desc* desc, const void* x, const void* h0, const void* c0,
                            const void* const* cores_in, const int64_t* strides_in, const void* bias_in,
                            const void* const* cores_hid, const int64_t* strides_hid, const void* bias_hid,
                            float* packed_in, float* packed_hid, void* out, void* hT, void* cT, float* reserve,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Which family of kernels ttrnn_rnn_forward would run for this descriptor under the current options (pure host logic, no
 * launch): the benchmark sweep and the tests report / assert it, so that "an MFMA kernel ran" is a checked statement. */
#define TTRNN_ROUTE_VALU 0            /* any-shape VALU kernels (ttrnn_generic.hip)                                   */
#define TTRNN_ROUTE_STAGEWISE_MFMA 1  /* shape-specialised stage-wise chain on the MFMA (ttrnn_fast*.hip)              */
#define TTRNN_ROUTE_FUSED_CORE 2      /* shape-specialised fused-core kernels (ttrnn_fast_f10*.hip)                    */
#define TTRNN_ROUTE_MERGED_BIG 3      /* merged two-core kernels of the H = 1024, d = 4, r = 32 shape (ttrnn_fast_big) */
#define TTRNN_ROUTE_RUNTIME_MFMA 4    /* runtime-shape two-stage MFMA kernels, any d >= 2 (ttrnn_g2.hip)               */
int ttrnn_rnn_forward_route(const ttrnn_rnn_desc* desc);   /* TTRNN_ROUTE_* or a negative ttrnn_status */
/* Samples one workgroup of the recurrent forward kernel carries (1, or 2: the runtime-shape tier pairs samples where the hidden
 * matrix's merged head is streamed from L2 every step and the batch exceeds the CU count, so that each streamed block feeds two —
 * the naive per-gate sets of tt_linearset.py:5-38 at benchmarking.py's defaults).  Pure host logic; a negative ttrnn_status on a
 * bad descriptor. */
int ttrnn_rnn_forward_samples_per_workgroup(const ttrnn_rnn_desc* desc);

/* Reverse-time part of BPTT (reference: torch autograd through lstm.py:123-133 / gru.py:124-134).
 *   d_out[B][T][H], d_hT/d_cT[B][H] (any may be NULL = zeros)
 *   -> d_gates_in[B][T][G*H], d_gates_hid[B][T][G*H] (fp32: gradients w.r.t. the outputs of the
 *      input / hidden TTLinear; for LSTM both are identical and d_gates_hid may alias d_gates_in
 *      or be NULL), d_h0/d_c0[B][H] (may be NULL).
 * The weight / input gradients then follow from ttrnn_ttlinear_backward over the B*T rows
 * (x rows for in_w; h_{t-1} rows for hid_w).
 *   d_state (may be NULL): fp32 [B][T][H][2] = the TOTAL gradients w.r.t. h_t and c_t of every step (slot 1 is 0 for GRU) —
 *      what the tensor hooks of ActivGradLogger see (rnn_utils.py:127-171, lstm.py:35-39).  Requesting it selects the
 *      runtime-shape / any-shape reverse kernels, which write it from the registers that already hold the values. */
size_t ttrnn_rnn_backward_workspace(const ttrnn_rnn_desc* desc);
/* The same for a caller that knows whether it will pass d_state (want_state != 0): without a d_state request the
 * shape-specialised routes need only their own (small) workspace, not the per-sample plan of the routes that write d_state;
 * ttrnn_rnn_backward_workspace(desc) = the larger of the two answers. */
size_t ttrnn_rnn_backward_workspace_ex(const ttrnn_rnn_desc* desc, int want_state);
/* Kernel family of the reverse-time kernel for this descriptor under the current options (TTRNN_ROUTE_*; want_state != 0: a
 * d_state request).  Pure host logic, as ttrnn_rnn_forward_route. */
int ttrnn_rnn_backward_route(const ttrnn_rnn_desc* desc, int want_state);
int ttrnn_rnn_backward(const ttrnn_rnn_desc* desc, const void* out, const void* h0, const void* c0,
                       const float* packed_hid, const float* reserve,
                       const void* d_out, const void* d_hT, const void* d_cT,
                       float* d_gates_in, float* d_gates_hid, void* d_h0, void* d_c0, float* d_state,
                       void* workspace, size_t workspace_bytes, void* stream);

/* The same, also returning by-products of the reverse-time kernel that the weight-gradient step would otherwise compute by
 * passes of its own over d_gates (cfg2: 66 us of a 2.0 ms training step for the input_size == 1 reduction; column-maximum
 * passes of the two-piece fp16 dense gradient):
 *   stats  fp32 [4][G*H] (overwritten), NULL = none wanted:
 *          rows 0 / 1   max_n |d_gates_in[n][c]| / max_n |d_gates_hid[n][c]|                        TTRNN_BWD_STATS_COLMAX
 *          rows 2 / 3   sum_n x[n] d_gates_in[n][c] / sum_n d_gates_in[n][c]  (input_size == 1, x != NULL; summed per sample in
 *                       the kernel, then over the samples in a fixed order: repeatable bit for bit)  TTRNN_BWD_STATS_IN1SUMS
 *   x      the layer's input [B][T][1] (storage dtype), needed for rows 2 / 3 only.
 * ttrnn_rnn_backward_stats(desc): bitmask of what the route of this descriptor delivers under the current options (the
 * fused-core, merged-big and — where its LDS plan has room — runtime-shape reverse kernels do; 0 elsewhere).  Passing stats != NULL where it returns 0, or together with d_state, is
 * TTRNN_ERR_UNSUPPORTED.  Hand the rows to ttrnn_ttlinear_backward_hinted. */
#define TTRNN_BWD_STATS_COLMAX 1
#define TTRNN_BWD_STATS_IN1SUMS 2
/* ABI 5: where ttrnn_rnn_backward_stats reports this bit, `stats` must hold [4][G*H] + [B*T] floats: behind the four rows the
 * kernel leaves max_c |d_gates_in[n][c]| of every row n = b*T + t (the step's exact maximum) — ttrnn_lin_hints::dy_rowmax. */
#define TTRNN_BWD_STATS_ROWMAX 4
#define TTRNN_BWD_STATS_ROWS 4
int ttrnn_rnn_backward_stats(const ttrnn_rnn_desc* desc);
int ttrnn_rnn_backward_ex(const ttrnn_rnn_desc* desc, const void* out, const void* h0, const void* c0,
                          const float* packed_hid, const float* reserve,
                          const void* d_out, const void* d_hT, const void* d_cT,
                          float* d_gates_in, float* d_gates_hid, void* d_h0, void* d_c0, float* d_state,
                          const void* x, float* stats,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- ABI 7: the weight gradients of one recurrent layer THROUGH THE CHAIN --------------------------------------------------
 * Replaces the autograd of a cell's two TTLinear calls over all B*T rows (reference: tensorized_rnn/lstm.py:23-26 /
 * gru.py:33-36 -> t3nsor/layers.py:121-127 -> t3nsor/ops.py:78-93, differentiated by torch) for the shapes whose chain is the
 * cheaper contraction — 2*in*out > 1.5 x the chain's FLOPs for the hidden matrix: low ranks on large modes, i.e. the
 * reference's own speaker-verification encoder (experiments/speaker_verification/encoder/params_model.py:2-4,14-16: H = 768,
 * n_cores = 2, rank = 2; the dense 768 x 3072 gradient costs 4.8 x the chain there).  Every other descriptor keeps the dense
 * route of ttrnn_ttlinear_backward_hinted, which is cheaper for it (cfg2 / cfg4 / cfg5: 0.2 ... 0.8 x).
 *   mats               bit 0: the input matrix's gradient is wanted, bit 1: the hidden matrix's.  With both bits on an LSTM layer
 *                      the gate gradients are read from HBM ONCE for both matrices.
 *   x [B][T][in], out [B][T][H], h0 [B][H] or NULL (zeros): fp32; the hidden matrix's operand rows are h_{t-1}, read in place.
 *   d_gates_in / d_gates_hid [B][T][G*H] fp32 (ttrnn_rnn_backward_ex; the LSTM's are one buffer).
 *   d_packed_in / d_packed_hid (accumulated into: zero-fill before; NULL where `mats` has no bit), d_bias_in / d_bias_hid
 *                      (accumulated into; NULL = not wanted).
 *   x_colmax [in] / h_colmax [H] / dy_colmax_in / dy_colmax_hid [G*H]: optional UPPER BOUNDS (fp32, as ttrnn_lin_hints) of the
 *                      operands' magnitudes — what missing ones cost is a pass over the operand.
 * ttrnn_rnn_wgrad_workspace: bytes, or 0 = this descriptor (or `mats`) is not taken by the chain kernel under the current options
 * (fp32 storage and split fp32 math only; option dev2 bit 0 switches the route off) — call ttrnn_ttlinear_backward_hinted then.
 * Sums over the rows are added in a fixed order: the gradients are repeatable bit for bit. */
typedef struct ttrnn_wgrad_args {
  const void* x;
  const void* out;
  const void* h0;
  const float* d_gates_in;
  const float* d_gates_hid;
  const float* packed_in;
  const float* packed_hid;
  float* d_packed_in;
  float* d_packed_hid;
  float* d_bias_in;
  float* d_bias_hid;
  const float* x_colmax;
  const float* h_colmax;
  const float* dy_colmax_in;
  const float* dy_colmax_hid;
} ttrnn_wgrad_args;
size_t ttrnn_rnn_wgrad_workspace(const ttrnn_rnn_desc* desc, int mats);
int ttrnn_rnn_wgrad(const ttrnn_rnn_desc* desc, int mats, const ttrnn_wgrad_args* args, void* workspace, size_t workspace_bytes,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTRNN_H_ */
