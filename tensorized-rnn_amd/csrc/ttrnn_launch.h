// ttrnn_launch.h — launch plans shared by the kernel translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include "ttrnn_core.h"

namespace ttrnn {

// Device-side event counters (one small array per device, zeroed when first used): kernels cannot return a status through an
// API that never synchronises, so the rare events a caller must be able to see are counted on the device and read —
// with a synchronisation — only by ttrnn_device_status (include/ttrnn.h).
// (TTRNN_STAT_* indices: include/ttrnn.h)
unsigned* device_status_ptr();        // device pointer to TTRNN_STAT_COUNT counters of the current device (nullptr on failure)


// number of CUs of the current device, queried once per device and process (a hipDeviceGetAttribute per launch showed up
// in the 0.83 ms cfg2 step)
inline int device_cu_count() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 256;
  }
  if (dev < 0 || dev >= 16) dev = 0;
  if (cached[dev] == 0) {
    int v = 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) (void)hipGetLastError();
    cached[dev] = v > 0 ? v : 256;
  }
  return cached[dev];
}

// Raises a kernel's dynamic-LDS limit (needed above 64 KB) once per (device, kernel): hipFuncSetAttribute acts on the
// CURRENT device's copy of the function, so the memo is keyed by the device id as well — a process that drives several GPUs
// (or switches device between calls) gets the limit raised on each of them.  Thread-safe.
int ensure_dynamic_lds(const void* fn, size_t bytes);
// Can all `blocks` workgroups of one launch of `fn` be resident at the same time on an otherwise idle device (occupancy per
// CU x CUs, cached per kernel)?  The two-workgroups-per-sample kernels poll each other and need it; what the query cannot see
// — CUs held by another stream or process at run time — is caught by their bounded wait (NaN + TTRNN_STAT_PAIR_TIMEOUTS).
bool resident_at_once(const void* fn, int block_threads, size_t dyn_lds, long blocks);

// Where the recurrent kernels get the hoisted input projection from.
//   in1 == 0: gin = fp32 [B][T][H][4], one gate-interleaved row per (b, t)
//   in1 == 1: input_size == 1.  W_in x + b is linear in the scalar x: gin holds just TWO rows, chain(1)+b and
//             chain(0)+b (= b), produced by the same chain kernel, and the recurrent kernel forms
//             b + x[b][t] * (row0 - row1) itself from the raw input x (storage dtype) — no [B][T] buffer at all.
struct GinSrc {
  const float* gin;
  const void* x;
  int in1;
  // a stage-wise kernel queued BEHIND a two-piece fp16 kernel as its fallback (ttrnn_fast_f2.hip): runs only when *run_if != 0 —
  // the fp16 kernel's prep found an operand row whose second pieces would be subnormal and that kernel stepped aside — and then
  // counts the launch in status[TTRNN_STAT_GUARD_TRIPS].  NULL (every other caller): run unconditionally.
  const int* run_if = nullptr;
  unsigned* status = nullptr;
};

struct LinPlan {
  int nb;            // rows per tile
  int bs;            // per-row stride of the ping-pong buffers (floats)
  int ss;            // per-row stride of the stage-input stash (backward only)
  int grid;
  bool w_lds;        // packed cores staged in LDS
  bool buf_global;   // chain intermediates in the global workspace (sample too large for LDS)
  bool buf_mixed;    // backward, buf_global: the two ping-pong buffers in LDS all the same, only the stash in the workspace
  bool acc_lds;      // backward: weight/bias gradient accumulators in LDS
  bool acc_slab;     // backward, accumulators too large for LDS: one private global slab per workgroup (summed by a second
                     // kernel) instead of atomics from every row into the one gradient buffer
  bool acc_fixed;    // backward, acc_lds with several workgroups: the LDS accumulators are flushed into per-workgroup slabs (plain
                     // stores) and summed in a fixed order by the second kernel — no atomics, bitwise repeatable
  size_t slab_off;   // byte offset of the slabs inside the workspace
  size_t lds_bytes, ws_bytes;
};

struct RnnPlan {
  int nb, grid;
  bool w_lds, buf_global;
  size_t lds_bytes, ws_bytes;
};

LinPlan plan_ttlinear_fwd(const TtShape& s, int64_t n_rows);
LinPlan plan_ttlinear_bwd(const TtShape& s, int64_t n_rows, bool fixed_order = false);
RnnPlan plan_rnn_generic(const RnnShape& rs, bool backward);

const void* unit_rows_ptr(int dtype);   // device constant {1, 0} in the storage dtype (NULL: symbol lookup failed)
// input_size == 1: dv[o] = sum_n x[n] dy[n][o] (overwritten), db[o] += sum_n dy[n][o]; `part`: in1_reduce_part_bytes(out) bytes of
// scratch for the workgroups' partial sums (summed in a fixed order: repeatable bit for bit, no atomics)
size_t in1_reduce_part_bytes(int out);
int launch_in1_reduce(int dtype, int dy_dtype, int64_t n_rows, int out, const void* x, const void* dy, float* dv,
                      float* db, float* part, hipStream_t stream);
int launch_pack(const TtShape& s, const void* const* cores, const int64_t* strides, int dtype, float* packed,
                hipStream_t stream);
int launch_pack2(const TtShape& sa, const void* const* cores_a, const int64_t* strides_a, float* packed_a,
                 const TtShape& sb, const void* const* cores_b, const int64_t* strides_b, float* packed_b, int dtype,
                 hipStream_t stream);
int launch_unpack(const TtShape& s, const float* packed_grad, void* const* grads, const int64_t* strides, int dtype,
                  hipStream_t stream);
// ilv_h / ilv_mode: optional gate-interleaved output layout, see ttrnn_core.h:ilv_index
int launch_ttlinear_fwd(const TtShape& s, const LinPlan& p, int dtype, int64_t n_rows, const float* packed,
                        const void* bias, const void* x, void* y, void* ws, hipStream_t stream, int ilv_h = 0,
                        int ilv_mode = 0, int epi = 0, float* aux = nullptr);      // epi: TTRNN_EPI_* applied in-kernel (plain rows)
int launch_ttlinear_bwd(const TtShape& s, const LinPlan& p, int dtype, int dy_dtype, int64_t n_rows, const float* packed,
                        const void* x, const void* dy, void* dx, float* d_packed, float* d_bias, void* ws,
                        hipStream_t stream);
int launch_rnn_fwd_generic(const RnnShape& rs, const RnnPlan& p, int dtype, const void* x, const void* h0,
                           const void* c0, const float* packed_in, const void* bias_in, const float* packed_hid,
                           const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                           hipStream_t stream);
int launch_rnn_bwd_generic(const RnnShape& rs, const RnnPlan& p, int dtype, const void* out, const void* h0,
                           const void* c0, const float* packed_hid, const float* reserve, const void* d_out,
                           const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0,
                           void* ws, hipStream_t stream, float* dstate = nullptr);

// shape-specialised MFMA recurrent kernel (ttrnn_fast.hip); gin = hoisted input projection, fp32 [B][T][G*H]
bool fast_rnn_fwd_available(const RnnShape& rs, int dtype);
int launch_rnn_fwd_fast(const RnnShape& rs, int dtype, GinSrc gin, const void* h0, const void* c0,
                        const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT, float* reserve,
                        hipStream_t stream);

// shape-specialised batched TTLinear forward (ttrnn_fast_lin.hip); ilv as launch_ttlinear_fwd.  x / bias have
// storage type `dtype`; y has `dtype` too unless y_f32 (hoisted gate inputs are always fp32).
bool fast_ttlinear_fwd_available(const TtShape& s, int dtype, int ilv_h);
int launch_ttlinear_fwd_fast(const TtShape& s, int dtype, bool y_f32, int64_t n_rows, const float* packed,
                             const void* bias, const void* x, void* y, int ilv_h, int ilv_mode, hipStream_t stream);

// bf16-storage recurrent kernel on the bf16 MFMA (ttrnn_fast_bf16.hip)
bool fast_rnn_fwd_bf16_available(const RnnShape& rs, int dtype);
int launch_rnn_fwd_bf16(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0,
                        const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT, float* reserve,
                        hipStream_t stream);

// fp32 storage, cores 1 and 0 contracted once per launch, fused stage on split-bf16 MFMAs (ttrnn_fast_f10.hip);
// selected by ttrnn_set_fp32_math(TTRNN_MATH_SPLIT), the default
bool f10_rnn_fwd_available(const RnnShape& rs, int dtype);
size_t f10_workspace_bytes(const RnnShape& rs, int dtype);   // fused-core fragments, independent of the math mode
int launch_rnn_fwd_f10(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                       const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                       hipStream_t stream, int phase = 0);

// two-core (d = 2) hidden matrices: both chain stages on two-piece fp16 operands, gates on the accumulators, one barrier per step
// (ttrnn_fast_f2.hip; reached through launch_rnn_fwd_f10 / f10_workspace_bytes / f10_rnn_fwd_available)
bool f2_rnn_fwd_available(const RnnShape& rs, int dtype);
size_t f2_workspace_bytes(const RnnShape& rs, int dtype);
int launch_rnn_fwd_f2(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                      const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws, hipStream_t stream, int phase);
// ... and their reverse-time kernel (two fp16 pieces, per-wave step scales, one barrier per step); ws: f2_rnn_bwd_workspace_bytes
bool f2_rnn_bwd_available(const RnnShape& rs, int dtype);
size_t f2_rnn_bwd_workspace_bytes(const RnnShape& rs, int dtype);
int launch_rnn_bwd_f2(const RnnShape& rs, const void* c0, const float* packed_hid, const float* reserve, const void* d_out,
                      const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, void* ws,
                      hipStream_t stream);

// the bf16 GRU as four-wave workgroups with in-lane gates (ttrnn_fast_f10gq.hip); ws: f10gq_workspace_bytes
size_t f10gq_workspace_bytes();
bool f10gq_available(const RnnShape& rs, int dtype);
int launch_gru_fwd_f10gq(const RnnShape& rs, GinSrc gin, const void* h0, const float* packed_hid, const void* bias_hid,
                         void* out, void* hT, float* reserve, void* ws, hipStream_t stream, int phase);
// ONE launch for pack + unit-row K-in + scale header + fragments of an input_size == 1 fused-core forward (ttrnn_fast_setup.hip)
bool f10_setup_available(const RnnShape& rs, int dtype);
int launch_f10_setup(const RnnShape& rs, const void* const* cores_in, const int64_t* strides_in, const void* bias_in,
                     const void* const* cores_hid, const int64_t* strides_hid, float* packed_in, float* packed_hid, float* gin,
                     void* ws, hipStream_t stream);
// the four-/eight-wave LSTM kernel with ONE barrier per step, S2 inside the gate waves (ttrnn_fast_f10s.hip); ws as launch_rnn_fwd_f10_q
bool f10s_available(const RnnShape& rs, bool with_h0);
int launch_rnn_fwd_f10_s(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid, const void* ws,
                         const float* bias_hid, void* out, void* hT, void* cT, float* reserve, hipStream_t stream);
// the fp32-storage GRU on two fp16 pieces (ttrnn_fast_f10gh.hip; split math mode); ws: f10gh_workspace_bytes
size_t f10gh_workspace_bytes(const RnnShape& rs);
bool f10gh_available(const RnnShape& rs, int dtype);
int launch_gru_fwd_f10gh(const RnnShape& rs, GinSrc gin, const void* h0, const float* packed_hid, const void* bias_hid, void* out,
                         void* hT, float* reserve, void* ws, hipStream_t stream, int phase);
// fp32 TT-GRU H = 512, r = 8 (benchmarking.py defaults with --gru): gates on the accumulators, behind the tier's K-in (ttrnn_fast_f10g5.hip)
// ttrnn_fast_f10n.hip: the naive per-gate TT-LSTM / TT-GRU (hid_blocks = 4 / 3) of H = 256, d = 3, r = 8 on a fused-core kernel, one gate per wave,
// behind the runtime tier's K-in (gin / bilv in its conventions); ws: f10n_workspace_bytes (the tier's `rec` region)
bool f10n_available(const RnnShape& rs, int dtype);
size_t f10n_workspace_bytes(const RnnShape& rs);
int launch_lstm_fwd_f10n(const RnnShape& rs, GinSrc gin, const float* bilv, const void* h0, const void* c0, const float* packed_hid,
                         void* out, void* hT, void* cT, float* reserve, void* ws, hipStream_t stream);
bool f10g5_available(const RnnShape& rs, int dtype);
size_t f10g5_workspace_bytes(const RnnShape& rs);
int launch_gru_fwd_f10g5(const RnnShape& rs, const float* gin, const void* h0, const float* packed_hid, void* out, void* hT,
                         float* reserve, void* ws, hipStream_t stream);
// ... and behind the runtime-shape tier's dense K-in (input_size != 1): gin in the tier's slot convention, ws = the tier's rec region
int launch_gru_fwd_f10gh_g2(const RnnShape& rs, GinSrc src, const float* bilv, const void* h0, const float* packed_hid, void* out,
                            void* hT, float* reserve, void* ws, hipStream_t stream);
bool f10gh_own_plan(const RnnShape& rs);      // the shape has the file's own input_size == 1 plan (r = 8)
// two samples per workgroup (ttrnn_fast_f10nb.hip); wfrag = the fragments launch_rnn_fwd_f10 prepared
int launch_rnn_fwd_f10_nb2(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                           const void* wfrag, const float* bias_hid, void* out, void* hT, void* cT, float* reserve,
                           hipStream_t stream);
int launch_rnn_fwd_f10_q(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                           const void* wfrag, const float* bias_hid, void* out, void* hT, void* cT, float* reserve,
                           hipStream_t stream);

// H = 512, r = 8 (benchmarking.py's default shape): fused-core forward as eight-wave workgroups; gin with both biases folded in
bool f10_h512_fwd_available(const RnnShape& rs, int dtype);
size_t f10_h512_workspace_bytes();
int launch_rnn_fwd_f10_h512(const RnnShape& rs, const float* gin, const void* h0, const void* c0, const float* packed_hid,
                            void* out, void* hT, void* cT, float* reserve, void* ws, hipStream_t stream);

// the same fused-core recurrent kernel on the fp32 MFMA (TTRNN_MATH_EXACT; ttrnn_fast_f10x.hip); ws as above
bool f10x_rnn_fwd_available(const RnnShape& rs, int dtype);
int launch_rnn_fwd_f10x(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                        const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                        hipStream_t stream);

// batched gate-interleaved LSTM input projection through a hidden-shaped TT-matrix on the fused core (K-in of the
// layers above the first); ws: f10_ttlinear_workspace_bytes
bool f10_ttlinear_fwd_available(const TtShape& s, int dtype, int ilv_h, int ilv_mode);
size_t f10_ttlinear_workspace_bytes(const TtShape& s, int dtype, int ilv_h, int ilv_mode);
int launch_ttlinear_fwd_f10(const TtShape& s, int64_t n_rows, const float* packed, const void* bias, const void* x,
                            void* y, void* ws, hipStream_t stream);

// dense split-bf16 GEMM for the batched input projection (ttrnn_fast_gemm.hip): y[n][M] = x[n][K] W[K][M] (+ gate bias);
// `planes` come from launch_gemm_split_prep(W dense fp32 [K][M], produced by a chain kernel on the K unit rows)
bool gemm_split_ok(int K, int M);
size_t gemm_split_identity_bytes(int K);
size_t gemm_split_dense_bytes(int K, int M);
size_t gemm_split_plane_bytes(int K, int M);
int launch_fill_identity(int dtype, int K, void* id, hipStream_t stream);
int launch_gemm_split_prep(const float* WG, int K, int M, void* planes, hipStream_t stream, bool transposed = false);
int launch_gemm_split(int dtype, int64_t n_rows, int K, int M, const void* x, const void* planes, const void* bias,
                      int Hb, float* y, hipStream_t stream, const float* bias_ilv = nullptr);

// the same GEMM on two-piece fp16 operands (three MFMA terms instead of six; per-row scales of x and one scale of W,
// powers of two): forward input projections.  `planes` as above (two of the three planes are used).
size_t gemm_half_scratch_bytes(int64_t n_rows, int K, int M);
// ... and on PRE-SPLIT pieces of x with both operands by LDS-DMA (ttrnn_fast_gemm3.hip): taken by launch_gemm_half itself where
// 256 x 256 tiles fill the chip; xplanes: gemm3_xplane_bytes, part of gemm_half_scratch_bytes
bool gemm3_ok(int64_t n_rows, int K, int M);
size_t gemm3_xplane_bytes(int64_t n_rows, int K);
int launch_gemm3h(int dtype, int64_t n_rows, int K, int M, const void* x, const void* planes, void* scratch, void* xplanes,
                  const void* bias, int Hb, float* y, hipStream_t stream, const float* bias_ilv);
// the two scale passes (row maxima of x, maximum of W) are two more launches: below ~4 G multiply-adds the three-piece
// bf16 GEMM (no passes) is as fast
// bf16 GEMM (no passes) is as fast; option gemm_pieces = 2 / 3 forces either (A/B switch, tests)
bool gemm_use_half(int64_t n_rows, int K, int M);
int launch_gemm_half_prep(const float* WG, int K, int M, void* planes, void* scratch, hipStream_t stream,
                          bool transposed = false);
int launch_gemm_half(int dtype, int64_t n_rows, int K, int M, const void* x, const void* planes, void* scratch,
                     const void* bias, int Hb, float* y, hipStream_t stream, const float* bias_ilv = nullptr,
                     const float* rowmax = nullptr);      // rowmax: upper bounds of the rows' maxima (no pass over x for them)

// dense weight gradient dW[in][out] = x^T dy on the fp32 MFMA (ttrnn_fast_gemm.hip); the TT cores' gradients are linear in it
bool dense_wgrad_ok(int in, int out);
bool dense_wgrad_shift_ok(int64_t n_rows, int64_t shift_T);
// shift_T > 0: x = out[B][shift_T][in] of a recurrent layer and the operand's row n is the previous step's output (row n - 1;
// row n / shift_T of shift_first — zeros if NULL — where n % shift_T == 0): h_{t-1} rows read in place.
// x_colmax / dy_colmax (optional, device): fp32 bit patterns of upper bounds of the columns' maxima (in / out entries) handed
// over by the producer of the operand (ttrnn_ttlinear_backward_hinted): with dy's given the two-piece fp16 variant needs
// no pass over dy and is taken at every size
int launch_dense_wgrad(int dtype, int64_t n_rows, int in, int out, const void* x, const float* dy, float* dW,
                       float* d_bias, hipStream_t stream, bool split, float* scratch, const unsigned* x_colmax = nullptr,
                       const unsigned* dy_colmax = nullptr, int shift_T = 0, const void* shift_first = nullptr);
size_t dense_wgrad_scratch_bytes(int in, int out);

// shapes too large for on-chip residency (ttrnn_fast_big.hip): chain images in an L2-resident workspace
bool big_rnn_fwd_available(const RnnShape& rs, int dtype);
size_t big_rnn_fwd_workspace(const RnnShape& rs);
int launch_rnn_fwd_big(const RnnShape& rs, int dtype, const void* x, const void* h0, const void* c0,
                       const float* packed_in, const void* bias_in, const float* packed_hid, const void* bias_hid,
                       void* out, void* hT, void* cT, float* reserve, void* workspace, hipStream_t stream);

// the pair kernel of the merged two-core matrix on two-piece fp16 operands (ttrnn_fast_bigh.hip); scratch: bigh_workspace_bytes()
size_t bigh_workspace_bytes();
bool bigh_pair_resident(int dtype, int B);       // occupancy query: all 2B workgroups of the pair kernel resident at once?
int launch_lstm_fwd_big2h(const RnnShape& rs, int dtype, const float* gin, const void* h0, const void* c0,
                          const float* m2_hid, const void* bias_in, const void* bias_hid, void* out, void* hT, void* cT,
                          float* reserve, unsigned long long* hxb, void* scratch, hipStream_t stream);

// ... and its reverse-time counterpart (ttrnn_fast_bigbh.hip); fragT: ttrnn_fast_bigb.hip:k_bigb_prep's buffer
size_t bigbh_workspace_bytes();
bool bigbh_pair_resident(int dtype, int B);
const float* bigbh_guard_rows(const void* scratch, int* n_rows);      // the row sums big_guard_tripped reads (ttrnn_big.h)
int launch_lstm_bwd_big2h(const RnnShape& rs, int dtype, const void* c0, const float* fragT, const float* reserve,
                          const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, void* d_h0, void* d_c0,
                          unsigned long long* hxb, void* scratch, hipStream_t stream, unsigned* colmax = nullptr);

// BPTT of the big shape through the merged two-core matrix (ttrnn_fast_bigb.hip): reverse-time kernel (one or two
// workgroups per sample) and the batched TTLinear backward (dx through the transposed merged chain; weight + bias
// gradients accumulated in MFMA registers)
bool big_rnn_bwd_available(const RnnShape& rs, int dtype);
size_t big_rnn_bwd_workspace(const RnnShape& rs);
int launch_rnn_bwd_big(const RnnShape& rs, int dtype, const void* c0, const float* packed_hid, const float* reserve,
                       const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0,
                       void* d_c0, void* ws, hipStream_t stream, float* stats = nullptr);
bool big_ttlinear_bwd_available(const TtShape& s, int dtype, int dy_dtype);
size_t big_ttlinear_bwd_workspace_bytes(const TtShape& s);
int launch_ttlinear_bwd_big(const TtShape& s, int dtype, int64_t n_rows, const float* packed, const void* x,
                            const void* dy, void* dx, float* d_packed, float* d_bias, void* ws, hipStream_t stream,
                            const unsigned* x_colmax = nullptr, const unsigned* dy_colmax = nullptr, int shift_T = 0,
                            const void* shift_first = nullptr);

// runtime-shape two-stage MFMA kernels (ttrnn_g2.hip): any TT-LSTM / TT-GRU layer whose hidden matrix has d >= 2 cores
bool g2_rnn_available(const RnnShape& rs, int dtype);         // forward kernel
bool g2_rnn_fwd_paired(const RnnShape& rs);                   // ... as k_g2_fwd_p: two samples per workgroup share the head stream
bool g2_rnn_bwd_available(const RnnShape& rs, int dtype);     // reverse-time kernel (larger LDS footprint for wide shapes)
size_t g2_rnn_fwd_workspace(const RnnShape& rs);
size_t g2_rnn_bwd_workspace(const RnnShape& rs);
int launch_rnn_fwd_g2(const RnnShape& rs, int dtype, const void* x, const void* h0, const void* c0, const float* packed_in,
                      const void* bias_in, const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT,
                      float* reserve, void* workspace, hipStream_t stream);
int launch_rnn_bwd_g2(const RnnShape& rs, int dtype, const void* out, const void* h0, const void* c0, const float* packed_hid,
                      const float* reserve, const void* d_out, const void* d_hT, const void* d_cT, float* dg_in,
                      float* dg_hid, void* d_h0, void* d_c0, void* ws, hipStream_t stream, float* dstate = nullptr,
                      float* stats = nullptr);
bool g2_rnn_bwd_colmax(const RnnShape& rs);      // TTRNN_BWD_STATS_COLMAX from the runtime-shape reverse-time kernel?

// shape-specialised reverse-time kernel (ttrnn_fast_bwd.hip)
bool fast_rnn_bwd_available(const RnnShape& rs, int dtype);
int launch_rnn_bwd_fast(const RnnShape& rs, int dtype, const void* out, const void* h0, const void* c0,
                        const float* packed_hid, const float* reserve, const void* d_out, const void* d_hT,
                        const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, hipStream_t stream);

// reverse-time TT-LSTM kernel on the fused core, split fp32 math (ttrnn_fast_f10b.hip); ws: fragments built per launch
bool f10_rnn_bwd_available(const RnnShape& rs, int dtype);
size_t f10_rnn_bwd_workspace_bytes(const RnnShape& rs, int dtype);
// By-products of a reverse-time kernel for the weight-gradient step (ttrnn_rnn_backward_ex): `stats` = fp32 [4][G*H] —
// rows 0 / 1: column maxima of d_gates_in / d_gates_hid (bit patterns of non-negative floats: atomicMax), rows 2 / 3 (layers
// with input_size == 1, x != NULL): sum_n x[n] d_gates_in[n][:] and sum_n d_gates_in[n][:], summed per sample in the kernel
// (`part` [B][2][G*H] in the workspace) and over the samples in a fixed order by launch_bwd_stats_finish.
struct BwdStats {
  const void* x = nullptr;       // [B][T] (storage dtype) or NULL
  float* part = nullptr;         // [B][2][G*H] or NULL
  unsigned* colmax = nullptr;    // = stats rows 0, 1 (zeroed by the launcher) or NULL
  float* rowmax = nullptr;       // [B*T]: the step's exact max |dg| (TTRNN_BWD_STATS_ROWMAX; two-piece LSTM kernel) or NULL
};
int launch_bwd_stats_finish(int cell, int Bn, int GH, const float* part, float* stats, hipStream_t stream);
size_t bwd_stats_part_bytes(const RnnShape& rs);
int launch_rnn_bwd_f10(const RnnShape& rs, int dtype, const void* out, const void* h0, const void* c0,
                       const float* packed_hid, const float* reserve, const void* d_out, const void* d_hT,
                       const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, void* ws,
                       hipStream_t stream, const void* x_in1 = nullptr, float* stats = nullptr);

// the same on two fp16 pieces per operand (ttrnn_fast_f10bh.hip): the TT-LSTM shapes in split mode (option gemm_pieces = 3:
// the three-bf16-piece kernel)
bool f10bh_available(const RnnShape& rs, int dtype);
// ... and for the runtime tier's H = 512, r = 8 TT-LSTM (the reference's benchmarking.py defaults): called from that tier's branch
// of ttrnn_rnn_backward_ex when no per-step state gradients are asked for; stats: the column maxima (rows 0 / 1) or NULL
bool f10bh_h512_available(const RnnShape& rs, int dtype);
// naive per-gate sets of H = 256 (ttrnn_fast_f10n.hip: forward, unjoin; ttrnn_fast_f10bh.hip: the reverse-time kernel, round 6)
size_t f10n_unjoined_floats(int cell);
int launch_f10n_unjoin(const RnnShape& rs, const float* packed_hid, float* pg, hipStream_t stream);
bool f10n_shape_matches(const RnnShape& rs);
bool f10n_bwd_available(const RnnShape& rs, int dtype);
size_t f10n_bwd_workspace_bytes(const RnnShape& rs);
int launch_rnn_bwd_f10n(const RnnShape& rs, const void* out, const void* h0, const void* c0, const float* packed_hid,
                        const float* reserve, const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid,
                        void* d_h0, void* d_c0, void* ws, hipStream_t stream, float* stats);
size_t f10bh_h512_workspace_bytes();
int launch_rnn_bwd_f10_h512(const RnnShape& rs, const void* out, const void* h0, const void* c0, const float* packed_hid,
                            const float* reserve, const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid,
                            void* d_h0, void* d_c0, void* ws, hipStream_t stream, float* stats);
// ttrnn_fast_proj.hip: d_packed += the adjoint of (three cores -> dense matrix) applied to dW (fp32 [in][out]); ws:
// proj3_workspace_bytes (0: not offered for this shape / switched off by option dev bit 10)
size_t proj3_workspace_bytes(const TtShape& s);
int launch_proj3(const TtShape& s, const float* packed, const float* dW, float* d_packed, void* ws, hipStream_t stream);
size_t f10bh_workspace_bytes(const RnnShape& rs);
int launch_lstm_bwd_f10h(const RnnShape& rs, const void* c0, const float* packed_hid, const float* reserve,
                         const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0,
                         void* d_c0, void* ws, unsigned long long* diag, hipStream_t stream, const BwdStats& bs);

int launch_gru_bwd_f10h(const RnnShape& rs, int dtype, const void* out, const void* h0, const float* packed_hid,
                        const float* reserve, const void* d_out, const void* d_hT, float* dg_in, float* dg_hid, void* d_h0,
                        void* ws, hipStream_t stream, const BwdStats& bs);

size_t f10b_fragment_bytes(const TtShape& s);     // the transposed fused-core fragments alone
int launch_f10b_prep(const TtShape& s, const float* packed, void* wfrag, hipStream_t stream, float* zero = nullptr,
                     int zero_n = 0);      // zero: a caller's accumulator cleared by the same launch

// batched weight (+ bias) gradients of a hidden-shaped TT-matrix through the fused core, dx optional (ttrnn_fast_f10w.hip)
bool f10_ttlinear_wgrad_available(const TtShape& s, int dtype, int dy_dtype);
size_t f10_ttlinear_wgrad_workspace_bytes(const TtShape& s);
bool f10_ttlinear_wgrad_has_dx(const TtShape& s);
int launch_ttlinear_wgrad_f10(const TtShape& s, int dtype, int64_t n_rows, const float* packed, const void* x,
                              const void* dy, void* dx, float* d_packed, float* d_bias, void* ws, hipStream_t stream);

// shape-specialised batched TTLinear backward (ttrnn_fast_bwd.hip); accumulates into d_packed / d_bias
bool fast_ttlinear_bwd_available(const TtShape& s, int dtype, int dy_dtype);
int launch_ttlinear_bwd_fast(const TtShape& s, int dtype, int dy_dtype, int64_t n_rows, const float* packed,
                             const void* x, const void* dy, void* dx, float* d_packed, float* d_bias,
                             hipStream_t stream);

// TT-LSTM forward of the speaker encoder's shape (H = 768, two cores, rank 2, 40 inputs): both chain stages and the input
// projection inside one persistent kernel, no hoisted K-in (ttrnn_fast_w2.hip)
bool w2_rnn_fwd_available(const RnnShape& rs, int dtype);
size_t w2_rnn_fwd_workspace_bytes();
int launch_rnn_fwd_w2(const RnnShape& rs, const void* x, const void* h0, const void* c0, const float* packed_in, const void* bias_in,
                      const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                      hipStream_t stream, int phase = 0 /* TTRNN_PHASE_ALL */);

bool w2_rnn_bwd_available(const RnnShape& rs, int dtype);
size_t w2_rnn_bwd_workspace_bytes();
int launch_rnn_bwd_w2(const RnnShape& rs, const void* out, const void* h0, const void* c0, const float* packed_hid, const float* reserve,
                      const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, void* ws,
                      hipStream_t stream, float* stats = nullptr);

// chain weight gradients of up to two TT-matrices sharing one pass over dy (ttrnn_fast_c2w.hip, plan: ttrnn_c2w.h)
bool c2w_prefers_chain(const TtShape& s);      // 2 in out > 1.5 x the chain's FLOPs
size_t c2w_workspace_bytes(const TtShape* const* shapes, int nmat, bool small_only = false);      // 0: the kernel does not take these shapes
int launch_c2w(const TtShape* const* shapes, int nmat, int64_t n_rows, const float* const* packed, const float* const* x,
               const float* const* first, const int* T, const float* dy, const unsigned* const* x_cmax, const int* x_cn,
               const unsigned* dy_cmax, float* const* d_packed, float* d_bias0, float* d_bias1, void* workspace,
               size_t workspace_bytes, hipStream_t stream);

}  // namespace ttrnn
