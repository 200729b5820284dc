// ttrnn_api.hip — the extern "C" surface of libttrnn.so (see include/ttrnn.h for the contract).
// Validates descriptors, picks a kernel (shape-specialised MFMA kernel when one exists for the
// descriptor, else the any-shape kernel), and launches on the caller's stream.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <utility>
#include <vector>
#include "ttrnn.h"
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"

using namespace ttrnn;

// ---- options: one table, filled once from TTRNN_* environment variables, atomics afterwards (ttrnn_opts.h) ---------
namespace ttrnn {
namespace {
struct OptTable {
  std::atomic<int> v[OPT_COUNT];
  OptTable() {
    static const char* const env[OPT_COUNT] = {
        "TTRNN_FP32_MATH", "TTRNN_FORCE_GENERIC", "TTRNN_NO_GEMM", "TTRNN_NO_IN1", "TTRNN_NO_F10", "TTRNN_NO_G2",
        "TTRNN_FORCE_G2",
        "TTRNN_DIAG", "TTRNN_BF16_FP32_MFMA", "TTRNN_BIG_MERGE", "TTRNN_BIG_NO_GEMM", "TTRNN_BIG_NO_PAIR",
        "TTRNN_NO_BIGB", "TTRNN_BIGW_SLICES", "TTRNN_F10_NB1", "TTRNN_DENSE_FP32", "TTRNN_F10_NB2", "TTRNN_GEMM_PIECES",
        "TTRNN_BIG_FP32_MFMA", "TTRNN_PAIR_FAULT", "TTRNN_NO_GEMM3", "TTRNN_DEV", "TTRNN_DEV2"};
    for (int i = 0; i < OPT_COUNT; ++i) {
      const char* e = getenv(env[i]);
      int val = 0;
      if (i == OPT_FP32_MATH) val = (e && (e[0] == 'e' || e[0] == '0')) ? TTRNN_MATH_EXACT : TTRNN_MATH_SPLIT;
      else if (i == OPT_BIG_MERGE) val = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2;
      else if (i == OPT_GEMM_PIECES) val = (e && (e[0] == '2' || e[0] == '3')) ? e[0] - '0' : 0;
      else if (i == OPT_DEV || i == OPT_DEV2) val = e ? (atoi(e) & 0x3FFFFFFF) : 0;
      else val = (e && e[0] == '1') ? 1 : 0;
      v[i].store(val, std::memory_order_relaxed);
    }
  }
};
OptTable& table() {
  static OptTable t;      // thread-safe initialisation (C++11 magic static)
  return t;
}
const char* const kOptNames[OPT_COUNT] = {
    "fp32_math", "force_generic", "no_gemm", "no_in1", "no_f10", "no_g2", "force_g2", "diag", "bf16_fp32_mfma", "big_merge",
    "big_no_gemm", "big_no_pair", "no_bigb", "bigw_slices", "f10_nb1", "dense_fp32", "f10_nb2", "gemm_pieces", "big_fp32_mfma",
    "pair_fault", "no_gemm3", "dev", "dev2"};
}  // namespace
int opt(OptId id) { return table().v[id].load(std::memory_order_relaxed); }
const char* opt_name(OptId id) { return kOptNames[id]; }
static int opt_find(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOptNames[i]) == 0) return i;
  return -1;
}
int opt_set(const char* name, int value) {
  const int i = opt_find(name);
  if (i < 0) return -1;
  if (i == OPT_FP32_MATH && value != TTRNN_MATH_EXACT && value != TTRNN_MATH_SPLIT) return -1;
  if (i == OPT_BIG_MERGE && (value < 0 || value > 2)) return -1;
  if (i == OPT_GEMM_PIECES && value != 0 && value != 2 && value != 3) return -1;
  if ((i == OPT_DEV || i == OPT_DEV2) && (value < 0 || value > 0x3FFFFFFF)) return -1;
  if (i != OPT_FP32_MATH && i != OPT_BIG_MERGE && i != OPT_GEMM_PIECES && i != OPT_DEV && i != OPT_DEV2 && value != 0 && value != 1) return -1;
  table().v[i].store(value, std::memory_order_relaxed);
  return 0;
}
int opt_get(const char* name, int* value) {
  const int i = opt_find(name);
  if (i < 0 || !value) return -1;
  *value = table().v[i].load(std::memory_order_relaxed);
  return 0;
}
}  // namespace ttrnn

namespace ttrnn {
// event counters of ttrnn_device_status: a zero-initialised device global per device (no allocation, no launch)
__device__ unsigned g_ttrnn_status[TTRNN_STAT_COUNT];
unsigned* device_status_ptr() {
  static std::mutex mu;
  static unsigned* cached[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lock(mu);
  if (!cached[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_ttrnn_status)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    cached[dev] = (unsigned*)p;
  }
  return cached[dev];
}

int ensure_dynamic_lds(const void* fn, size_t bytes) {
  // (small threshold only: a kernel's static __shared__ arrays count against the default 64 KB as well; the limit is set
  // to exactly what the launch asks for — a blanket 160 KB is refused for kernels that also have static LDS)
  if (bytes <= 32 * 1024) return TTRNN_OK;
  struct Entry { int dev; const void* fn; size_t bytes; };
  static std::mutex mu;
  static std::vector<Entry> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TTRNN_ERR_LAUNCH; }
  std::lock_guard<std::mutex> lock(mu);
  Entry* hit = nullptr;
  for (auto& e : done)
    if (e.dev == dev && e.fn == fn) { hit = &e; break; }
  if (hit && hit->bytes >= bytes) return TTRNN_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    return TTRNN_ERR_LAUNCH;
  }
  if (hit) hit->bytes = bytes;
  else done.push_back(Entry{dev, fn, bytes});
  return TTRNN_OK;
}
bool resident_at_once(const void* fn, int block_threads, size_t dyn_lds, long blocks) {
  struct Entry { int dev; const void* fn; size_t lds; int per_cu; };
  static std::mutex mu;
  static std::vector<Entry> seen;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  int per_cu = -1;
  {
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : seen)
      if (e.dev == dev && e.fn == fn && e.lds == dyn_lds) { per_cu = e.per_cu; break; }
    if (per_cu < 0) {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, block_threads, dyn_lds) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
      }
      per_cu = n;
      seen.push_back(Entry{dev, fn, dyn_lds, per_cu});
    }
  }
  return (long)per_cu * device_cu_count() >= blocks;
}
}  // namespace ttrnn

static bool force_generic() { return opt(OPT_FORCE_GENERIC) != 0; }
static bool no_gemm() { return opt(OPT_NO_GEMM) != 0; }
static int fp32_math() { return opt(OPT_FP32_MATH); }

extern "C" {

int ttrnn_set_fp32_math(int mode) { return opt_set("fp32_math", mode) == 0 ? TTRNN_OK : TTRNN_ERR_UNSUPPORTED; }

int ttrnn_set_option(const char* name, int value) { return opt_set(name, value) == 0 ? TTRNN_OK : TTRNN_ERR_UNSUPPORTED; }
int ttrnn_get_option(const char* name, int* value) { return opt_get(name, value) == 0 ? TTRNN_OK : TTRNN_ERR_UNSUPPORTED; }

int ttrnn_get_fp32_math(void) { return fp32_math(); }

int ttrnn_abi_version(void) { return TTRNN_ABI_VERSION; }

const char* ttrnn_status_string(int status) {
  switch (status) {
    case TTRNN_OK: return "ok";
    case TTRNN_ERR_BAD_DESC: return "bad descriptor (modes / ranks / sizes inconsistent)";
    case TTRNN_ERR_NULL: return "required pointer is NULL";
    case TTRNN_ERR_UNSUPPORTED: return "unsupported configuration";
    case TTRNN_ERR_WORKSPACE: return "workspace too small";
    case TTRNN_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown status";
  }
}

int ttrnn_device_status(unsigned int* counters, int n, int reset) {
  if (!counters || n < 1) return TTRNN_ERR_NULL;
  unsigned* p = device_status_ptr();
  if (!p) return TTRNN_ERR_LAUNCH;
  unsigned host[TTRNN_STAT_COUNT] = {0};
  if (hipMemcpy(host, p, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return TTRNN_ERR_LAUNCH; }
  for (int i = 0; i < n; ++i) counters[i] = i < TTRNN_STAT_COUNT ? host[i] : 0u;
  if (reset && hipMemset(p, 0, sizeof(host)) != hipSuccess) { (void)hipGetLastError(); return TTRNN_ERR_LAUNCH; }
  return TTRNN_OK;
}

int ttrnn_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n > 0 ? 1 : 0;
}

int64_t ttrnn_packed_elems(const ttrnn_ttm* w) {
  TtShape s;
  if (tt_shape_init(&s, w) != TTRNN_OK) return -1;
  return 2 * (int64_t)s.wtotal;
}

int ttrnn_pack_cores(const ttrnn_ttm* w, const void* const* cores, const int64_t* strides, int dtype, float* packed,
                     void* stream) {
  TtShape s;
  int st = tt_shape_init(&s, w);
  if (st != TTRNN_OK) return st;
  if (!cores || !strides || !packed) return TTRNN_ERR_NULL;
  for (int k = 0; k < s.d; ++k) if (!cores[k]) return TTRNN_ERR_NULL;
  if (dtype != TTRNN_F32 && dtype != TTRNN_BF16) return TTRNN_ERR_UNSUPPORTED;
  return launch_pack(s, cores, strides, dtype, packed, (hipStream_t)stream);
}

int ttrnn_pack_cores2(const ttrnn_ttm* wa, const void* const* cores_a, const int64_t* strides_a, float* packed_a,
                      const ttrnn_ttm* wb, const void* const* cores_b, const int64_t* strides_b, float* packed_b,
                      int dtype, void* stream) {
  TtShape sa, sb;
  int st = tt_shape_init(&sa, wa);
  if (st != TTRNN_OK) return st;
  st = tt_shape_init(&sb, wb);
  if (st != TTRNN_OK) return st;
  if (!cores_a || !strides_a || !packed_a || !cores_b || !strides_b || !packed_b) return TTRNN_ERR_NULL;
  for (int k = 0; k < sa.d; ++k) if (!cores_a[k]) return TTRNN_ERR_NULL;
  for (int k = 0; k < sb.d; ++k) if (!cores_b[k]) return TTRNN_ERR_NULL;
  if (dtype != TTRNN_F32 && dtype != TTRNN_BF16) return TTRNN_ERR_UNSUPPORTED;
  return launch_pack2(sa, cores_a, strides_a, packed_a, sb, cores_b, strides_b, packed_b, dtype, (hipStream_t)stream);
}

int ttrnn_unpack_core_grads(const ttrnn_ttm* w, const float* packed_grad, void* const* core_grads,
                            const int64_t* strides, int dtype, void* stream) {
  TtShape s;
  int st = tt_shape_init(&s, w);
  if (st != TTRNN_OK) return st;
  if (!packed_grad || !core_grads || !strides) return TTRNN_ERR_NULL;
  for (int k = 0; k < s.d; ++k) if (!core_grads[k]) return TTRNN_ERR_NULL;
  if (dtype != TTRNN_F32 && dtype != TTRNN_BF16) return TTRNN_ERR_UNSUPPORTED;
  return launch_unpack(s, packed_grad, core_grads, strides, dtype, (hipStream_t)stream);
}

// ---- TTLinear ---------------------------------------------------------------------------------
// input_size == 1 backward: dv (fp32[out]) + the unit input row live in the workspace
// input_size == 1 backward: [dv: out floats | the reduction's partial sums (launch_in1_reduce)]
static size_t in1_dv_bytes(const TtShape& s) { return ((size_t)s.out_size * sizeof(float) + 255 + 256) & ~(size_t)255; }
static size_t in1_bwd_bytes(const TtShape& s) { return in1_dv_bytes(s) + ((in1_reduce_part_bytes(s.out_size) + 255) & ~(size_t)255); }

// Dense-gradient backward (ttrnn_fast_gemm.hip) of a shape with a fused-core weight-gradient kernel: workspace =
// [that kernel's own | identity rows in x in | dW in x out | dense W in x out | bf16 planes of W^T (K = out, M = in) |
//  partial dW tiles of the row ranges]
static size_t dense_bwd_f10w(const TtShape& s) { return (f10_ttlinear_wgrad_workspace_bytes(s) + 255) & ~(size_t)255; }
static size_t dense_bwd_bytes(const TtShape& s) {
  if (f10_ttlinear_wgrad_workspace_bytes(s) == 0 || !dense_wgrad_ok(s.in_size, s.out_size)) return 0;
  return dense_bwd_f10w(s) + gemm_split_identity_bytes(s.in_size) + 2 * gemm_split_dense_bytes(s.in_size, s.out_size) +
         gemm_split_plane_bytes(s.out_size, s.in_size) + dense_wgrad_scratch_bytes(s.in_size, s.out_size) +
         proj3_workspace_bytes(s);      // (last: the pull-back of dW to the cores, ttrnn_fast_proj.hip)
}

// The same dense-gradient backward for shapes WITHOUT a specialised kernel (the callers' side of ttrnn_g2.hip): dW = x^T dy
// as one dense GEMM, pulled back to the cores by the any-shape backward kernel on the `in` identity rows (the adjoint of
// "cores -> dense matrix" is linear in dW); dx = dy W^T as a split-bf16 GEMM with W = the any-shape chain on the identity
// rows.  Workspace = [any-shape backward (in rows) | identity | dW | W | planes of W^T | row-range partials | any-shape forward]
struct GenDense {
  bool ok, dx_ok;
  size_t lin_bwd, ident, dwd, wd, planes, scratch, lin_fwd, proj, total;
};
static GenDense gen_dense(const TtShape& s) {
  GenDense g{};
  g.ok = s.in_size >= 4 && dense_wgrad_ok(s.in_size, s.out_size);
  if (!g.ok) return g;
  g.dx_ok = gemm_split_ok(s.out_size, s.in_size);
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  g.lin_bwd = al(plan_ttlinear_bwd(s, s.in_size, true).ws_bytes);
  g.ident = gemm_split_identity_bytes(s.in_size);
  g.dwd = gemm_split_dense_bytes(s.in_size, s.out_size);
  g.wd = g.dx_ok ? gemm_split_dense_bytes(s.in_size, s.out_size) : 0;
  g.planes = g.dx_ok ? gemm_split_plane_bytes(s.out_size, s.in_size) : 0;
  g.scratch = dense_wgrad_scratch_bytes(s.in_size, s.out_size);
  g.lin_fwd = g.dx_ok ? al(plan_ttlinear_fwd(s, s.in_size).ws_bytes) : 0;
  g.proj = proj3_workspace_bytes(s);               // d = 3: the pull-back as three small launches (ttrnn_fast_proj.hip)
  g.total = g.lin_bwd + g.ident + g.dwd + g.wd + g.planes + g.scratch + g.lin_fwd + g.proj;
  return g;
}
// input_size == 1 without a specialised kernel: dv + the any-shape backward's own workspace for ONE row
static size_t gen_in1_bytes(const TtShape& s) {
  return in1_bwd_bytes(s) + ((plan_ttlinear_bwd(s, 1).ws_bytes + 255) & ~(size_t)255);
}

// dx = dy W^T on two fp16 pieces (hints->dy_rowmax given): the GEMM's column scales of W^T and row scales of dy, behind the
// dense-gradient layout of the fused-core shapes (0: that variant is not offered for this call)
static size_t dense_dx_half_bytes(const TtShape& s, int64_t n_rows) {
  if (dense_bwd_bytes(s) == 0 || !gemm_split_ok(s.out_size, s.in_size) || !gemm_use_half(n_rows, s.out_size, s.in_size) ||
      gemm3_ok(n_rows, s.out_size, s.in_size))
    return 0;
  return gemm_half_scratch_bytes(n_rows, s.out_size, s.in_size);
}

size_t ttrnn_ttlinear_workspace(const ttrnn_ttm* w, int64_t n_rows) {
  TtShape s;
  if (tt_shape_init(&s, w) != TTRNN_OK || n_rows < 0) return 0;
  const LinPlan f = plan_ttlinear_fwd(s, n_rows);
  const LinPlan b = plan_ttlinear_bwd(s, n_rows);
  size_t ws = f.ws_bytes > b.ws_bytes ? f.ws_bytes : b.ws_bytes;
  if (s.in_size == 1 && in1_bwd_bytes(s) > ws) ws = in1_bwd_bytes(s);
  const size_t f10w = f10_ttlinear_wgrad_workspace_bytes(s);      // fused-core weight gradients (any math mode)
  if (f10w > ws) ws = f10w;
  const size_t dnb = dense_bwd_bytes(s) + dense_dx_half_bytes(s, n_rows);      // dense-gradient backward of the fused-core shapes
  if (dnb > ws) ws = dnb;
  const size_t bigw = big_ttlinear_bwd_workspace_bytes(s);        // merged-core backward of the big shape
  if (bigw > ws) ws = bigw;
  const GenDense gd = gen_dense(s);                               // dense-gradient backward of every other shape
  if (gd.ok && gd.total > ws) ws = gd.total;
  if (s.in_size == 1 && gen_in1_bytes(s) > ws) ws = gen_in1_bytes(s);
  return ws;
}

int ttrnn_ttlinear_forward(const ttrnn_ttm* w, int dtype, int64_t n_rows, const float* packed, const void* bias,
                           const void* x, void* y, void* workspace, size_t workspace_bytes, void* stream) {
  TtShape s;
  int st = tt_shape_init(&s, w);
  if (st != TTRNN_OK) return st;
  if (n_rows < 0) return TTRNN_ERR_BAD_DESC;
  if (dtype != TTRNN_F32 && dtype != TTRNN_BF16) return TTRNN_ERR_UNSUPPORTED;
  if (n_rows == 0) return TTRNN_OK;
  if (!packed || !x || !y) return TTRNN_ERR_NULL;
  if (!force_generic() && fast_ttlinear_fwd_available(s, dtype, 0))
    return launch_ttlinear_fwd_fast(s, dtype, false, n_rows, packed, bias, x, y, 0, 0, (hipStream_t)stream);
  const LinPlan p = plan_ttlinear_fwd(s, n_rows);
  if (p.ws_bytes > 0 && (!workspace || workspace_bytes < p.ws_bytes)) return TTRNN_ERR_WORKSPACE;
  return launch_ttlinear_fwd(s, p, dtype, n_rows, packed, bias, x, y, workspace, (hipStream_t)stream);
}

int ttrnn_ttlinear_backward(const ttrnn_ttm* w, int dtype, int dy_dtype, int64_t n_rows, const float* packed, const void* x,
                            const void* dy, void* dx, float* d_packed, float* d_bias, void* workspace,
                            size_t workspace_bytes, void* stream) {
  return ttrnn_ttlinear_backward_hinted(w, dtype, dy_dtype, n_rows, packed, x, dy, dx, d_packed, d_bias, nullptr, workspace,
                                        workspace_bytes, stream);
}

}  // extern "C"

// probe: launch nothing, return 1 / 0 = the route this call would take reads x through hints->x_period / not
static int lin_backward(const ttrnn_ttm* w, int dtype, int dy_dtype, int64_t n_rows, const float* packed, const void* x,
                        const void* dy, void* dx, float* d_packed, float* d_bias, const ttrnn_lin_hints* hints,
                        void* workspace, size_t workspace_bytes, void* stream, bool probe) {
  const int shT = hints && hints->x_period > 0 && d_packed ? (int)hints->x_period : 0;
  const void* shF = shT > 0 ? hints->x_first : nullptr;
  if (hints && hints->x_period > 0 && (hints->x_period >= ((int64_t)1 << 31) || n_rows % hints->x_period != 0))
    return probe ? 0 : TTRNN_ERR_BAD_DESC;
  if (shT > 0 && !probe && !dense_wgrad_shift_ok(n_rows, shT)) return TTRNN_ERR_UNSUPPORTED;
// a route that reads x row by row itself: no shifted rows there
#define TT_ROUTE_PLAIN_ROWS()                          \
  do {                                                 \
    if (probe) return 0;                               \
    if (shT > 0) return TTRNN_ERR_UNSUPPORTED;         \
  } while (0)
  const unsigned* hx = hints ? reinterpret_cast<const unsigned*>(hints->x_colmax) : nullptr;
  const unsigned* hdy = hints ? reinterpret_cast<const unsigned*>(hints->dy_colmax) : nullptr;
  const float* hsum = hints ? hints->xdy_sum : nullptr;
  TtShape s;
  int st = tt_shape_init(&s, w);
  if (st != TTRNN_OK) return st;
  if (n_rows < 0) return TTRNN_ERR_BAD_DESC;
  if (dtype != TTRNN_F32 && dtype != TTRNN_BF16) return TTRNN_ERR_UNSUPPORTED;
  if (dy_dtype != TTRNN_F32 && dy_dtype != TTRNN_BF16) return TTRNN_ERR_UNSUPPORTED;
  if (n_rows == 0) return probe ? 0 : TTRNN_OK;
  if (!packed || !dy) return probe ? 0 : TTRNN_ERR_NULL;
  if (d_packed && !x) return probe ? 0 : TTRNN_ERR_NULL;
  if (!dx && !d_packed && !d_bias) return probe ? 0 : TTRNN_OK;
  if (!force_generic() && fast_ttlinear_bwd_available(s, dtype, dy_dtype)) {
    if (s.in_size == 1 && !dx && d_packed && workspace && workspace_bytes >= in1_bwd_bytes(s) && !opt(OPT_NO_IN1)) {
      // y_n = b + x_n * chain(1): reduce dy over the rows once (dv = sum x_n dy_n, d_bias = sum dy_n), then
      // back-propagate dv through the chain on the single unit row
      TT_ROUTE_PLAIN_ROWS();
      float* dv = (float*)workspace;
      const void* unit = unit_rows_ptr(dtype);
      if (!unit) return TTRNN_ERR_LAUNCH;
      if (hsum && !d_bias) {
        // the producer of dy has the row sums already (ttrnn_rnn_backward_ex, TTRNN_BWD_STATS_IN1SUMS): no pass over dy
        // ONE row: the any-shape kernel (a workgroup, three short stages) is quicker than the batched MFMA kernel's launch
        // built for row tiles (cfg2: 28 -> ~13 us)
        if (workspace_bytes >= gen_in1_bytes(s)) {
          const void* unit32 = unit_rows_ptr(TTRNN_F32);
          const LinPlan p1 = plan_ttlinear_bwd(s, 1);
          return launch_ttlinear_bwd(s, p1, TTRNN_F32, TTRNN_F32, 1, packed, unit32, hsum, nullptr, d_packed, nullptr,
                                     (char*)workspace + in1_bwd_bytes(s), (hipStream_t)stream);
        }
        return launch_ttlinear_bwd_fast(s, dtype, TTRNN_F32, 1, packed, unit, hsum, nullptr, d_packed, nullptr,
                                        (hipStream_t)stream);
      }
      st = launch_in1_reduce(dtype, dy_dtype, n_rows, s.out_size, x, dy, dv, d_bias, (float*)((char*)workspace + in1_dv_bytes(s)),
                             (hipStream_t)stream);
      if (st != TTRNN_OK) return st;
      return launch_ttlinear_bwd_fast(s, dtype, TTRNN_F32, 1, packed, unit, dv, nullptr, d_packed, nullptr,
                                      (hipStream_t)stream);
    }
    if ((fp32_math() == TTRNN_MATH_SPLIT || dtype == TTRNN_BF16) && d_packed && !no_gemm() &&
        f10_ttlinear_wgrad_available(s, dtype, dy_dtype) && dense_bwd_bytes(s) > 0 && n_rows >= 4 * (int64_t)s.in_size &&
        (!dx || (dtype == TTRNN_F32 && fp32_math() == TTRNN_MATH_SPLIT && gemm_split_ok(s.out_size, s.in_size) &&
                 fast_ttlinear_fwd_available(s, dtype, 0))) &&
        workspace && workspace_bytes >= dense_bwd_bytes(s)) {
      // Every row is independent, so the TT structure buys nothing here: one dense GEMM dW = x^T dy (fp32 MFMA), whose
      // image under the adjoint of "cores -> dense matrix" is what the fused-core weight-gradient kernel computes when it
      // is fed the `in` unit rows as x and dW's rows as dy; dx = dy W^T is a second dense GEMM (split-bf16).
      if (probe) return dense_wgrad_shift_ok(n_rows, hints->x_period) ? 1 : 0;
      hipStream_t sm = (hipStream_t)stream;
      char* wsb = (char*)workspace;
      void* ident = wsb + dense_bwd_f10w(s);
      float* dWd = (float*)((char*)ident + gemm_split_identity_bytes(s.in_size));
      float* Wd = (float*)((char*)dWd + gemm_split_dense_bytes(s.in_size, s.out_size));
      void* planes = (char*)Wd + gemm_split_dense_bytes(s.in_size, s.out_size);
      // (the unit rows feed the dense W of the dx GEMM — and the pull-back only where ttrnn_fast_proj.hip does not do it)
      const size_t pj = proj3_workspace_bytes(s);
      if (dx || pj == 0) st = launch_fill_identity(dtype, s.in_size, ident, sm);
      if (st == TTRNN_OK) st = launch_dense_wgrad(dtype, n_rows, s.in_size, s.out_size, x, (const float*)dy, dWd, d_bias, sm,
                                                 fp32_math() == TTRNN_MATH_SPLIT || dtype == TTRNN_BF16,
                                                 (float*)((char*)planes + gemm_split_plane_bytes(s.out_size, s.in_size)), hx,
                                                 hdy, shT, shF);
      // pull dW back to the cores: three small fp32 launches (ttrnn_fast_proj.hip) instead of the fused-core weight-gradient
      // kernel on the unit rows (r = 16: 60 + 24 us -> ~15)
      if (st == TTRNN_OK) {
        st = pj > 0 ? launch_proj3(s, packed, dWd, d_packed, wsb + dense_bwd_bytes(s) - pj, sm)
                    : launch_ttlinear_wgrad_f10(s, dtype, s.in_size, packed, ident, dWd, nullptr, d_packed, nullptr, workspace, sm);
      }
      if (st != TTRNN_OK || !dx) return st;
      st = launch_ttlinear_fwd_fast(s, dtype, true, s.in_size, packed, nullptr, ident, Wd, 0, 0, sm);     // W[j][o]
      // two fp16 pieces (three MFMA terms instead of six) where the caller hands over the row maxima of dy — the reverse-time
      // kernel's by-product — and the workspace has room for the scales (cfg4's two stacked layers: 257 -> ~150 us each)
      const size_t dxh = hints && hints->dy_rowmax ? dense_dx_half_bytes(s, n_rows) : 0;
      if (dxh > 0 && workspace_bytes >= dense_bwd_bytes(s) + dxh) {
        void* gscr = wsb + dense_bwd_bytes(s);
        if (st == TTRNN_OK) st = launch_gemm_half_prep(Wd, s.out_size, s.in_size, planes, gscr, sm, true);
        if (st == TTRNN_OK)
          st = launch_gemm_half(TTRNN_F32, n_rows, s.out_size, s.in_size, dy, planes, gscr, nullptr, 0, (float*)dx, sm, nullptr,
                                hints->dy_rowmax);
        return st;
      }
      if (st == TTRNN_OK) st = launch_gemm_split_prep(Wd, s.out_size, s.in_size, planes, sm, true);
      if (st == TTRNN_OK)
        st = launch_gemm_split(TTRNN_F32, n_rows, s.out_size, s.in_size, dy, planes, nullptr, 0, (float*)dx, sm);
      return st;
    }
    if ((fp32_math() == TTRNN_MATH_SPLIT || dtype == TTRNN_BF16) && d_packed &&
        f10_ttlinear_wgrad_available(s, dtype, dy_dtype) && (!dx || f10_ttlinear_wgrad_has_dx(s)) &&
        workspace && workspace_bytes >= f10_ttlinear_wgrad_workspace_bytes(s)) {
      TT_ROUTE_PLAIN_ROWS();
      return launch_ttlinear_wgrad_f10(s, dtype, n_rows, packed, x, dy, dx, d_packed, d_bias, workspace,
                                       (hipStream_t)stream);
    }
    // A shape with stage-wise MFMA kernels but no fused-core weight-gradient kernel (the TT-GRU r = 16 of the speaker encoder's
    // --gru variant, speaker_encoder.py:69-78): over many rows the dense-gradient route below — one GEMM per gradient, the
    // pull-back by ttrnn_fast_proj.hip — beats the row-by-row stage-wise backward by an order of magnitude (6 ms -> 0.6 per matrix
    // at 81 920 rows); until the end of round 3 such shapes never reached it
    const GenDense gdf = gen_dense(s);
    const bool gdf_ok = !no_gemm() && gdf.ok && gdf.proj > 0 && d_packed && dy_dtype == TTRNN_F32 && n_rows >= 4 * (int64_t)s.in_size &&
                        (!dx || (gdf.dx_ok && dtype == TTRNN_F32 && fp32_math() == TTRNN_MATH_SPLIT)) && workspace &&
                        workspace_bytes >= gdf.total;
    if (!gdf_ok) {
      TT_ROUTE_PLAIN_ROWS();
      return launch_ttlinear_bwd_fast(s, dtype, dy_dtype, n_rows, packed, x, dy, dx, d_packed, d_bias,
                                      (hipStream_t)stream);
    }
  }
  if (!force_generic() && (d_packed || (dx && !d_bias)) && big_ttlinear_bwd_available(s, dtype, dy_dtype)) {
    // big shape: dx, weight and bias gradients through the merged two-core matrix
    if (!workspace || workspace_bytes < big_ttlinear_bwd_workspace_bytes(s)) return probe ? 0 : TTRNN_ERR_WORKSPACE;
    if (opt(OPT_BIGW_SLICES) || !d_packed) TT_ROUTE_PLAIN_ROWS();      // (the per-row A/B kernel reads x itself)
    // honoured, but not advertised: at this width the materialised copy is the faster way round — cfg5's training step 24.0
    // ms reading `out` in place against 23.6 with the 0.28 ms copy, whose 537 MB the gradient GEMM then finds partly in the
    // Infinity Cache (the narrow shapes gain: cfg4 6.75 -> 6.72)
    if (probe) return 0;
    return launch_ttlinear_bwd_big(s, dtype, n_rows, packed, x, dy, dx, d_packed, d_bias, workspace, (hipStream_t)stream, hx,
                                   hdy, shT, shF);
  }
  if (!force_generic() && !no_gemm() && s.in_size == 1 && !dx && d_packed && workspace && workspace_bytes >= gen_in1_bytes(s) &&
      !opt(OPT_NO_IN1)) {
    // y_n = b + x_n * chain(1): one reduction over the rows, then the any-shape backward on the single unit row
    TT_ROUTE_PLAIN_ROWS();
    hipStream_t sm = (hipStream_t)stream;
    float* dv = (float*)workspace;
    const void* unit = unit_rows_ptr(TTRNN_F32);
    if (!unit) return TTRNN_ERR_LAUNCH;
    const LinPlan p1 = plan_ttlinear_bwd(s, 1);
    if (hsum && !d_bias)
      return launch_ttlinear_bwd(s, p1, TTRNN_F32, TTRNN_F32, 1, packed, unit, hsum, nullptr, d_packed, nullptr,
                                 (char*)workspace + in1_bwd_bytes(s), sm);
    st = launch_in1_reduce(dtype, dy_dtype, n_rows, s.out_size, x, dy, dv, d_bias, (float*)((char*)workspace + in1_dv_bytes(s)), sm);
    if (st != TTRNN_OK) return st;
    return launch_ttlinear_bwd(s, p1, TTRNN_F32, TTRNN_F32, 1, packed, unit, dv, nullptr, d_packed, nullptr,
                               (char*)workspace + in1_bwd_bytes(s), sm);
  }
  const GenDense gd = gen_dense(s);
  // (TTRNN_MATH_EXACT: the dense gradient stays — on the fp32 MFMA — but dx, a GEMM on bf16 pieces, goes back to the chain)
  const bool gd_split = fp32_math() == TTRNN_MATH_SPLIT || dtype == TTRNN_BF16;
  if (!force_generic() && !no_gemm() && gd.ok && d_packed && dy_dtype == TTRNN_F32 && n_rows >= 4 * (int64_t)s.in_size &&
      (!dx || (gd.dx_ok && dtype == TTRNN_F32 && gd_split)) && workspace && workspace_bytes >= gd.total) {
    if (probe) return dense_wgrad_shift_ok(n_rows, hints->x_period) ? 1 : 0;
    hipStream_t sm = (hipStream_t)stream;
    char* p = (char*)workspace;
    void* lin_bwd = p; p += gd.lin_bwd;
    void* ident = p; p += gd.ident;
    float* dWd = (float*)p; p += gd.dwd;
    float* Wd = (float*)p; p += gd.wd;
    void* planes = p; p += gd.planes;
    float* scratch = (float*)p; p += gd.scratch;
    void* lin_fwd = p;
    if (dx || gd.proj == 0) st = launch_fill_identity(TTRNN_F32, s.in_size, ident, sm);
    if (st == TTRNN_OK)
      st = launch_dense_wgrad(dtype, n_rows, s.in_size, s.out_size, x, (const float*)dy, dWd, d_bias, sm, gd_split, scratch, hx,
                              hdy, shT, shF);
    if (st == TTRNN_OK) {
      if (gd.proj > 0) {
        st = launch_proj3(s, packed, dWd, d_packed, (char*)lin_fwd + gd.lin_fwd, sm);
      } else {
        const LinPlan pb = plan_ttlinear_bwd(s, s.in_size, true);      // fixed-order sums: no atomics in the pull-back
        st = launch_ttlinear_bwd(s, pb, TTRNN_F32, TTRNN_F32, s.in_size, packed, ident, dWd, nullptr, d_packed, nullptr, lin_bwd, sm);
      }
    }
    if (st != TTRNN_OK || !dx) return st;
    const LinPlan pf = plan_ttlinear_fwd(s, s.in_size);
    st = launch_ttlinear_fwd(s, pf, TTRNN_F32, s.in_size, packed, nullptr, ident, Wd, lin_fwd, sm);            // W[j][o]
    if (st == TTRNN_OK) st = launch_gemm_split_prep(Wd, s.out_size, s.in_size, planes, sm, true);
    if (st == TTRNN_OK) st = launch_gemm_split(TTRNN_F32, n_rows, s.out_size, s.in_size, dy, planes, nullptr, 0, (float*)dx, sm);
    return st;
  }
  TT_ROUTE_PLAIN_ROWS();
  const LinPlan p = plan_ttlinear_bwd(s, n_rows);
  if (p.ws_bytes > 0 && (!workspace || workspace_bytes < p.ws_bytes)) return TTRNN_ERR_WORKSPACE;
  return launch_ttlinear_bwd(s, p, dtype, dy_dtype, n_rows, packed, x, dy, dx, d_packed, d_bias, workspace,
                             (hipStream_t)stream);
#undef TT_ROUTE_PLAIN_ROWS
}

extern "C" {

int ttrnn_ttlinear_backward_hinted(const ttrnn_ttm* w, int dtype, int dy_dtype, int64_t n_rows, const float* packed,
                                   const void* x, const void* dy, void* dx, float* d_packed, float* d_bias,
                                   const ttrnn_lin_hints* hints, void* workspace, size_t workspace_bytes, void* stream) {
  return lin_backward(w, dtype, dy_dtype, n_rows, packed, x, dy, dx, d_packed, d_bias, hints, workspace, workspace_bytes,
                      stream, false);
}

int ttrnn_ttlinear_backward_shift_ok(const ttrnn_ttm* w, int dtype, int dy_dtype, int64_t n_rows, int64_t x_period,
                                     int want_dx) {
  if (!w || x_period < 1) return 0;
  ttrnn_lin_hints h = {nullptr, nullptr, nullptr, x_period, nullptr};
  // same dispatch as the real call, with stand-in pointers and a workspace as large as ttrnn_ttlinear_workspace promises
  void* one = reinterpret_cast<void*>(uintptr_t(256));
  return lin_backward(w, dtype, dy_dtype, n_rows, reinterpret_cast<const float*>(one), one, one, want_dx ? one : nullptr,
                      reinterpret_cast<float*>(one), nullptr, &h, one, ttrnn_ttlinear_workspace(w, n_rows), nullptr, true) == 1
             ? 1
             : 0;
}


// ---- recurrent layer ----------------------------------------------------------------------------
// The shape-specialised path hoists the input projection: its workspace holds gin = W_in x + b_in for every
// (b, t), gate-interleaved per hidden unit as fp32 [B][T][H][4] (LSTM slots i,g,f,o; GRU r,z,n,-), followed by
// whatever the batched TTLinear launch needs.
struct FastFwdPlan {
  bool use, lin_fast, in1;
  size_t gin_bytes, lin_ws_bytes, f10_bytes, f10_lin_bytes, gemm_bytes;
  LinPlan lin;
};


static FastFwdPlan plan_fast_fwd(const RnnShape& rs, int dtype) {
  FastFwdPlan f{};
  f.use = !force_generic() && fast_rnn_fwd_available(rs, dtype);
  if (!f.use) return f;
  int64_t n_rows = (int64_t)rs.B * rs.T;
  f.lin_fast = fast_ttlinear_fwd_available(rs.in_s, dtype, rs.H);
  // input_size == 1: the projection is linear in a scalar -> only the two rows chain(1)+b, chain(0)+b are needed
  f.in1 = rs.in == 1 && f.lin_fast && !opt(OPT_NO_IN1);
  if (f.in1) n_rows = 2;
  f.gin_bytes = ((size_t)n_rows * 4 * rs.H * sizeof(float) + 255) & ~(size_t)255;   // [rows][H][4]
  if (f.in1) f.gin_bytes += 256;                                                    // + the two unit input rows
  if (!f.lin_fast && dtype != TTRNN_F32) {   // the generic K-in writes storage-typed output; gin must be fp32
    f.use = false;
    return f;
  }
  if (!f.lin_fast) {
    f.lin = plan_ttlinear_fwd(rs.in_s, n_rows);
    f.lin_ws_bytes = (f.lin.ws_bytes + 255) & ~(size_t)255;
  }
  f.f10_bytes = f10_workspace_bytes(rs, dtype);   // reserved whatever the math mode is at query time
  f.f10_lin_bytes = f.in1 ? 0 : f10_ttlinear_workspace_bytes(rs.in_s, dtype, rs.H, rs.cell == TTRNN_LSTM ? 2 : 1);
  // K-in as a dense split-bf16 GEMM (ttrnn_fast_gemm.hip): identity rows, dense W_in, its bf16 planes
  if (f.f10_lin_bytes > 0 && rs.cell == TTRNN_LSTM && dtype == TTRNN_F32 && gemm_split_ok(rs.in, 4 * rs.H))
    f.gemm_bytes = gemm_split_identity_bytes(rs.in) + gemm_split_dense_bytes(rs.in, 4 * rs.H) +
                   gemm_split_plane_bytes(rs.in, 4 * rs.H) + gemm_half_scratch_bytes((int64_t)rs.B * rs.T, rs.in, 4 * rs.H);
  return f;
}

// Forward route preference: fp32 GRUs other than the fused-core shape (H = 256, r = 8: k_gru_fwd_f10vh, round 5) run on the
// runtime-shape tier — since its stages moved to fp16 pieces it beats the stage-wise fp32-MFMA kernels on their shapes (H = 256,
// r = 8, B = 256, T = 784: 0.99 vs 1.58 ms).  The reverse-time route is chosen on its own (same reserve format).
static bool fwd_prefers_g2(const RnnShape& rs, int dtype) {
  if (force_generic()) return false;
  if (opt(OPT_FORCE_G2)) return g2_rnn_available(rs, dtype);
  // (input_size == 1: this file's plan with the fused set-up launch; otherwise the tier's dense K-in in front of the same kernel)
  if (f10gh_available(rs, dtype) && f10gh_own_plan(rs) && fast_rnn_fwd_available(rs, dtype) && (rs.in == 1 || !g2_rnn_available(rs, dtype))) return false;
  return rs.cell == TTRNN_GRU && dtype == TTRNN_F32 && fp32_math() == TTRNN_MATH_SPLIT && g2_rnn_available(rs, dtype);
}

static size_t rnn_workspace_other(const RnnShape& rs, int dtype);
size_t ttrnn_rnn_workspace(const ttrnn_rnn_desc* desc) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return 0;
  // (the encoder-shape kernel's fragments, or whatever the route behind it needs: an option may switch between calls)
  const size_t w2 = w2_rnn_fwd_available(rs, desc->dtype) ? w2_rnn_fwd_workspace_bytes() : 0;
  const size_t o = rnn_workspace_other(rs, desc->dtype);
  return w2 > o ? w2 : o;
}
static size_t rnn_workspace_other(const RnnShape& rs, int dtype) {
  const ttrnn_rnn_desc* desc = nullptr;
  (void)desc;
  if (fwd_prefers_g2(rs, dtype)) return g2_rnn_fwd_workspace(rs);
  const FastFwdPlan f = plan_fast_fwd(rs, dtype);
  if (f.use) return f.gin_bytes + f.lin_ws_bytes + f.f10_bytes + f.f10_lin_bytes + f.gemm_bytes;
  if (!force_generic() && big_rnn_fwd_available(rs, dtype)) return big_rnn_fwd_workspace(rs);
  if (!force_generic() && g2_rnn_available(rs, dtype)) return g2_rnn_fwd_workspace(rs);
  return plan_rnn_generic(rs, false).ws_bytes;
}

size_t ttrnn_rnn_backward_workspace_ex(const ttrnn_rnn_desc* desc, int want_state) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return 0;
  // the same decisions, in the same order, as ttrnn_rnn_backward_ex: a d_state request takes the runtime-shape / any-shape
  // route whatever the shape, and only then is their (per-sample, much larger) plan part of the answer
  const bool gen = force_generic();
  if (!want_state && !gen && rs.T > 0 && w2_rnn_bwd_available(rs, desc->dtype)) return w2_rnn_bwd_workspace_bytes();
  const bool g2_first = (opt(OPT_FORCE_G2) || want_state) && !gen && rs.T > 0 && g2_rnn_bwd_available(rs, desc->dtype);
  if (!g2_first && !want_state && !gen && fast_rnn_bwd_available(rs, desc->dtype)) {
    const size_t a = f10_rnn_bwd_workspace_bytes(rs, desc->dtype);     // fused-core fragments (0 for the stage-wise kernels)
    const size_t b2 = f2_rnn_bwd_workspace_bytes(rs, desc->dtype);     // two-core reverse kernel's fragments (ttrnn_fast_f2.hip)
    return a > b2 ? a : b2;
  }
  if (!g2_first && !want_state && !gen && rs.T > 0 && big_rnn_bwd_available(rs, desc->dtype)) return big_rnn_bwd_workspace(rs);
  if (!gen && rs.T > 0 && g2_rnn_bwd_available(rs, desc->dtype)) {
    const size_t a = g2_rnn_bwd_workspace(rs);
    const size_t b2 = !g2_first && f10bh_h512_available(rs, desc->dtype) ? f10bh_h512_workspace_bytes() : 0;
    const size_t b3 = !g2_first && !want_state && f10n_bwd_available(rs, desc->dtype) ? f10n_bwd_workspace_bytes(rs) : 0;
    return (a > b2 ? a : b2) > b3 ? (a > b2 ? a : b2) : b3;
  }
  return plan_rnn_generic(rs, true).ws_bytes;
}

// without knowledge of the call: enough for either (ttrnn_rnn_backward with or without d_state)
size_t ttrnn_rnn_backward_workspace(const ttrnn_rnn_desc* desc) {
  const size_t a = ttrnn_rnn_backward_workspace_ex(desc, 0), b = ttrnn_rnn_backward_workspace_ex(desc, 1);
  return a > b ? a : b;
}

size_t ttrnn_rnn_reserve_bytes(const ttrnn_rnn_desc* desc) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return 0;
  const size_t per = rs.cell == TTRNN_LSTM ? (size_t)5 * rs.H : (size_t)4 * rs.H;   // res_gate / res_cell (ttrnn_core.h)
  return (size_t)rs.B * rs.T * per * sizeof(float);
}

// does the forward route of this descriptor separate its weight-only launches (TTRNN_PHASE_*)?
static bool phase_split_ok(const RnnShape& rs, int dtype) {
  if (w2_rnn_fwd_available(rs, dtype)) return true;      // (its one weight-only launch: header + fragments)
  if (fwd_prefers_g2(rs, dtype)) return false;
  const FastFwdPlan f = plan_fast_fwd(rs, dtype);
  return f.use && f.in1 && (fp32_math() == TTRNN_MATH_SPLIT || dtype == TTRNN_BF16) && f10_rnn_fwd_available(rs, dtype);
}

// may ttrnn_rnn_forward be called with out == NULL (the caller wants hT / cT only)?  Every forward kernel of every route
// guards its [B][T][H] store (round 3; before: the four-wave fused-core kernel only); only a training forward (reserve != NULL)
// must write `out`, whose rows the weight-gradient step reads as h_{t-1}
static bool out_optional(const RnnShape&, int, bool training) { return !training; }

int ttrnn_rnn_out_optional(const ttrnn_rnn_desc* desc) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return 0;
  return rs.B > 0 && rs.T > 0 && out_optional(rs, desc->dtype, false) ? 1 : 0;
}

int ttrnn_rnn_prepare_supported(const ttrnn_rnn_desc* desc) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return 0;
  return !force_generic() && rs.B > 0 && rs.T > 0 && phase_split_ok(rs, desc->dtype) ? 1 : 0;
}

int ttrnn_rnn_forward(const ttrnn_rnn_desc* desc, const void* x, const void* h0, const void* c0,
                      const float* packed_in, const void* bias_in, const float* packed_hid, const void* bias_hid,
                      void* out, void* hT, void* cT, float* reserve, void* workspace, size_t workspace_bytes,
                      void* stream) {
  return ttrnn_rnn_forward_phase(desc, TTRNN_PHASE_ALL, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve,
                                 workspace, workspace_bytes, stream);
}

int ttrnn_rnn_forward_phase(const ttrnn_rnn_desc* desc, int phase, const void* x, const void* h0, const void* c0,
                            const float* packed_in, const void* bias_in, const float* packed_hid, const void* bias_hid,
                            void* out, void* hT, void* cT, float* reserve, void* workspace, size_t workspace_bytes,
                            void* stream) {
  RnnShape rs;
  int st = rnn_shape_init(&rs, desc);
  if (st != TTRNN_OK) return st;
  if (phase != TTRNN_PHASE_ALL && phase != TTRNN_PHASE_PREPARE && phase != TTRNN_PHASE_RUN) return TTRNN_ERR_BAD_DESC;
  if (rs.B == 0) return TTRNN_OK;
  if (!packed_in || !packed_hid) return TTRNN_ERR_NULL;
  // routes that do not separate their weight-only work: PREPARE has nothing to do, RUN does everything
  if (phase != TTRNN_PHASE_ALL && (force_generic() || rs.T == 0 || !phase_split_ok(rs, desc->dtype))) {
    if (phase == TTRNN_PHASE_PREPARE) return TTRNN_OK;
    phase = TTRNN_PHASE_ALL;
  }
  if (phase != TTRNN_PHASE_PREPARE && rs.T > 0 && (!x || (!out && !out_optional(rs, desc->dtype, reserve != nullptr))))
    return TTRNN_ERR_NULL;
  if (rs.has_bias_in && !bias_in) return TTRNN_ERR_NULL;
  if (rs.has_bias_hid && !bias_hid) return TTRNN_ERR_NULL;
  if (rs.T > 0 && w2_rnn_fwd_available(rs, desc->dtype)) {
    // the speaker encoder's shape: chain stages AND input projection in one persistent kernel (no K-in, no [B][T][4H] buffer)
    if (!workspace || workspace_bytes < w2_rnn_fwd_workspace_bytes()) return TTRNN_ERR_WORKSPACE;
    return launch_rnn_fwd_w2(rs, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve, workspace,
                             (hipStream_t)stream, phase);
  }
  const bool g2_first = fwd_prefers_g2(rs, desc->dtype);
  const FastFwdPlan f = g2_first ? FastFwdPlan{} : plan_fast_fwd(rs, desc->dtype);
  if (f.use) {
    if (!workspace || workspace_bytes < f.gin_bytes + f.lin_ws_bytes + f.f10_bytes + f.f10_lin_bytes + f.gemm_bytes)
      return TTRNN_ERR_WORKSPACE;
    float* gin = (float*)workspace;
    void* lin_ws = (char*)workspace + f.gin_bytes;
    const int ilv_mode = rs.cell == TTRNN_LSTM ? 2 : 1;
    const void* bin = rs.has_bias_in ? bias_in : nullptr;
    GinSrc src{gin, x, f.in1 ? 1 : 0};
    if (f.in1) {
      // K-in on the two unit rows x = [1, 0] (same chain kernel, microseconds); K-rec scales by the real x.  Weights only:
      // TTRNN_PHASE_RUN finds the two rows in the workspace
      const void* unit = unit_rows_ptr(desc->dtype);
      if (!unit) return TTRNN_ERR_LAUNCH;
      if (phase != TTRNN_PHASE_RUN)
        st = launch_ttlinear_fwd_fast(rs.in_s, desc->dtype, true, 2, packed_in, bin, unit, gin, rs.H, ilv_mode,
                                      (hipStream_t)stream);
    } else if (fp32_math() == TTRNN_MATH_SPLIT && f.gemm_bytes > 0 && (int64_t)rs.B * rs.T >= 2 * (int64_t)rs.in &&
               !no_gemm() && f10_ttlinear_fwd_available(rs.in_s, desc->dtype, rs.H, ilv_mode)) {
      // K-in as ONE dense GEMM: W_in (gate-interleaved columns) = the fused-core kernel on the `in` unit rows, then
      // gin = x W_in + b in split-bf16 arithmetic — the chain's extra FLOPs buy nothing where no step is sequential
      char* lin10 = (char*)workspace + f.gin_bytes + f.lin_ws_bytes + f.f10_bytes;
      char* gw = lin10 + f.f10_lin_bytes;
      float* wdense = (float*)(gw + gemm_split_identity_bytes(rs.in));
      void* planes = (char*)wdense + gemm_split_dense_bytes(rs.in, 4 * rs.H);
      st = launch_fill_identity(TTRNN_F32, rs.in, gw, (hipStream_t)stream);
      if (st == TTRNN_OK)
        st = launch_ttlinear_fwd_f10(rs.in_s, rs.in, packed_in, nullptr, gw, wdense, lin10, (hipStream_t)stream);
      // two-piece fp16 operands (three MFMA terms) where the scale passes pay off, else three bf16 pieces (gemm_use_half)
      void* gscr = (char*)planes + gemm_split_plane_bytes(rs.in, 4 * rs.H);
      if (!gemm_use_half((int64_t)rs.B * rs.T, rs.in, 4 * rs.H)) {
        if (st == TTRNN_OK) st = launch_gemm_split_prep(wdense, rs.in, 4 * rs.H, planes, (hipStream_t)stream);
        if (st == TTRNN_OK)
          st = launch_gemm_split(TTRNN_F32, (int64_t)rs.B * rs.T, rs.in, 4 * rs.H, x, planes, bin, rs.H, gin,
                                 (hipStream_t)stream);
      } else {
        if (st == TTRNN_OK) st = launch_gemm_half_prep(wdense, rs.in, 4 * rs.H, planes, gscr, (hipStream_t)stream);
        if (st == TTRNN_OK)
          st = launch_gemm_half(TTRNN_F32, (int64_t)rs.B * rs.T, rs.in, 4 * rs.H, x, planes, gscr, bin, rs.H, gin,
                                (hipStream_t)stream);
      }
    } else if (fp32_math() == TTRNN_MATH_SPLIT && f.f10_lin_bytes > 0 &&
               f10_ttlinear_fwd_available(rs.in_s, desc->dtype, rs.H, ilv_mode)) {
      // K-in of a layer fed by another layer (in = H): fused-core kernel, split fp32 math
      st = launch_ttlinear_fwd_f10(rs.in_s, (int64_t)rs.B * rs.T, packed_in, bin, x, gin,
                                   (char*)workspace + f.gin_bytes + f.lin_ws_bytes + f.f10_bytes, (hipStream_t)stream);
    } else if (f.lin_fast) {
      // K-in: every timestep's input projection in one batched launch (all CUs), then K-rec
      st = launch_ttlinear_fwd_fast(rs.in_s, desc->dtype, true, (int64_t)rs.B * rs.T, packed_in, bin, x, gin, rs.H,
                                    ilv_mode, (hipStream_t)stream);
    } else {
      st = launch_ttlinear_fwd(rs.in_s, f.lin, desc->dtype, (int64_t)rs.B * rs.T, packed_in, bin, x, gin, lin_ws,
                               (hipStream_t)stream, rs.H, ilv_mode);
    }
    if (st != TTRNN_OK) return st;
    // fused-core kernels: fp32 LSTM shapes under the split math mode; the bf16 GRU shape always (bf16 MFMA either way)
    if (desc->dtype == TTRNN_F32 && rs.cell == TTRNN_GRU && f10gh_available(rs, desc->dtype) && f10gh_own_plan(rs))
      return launch_gru_fwd_f10gh(rs, src, h0, packed_hid, bias_hid, out, hT, reserve, (char*)workspace + f.gin_bytes + f.lin_ws_bytes,
                                  (hipStream_t)stream, phase);
    if ((fp32_math() == TTRNN_MATH_SPLIT || desc->dtype == TTRNN_BF16) && f10_rnn_fwd_available(rs, desc->dtype))
      return launch_rnn_fwd_f10(rs, src, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve,
                                (char*)workspace + f.gin_bytes + f.lin_ws_bytes, (hipStream_t)stream, phase);
    if (fp32_math() == TTRNN_MATH_EXACT && f.f10_bytes > 0 && f10x_rnn_fwd_available(rs, desc->dtype))
      return launch_rnn_fwd_f10x(rs, src, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve,
                                 (char*)workspace + f.gin_bytes + f.lin_ws_bytes, (hipStream_t)stream);
    if (fast_rnn_fwd_bf16_available(rs, desc->dtype))
      return launch_rnn_fwd_bf16(rs, src, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, (hipStream_t)stream);
    return launch_rnn_fwd_fast(rs, desc->dtype, src, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve,
                               (hipStream_t)stream);
  }
  if (!g2_first && !force_generic() && big_rnn_fwd_available(rs, desc->dtype)) {
    if (!workspace || workspace_bytes < big_rnn_fwd_workspace(rs)) return TTRNN_ERR_WORKSPACE;
    return launch_rnn_fwd_big(rs, desc->dtype, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT,
                              reserve, workspace, (hipStream_t)stream);
  }
  if (!force_generic() && g2_rnn_available(rs, desc->dtype)) {
    // runtime-shape two-stage MFMA kernels: every other TT shape with d >= 2 cores
    if (!workspace || workspace_bytes < g2_rnn_fwd_workspace(rs)) return TTRNN_ERR_WORKSPACE;
    return launch_rnn_fwd_g2(rs, desc->dtype, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve,
                             workspace, (hipStream_t)stream);
  }
  const RnnPlan p = plan_rnn_generic(rs, false);
  if (p.ws_bytes > 0 && (!workspace || workspace_bytes < p.ws_bytes)) return TTRNN_ERR_WORKSPACE;
  return launch_rnn_fwd_generic(rs, p, desc->dtype, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT,
                                reserve, workspace, (hipStream_t)stream);
}

// ttrnn_pack_cores2 + ttrnn_rnn_forward as ONE call (include/ttrnn.h).  Where the route has a fused set-up launch
// (ttrnn_fast_setup.hip) that is: set-up kernel + recurrent kernel — two launches instead of five; elsewhere exactly the two calls.
int ttrnn_rnn_forward_cores_fused(const ttrnn_rnn_desc* desc) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return 0;
  return !force_generic() && rs.B > 0 && rs.T > 0 && phase_split_ok(rs, desc->dtype) && f10_setup_available(rs, desc->dtype) ? 1 : 0;
}

int ttrnn_rnn_forward_cores(const ttrnn_rnn_desc* desc, const void* x, const void* h0, const void* c0,
                            const void* const* cores_in, const int64_t* strides_in, const void* bias_in,
                            const void* const* cores_hid, const int64_t* strides_hid, const void* bias_hid,
                            float* packed_in, float* packed_hid, void* out, void* hT, void* cT, float* reserve,
                            void* workspace, size_t workspace_bytes, void* stream) {
  RnnShape rs;
  int st = rnn_shape_init(&rs, desc);
  if (st != TTRNN_OK) return st;
  if (!cores_in || !strides_in || !cores_hid || !strides_hid || !packed_in || !packed_hid) return TTRNN_ERR_NULL;
  for (int k = 0; k < rs.in_s.d; ++k) if (!cores_in[k]) return TTRNN_ERR_NULL;
  for (int k = 0; k < rs.hid_s.d; ++k) if (!cores_hid[k]) return TTRNN_ERR_NULL;
  if (rs.B > 0 && rs.T > 0 && ttrnn_rnn_forward_cores_fused(desc)) {
    // everything ttrnn_rnn_forward_phase(RUN) would refuse is refused HERE, before the set-up launch writes the caller's packed
    // buffers and the workspace: a failed call leaves them untouched, as pack + forward does
    if (rs.has_bias_in && !bias_in) return TTRNN_ERR_NULL;
    if (rs.has_bias_hid && !bias_hid) return TTRNN_ERR_NULL;
    if (!x || (!out && !out_optional(rs, desc->dtype, reserve != nullptr))) return TTRNN_ERR_NULL;
    const FastFwdPlan f = plan_fast_fwd(rs, desc->dtype);
    if (!workspace || workspace_bytes < f.gin_bytes + f.lin_ws_bytes + f.f10_bytes + f.f10_lin_bytes + f.gemm_bytes)
      return TTRNN_ERR_WORKSPACE;
    st = launch_f10_setup(rs, cores_in, strides_in, bias_in, cores_hid, strides_hid, packed_in, packed_hid, (float*)workspace,
                          (char*)workspace + f.gin_bytes + f.lin_ws_bytes, (hipStream_t)stream);
    if (st != TTRNN_OK) return st;
    return ttrnn_rnn_forward_phase(desc, TTRNN_PHASE_RUN, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve,
                                   workspace, workspace_bytes, stream);
  }
  st = launch_pack2(rs.in_s, cores_in, strides_in, packed_in, rs.hid_s, cores_hid, strides_hid, packed_hid, desc->dtype,
                    (hipStream_t)stream);
  if (st != TTRNN_OK) return st;
  return ttrnn_rnn_forward(desc, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve, workspace,
                           workspace_bytes, stream);
}

int ttrnn_rnn_forward_route(const ttrnn_rnn_desc* desc) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return TTRNN_ERR_BAD_DESC;
  if (force_generic()) return TTRNN_ROUTE_VALU;
  if (w2_rnn_fwd_available(rs, desc->dtype)) return TTRNN_ROUTE_FUSED_CORE;
  if (fwd_prefers_g2(rs, desc->dtype))      // (the fp32 GRU shape with input_size != 1: the tier's K-in + the fused-core recurrent kernel)
    return (!opt(OPT_FORCE_G2) && (f10gh_available(rs, desc->dtype) || f10g5_available(rs, desc->dtype) || f10n_available(rs, desc->dtype))) ? TTRNN_ROUTE_FUSED_CORE
                                                                                                           : TTRNN_ROUTE_RUNTIME_MFMA;
  const FastFwdPlan f = plan_fast_fwd(rs, desc->dtype);
  if (f.use) {
    if ((fp32_math() == TTRNN_MATH_SPLIT || desc->dtype == TTRNN_BF16) && f10_rnn_fwd_available(rs, desc->dtype))
      return TTRNN_ROUTE_FUSED_CORE;
    if (fp32_math() == TTRNN_MATH_EXACT && f.f10_bytes > 0 && f10x_rnn_fwd_available(rs, desc->dtype))
      return TTRNN_ROUTE_FUSED_CORE;
    return TTRNN_ROUTE_STAGEWISE_MFMA;
  }
  if (big_rnn_fwd_available(rs, desc->dtype)) return TTRNN_ROUTE_MERGED_BIG;
  if (g2_rnn_available(rs, desc->dtype))      // (H = 512, r = 8 in split mode: this tier's K-in + the fused-core recurrent kernel)
    return (f10_h512_fwd_available(rs, desc->dtype) || f10gh_available(rs, desc->dtype) || f10g5_available(rs, desc->dtype) ||
            (!opt(OPT_FORCE_G2) && f10n_available(rs, desc->dtype)))
               ? TTRNN_ROUTE_FUSED_CORE : TTRNN_ROUTE_RUNTIME_MFMA;
  return TTRNN_ROUTE_VALU;
}

int ttrnn_rnn_forward_samples_per_workgroup(const ttrnn_rnn_desc* desc) {
  const int route = ttrnn_rnn_forward_route(desc);
  if (route < 0) return route;
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return TTRNN_ERR_BAD_DESC;
  return (route == TTRNN_ROUTE_RUNTIME_MFMA && g2_rnn_fwd_paired(rs)) ? 2 : 1;
}

int ttrnn_rnn_backward_route(const ttrnn_rnn_desc* desc, int want_state) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return TTRNN_ERR_BAD_DESC;
  if (force_generic()) return TTRNN_ROUTE_VALU;
  if (!want_state && rs.T > 0 && w2_rnn_bwd_available(rs, desc->dtype)) return TTRNN_ROUTE_FUSED_CORE;
  const bool g2_first = (opt(OPT_FORCE_G2) || want_state) && rs.T > 0 && g2_rnn_bwd_available(rs, desc->dtype);
  if (!g2_first && !want_state && fast_rnn_bwd_available(rs, desc->dtype)) {
    if ((fp32_math() == TTRNN_MATH_SPLIT || desc->dtype == TTRNN_BF16) && rs.T > 0 && f10_rnn_bwd_available(rs, desc->dtype))
      return TTRNN_ROUTE_FUSED_CORE;
    if (fp32_math() == TTRNN_MATH_SPLIT && rs.T > 0 && f2_rnn_bwd_available(rs, desc->dtype)) return TTRNN_ROUTE_FUSED_CORE;
    return TTRNN_ROUTE_STAGEWISE_MFMA;
  }
  if (!g2_first && !want_state && rs.T > 0 && big_rnn_bwd_available(rs, desc->dtype)) return TTRNN_ROUTE_MERGED_BIG;
  if (rs.T > 0 && g2_rnn_bwd_available(rs, desc->dtype))
    return !g2_first && (f10bh_h512_available(rs, desc->dtype) || (!want_state && f10n_bwd_available(rs, desc->dtype)))
               ? TTRNN_ROUTE_FUSED_CORE
               : TTRNN_ROUTE_RUNTIME_MFMA;
  return TTRNN_ROUTE_VALU;
}

// which by-products (TTRNN_BWD_STATS_*) does the reverse-time route of this descriptor deliver?  The fused-core kernels:
// the column maxima, and the input_size == 1 sums; the merged-big kernels: the column maxima
static int bwd_stats_mask(const RnnShape& rs, int dtype) {
  if (force_generic() || opt(OPT_FORCE_G2) || rs.T < 1 || rs.B < 1) return 0;
  if (w2_rnn_bwd_available(rs, dtype)) return TTRNN_BWD_STATS_COLMAX;
  if (fast_rnn_bwd_available(rs, dtype) && (fp32_math() == TTRNN_MATH_SPLIT || dtype == TTRNN_BF16) &&
      f10_rnn_bwd_available(rs, dtype))
    return TTRNN_BWD_STATS_COLMAX | (rs.in == 1 ? TTRNN_BWD_STATS_IN1SUMS : 0) |
           (rs.cell == TTRNN_LSTM && f10bh_available(rs, dtype) ? TTRNN_BWD_STATS_ROWMAX : 0);
  if (!fast_rnn_bwd_available(rs, dtype) && big_rnn_bwd_available(rs, dtype)) return TTRNN_BWD_STATS_COLMAX;
  // the runtime-shape reverse-time kernel (the route ttrnn_rnn_backward_ex takes when neither of the above exists): the column
  // maxima where its LDS plan has room for them
  if (!fast_rnn_bwd_available(rs, dtype) && !big_rnn_bwd_available(rs, dtype) && g2_rnn_bwd_available(rs, dtype) &&
      g2_rnn_bwd_colmax(rs))
    return TTRNN_BWD_STATS_COLMAX;
  return 0;
}

int ttrnn_rnn_backward_stats(const ttrnn_rnn_desc* desc) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return 0;
  return bwd_stats_mask(rs, desc->dtype);
}

int ttrnn_rnn_backward(const ttrnn_rnn_desc* desc, const void* out, const void* h0, const void* c0,
                       const float* packed_hid, const float* reserve, const void* d_out, const void* d_hT,
                       const void* d_cT, float* d_gates_in, float* d_gates_hid, void* d_h0, void* d_c0,
                       float* d_state, void* workspace, size_t workspace_bytes, void* stream) {
  return ttrnn_rnn_backward_ex(desc, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in, d_gates_hid, d_h0, d_c0,
                               d_state, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int ttrnn_rnn_backward_ex(const ttrnn_rnn_desc* desc, const void* out, const void* h0, const void* c0,
                          const float* packed_hid, const float* reserve, const void* d_out, const void* d_hT,
                          const void* d_cT, float* d_gates_in, float* d_gates_hid, void* d_h0, void* d_c0,
                          float* d_state, const void* x, float* stats, void* workspace, size_t workspace_bytes,
                          void* stream) {
  RnnShape rs;
  int st = rnn_shape_init(&rs, desc);
  if (st != TTRNN_OK) return st;
  if (rs.B == 0) return TTRNN_OK;
  // by-products come from the fused-core and the merged-big reverse kernels only: a request elsewhere (or together with d_state, which
  // selects another route) is an error, not a silent omission — ask ttrnn_rnn_backward_stats first
  if (stats && (d_state || bwd_stats_mask(rs, desc->dtype) == 0)) return TTRNN_ERR_UNSUPPORTED;
  if (!packed_hid) return TTRNN_ERR_NULL;
  if (rs.T > 0 && (!reserve || !d_gates_in || !out)) return TTRNN_ERR_NULL;
  if (rs.cell == TTRNN_GRU && rs.T > 0 && !d_gates_hid) return TTRNN_ERR_NULL;
  // d_state (the per-step state gradients ActivGradLogger records) is written by the runtime-shape and the any-shape
  // reverse kernels; a request for it takes those routes (the shape-specialised kernels stay untouched)
  const bool g2_first = (opt(OPT_FORCE_G2) || d_state) && !force_generic() && rs.T > 0 && g2_rnn_bwd_available(rs, desc->dtype);
  const bool want_state = d_state != nullptr;
  if (!want_state && !force_generic() && rs.T > 0 && w2_rnn_bwd_available(rs, desc->dtype)) {
    // the speaker encoder's shape: wave-local transposed stages, one barrier per step (ttrnn_fast_w2.hip)
    if (!workspace || workspace_bytes < w2_rnn_bwd_workspace_bytes()) return TTRNN_ERR_WORKSPACE;
    return launch_rnn_bwd_w2(rs, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in, d_gates_hid, d_h0, d_c0, workspace,
                             (hipStream_t)stream, stats);
  }
  if (!g2_first && !want_state && !force_generic() && fast_rnn_bwd_available(rs, desc->dtype)) {
    if ((fp32_math() == TTRNN_MATH_SPLIT || desc->dtype == TTRNN_BF16) && rs.T > 0 &&
        f10_rnn_bwd_available(rs, desc->dtype)) {
      if (!workspace || workspace_bytes < f10_rnn_bwd_workspace_bytes(rs, desc->dtype)) return TTRNN_ERR_WORKSPACE;
      return launch_rnn_bwd_f10(rs, desc->dtype, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in,
                                d_gates_hid, d_h0, d_c0, workspace, (hipStream_t)stream, x, stats);
    }
    // two-core hidden matrices in split mode: both transposed stages on two fp16 pieces (ttrnn_fast_f2.hip)
    if (fp32_math() == TTRNN_MATH_SPLIT && rs.T > 0 && f2_rnn_bwd_available(rs, desc->dtype) && workspace &&
        workspace_bytes >= f2_rnn_bwd_workspace_bytes(rs, desc->dtype))
      return launch_rnn_bwd_f2(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in, d_gates_hid, d_h0, d_c0, workspace,
                               (hipStream_t)stream);
    return launch_rnn_bwd_fast(rs, desc->dtype, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in,
                               d_gates_hid, d_h0, d_c0, (hipStream_t)stream);
  }
  if (!g2_first && !want_state && !force_generic() && rs.T > 0 && big_rnn_bwd_available(rs, desc->dtype)) {
    if (!workspace || workspace_bytes < big_rnn_bwd_workspace(rs)) return TTRNN_ERR_WORKSPACE;
    return launch_rnn_bwd_big(rs, desc->dtype, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in, d_gates_hid,
                              d_h0, d_c0, workspace, (hipStream_t)stream, stats);
  }
  if (!force_generic() && rs.T > 0 && g2_rnn_bwd_available(rs, desc->dtype)) {
    // the reference's default benchmark shape in split mode: the fused-core reverse-time kernel on two fp16 pieces
    if (!g2_first && f10bh_h512_available(rs, desc->dtype) && workspace && workspace_bytes >= f10bh_h512_workspace_bytes())
      return launch_rnn_bwd_f10_h512(rs, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in, d_gates_hid, d_h0, d_c0,
                                     workspace, (hipStream_t)stream, stats);
    // naive per-gate sets of H = 256: the per-gate fused cores (k_rnn_bwd_f10n, round 6)
    if (!g2_first && !want_state && f10n_bwd_available(rs, desc->dtype) && workspace && workspace_bytes >= f10n_bwd_workspace_bytes(rs))
      return launch_rnn_bwd_f10n(rs, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in, d_gates_hid, d_h0, d_c0, workspace,
                                 (hipStream_t)stream, stats);
    if (!workspace || workspace_bytes < g2_rnn_bwd_workspace(rs)) return TTRNN_ERR_WORKSPACE;
    return launch_rnn_bwd_g2(rs, desc->dtype, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in, d_gates_hid,
                             d_h0, d_c0, workspace, (hipStream_t)stream, d_state, stats);
  }
  const RnnPlan p = plan_rnn_generic(rs, true);
  if (p.ws_bytes > 0 && (!workspace || workspace_bytes < p.ws_bytes)) return TTRNN_ERR_WORKSPACE;
  return launch_rnn_bwd_generic(rs, p, desc->dtype, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, d_gates_in,
                                d_gates_hid, d_h0, d_c0, workspace, (hipStream_t)stream, d_state);
}

}  // extern "C"


// ---- ABI 7: chain weight gradients of a recurrent layer (ttrnn_fast_c2w.hip) ---------------------------------------------------
// which matrices of `mats` the chain kernel takes in ONE launch for this descriptor (0: none)
static int wgrad_chain_mats(const RnnShape& rs, int dtype, int mats, const TtShape* shapes[2], int* hid_slot) {
  *hid_slot = -1;
  if (dtype != TTRNN_F32 || fp32_math() != TTRNN_MATH_SPLIT || force_generic() || no_gemm() || (opt(OPT_DEV2) & 1)) return 0;
  if (rs.hid_blocks > 1) return 0;                 // (joint matrices of naive sets: their block structure is not exploited here)
  if (!(mats & 2) || !c2w_prefers_chain(rs.hid_s)) return 0;      // the hidden matrix decides
  int n = 0;
  const bool small = !(opt(OPT_DEV2) & 8);      // (dev2 bit 3: the large kernel variant too — measured slower than the dense gradient)
  // the input matrix rides along where it shares the gate gradients (LSTM) and the kernel takes the pair
  if ((mats & 1) && rs.cell == TTRNN_LSTM && rs.in >= 4 && !(opt(OPT_DEV2) & 2)) {
    const TtShape* both[2] = {&rs.in_s, &rs.hid_s};
    if (c2w_workspace_bytes(both, 2, small) > 0) { shapes[0] = &rs.in_s; shapes[1] = &rs.hid_s; *hid_slot = 1; return 3; }
  }
  shapes[0] = &rs.hid_s;
  *hid_slot = 0;
  n = c2w_workspace_bytes(shapes, 1, small) > 0 ? 2 : 0;
  return n;
}

extern "C" {

size_t ttrnn_rnn_wgrad_workspace(const ttrnn_rnn_desc* desc, int mats) {
  RnnShape rs;
  if (rnn_shape_init(&rs, desc) != TTRNN_OK) return 0;
  const TtShape* shapes[2] = {nullptr, nullptr};
  int hs;
  const int taken = wgrad_chain_mats(rs, desc->dtype, mats, shapes, &hs);
  if (taken != mats) return 0;                     // all or nothing: the caller keeps ONE code path per layer
  return c2w_workspace_bytes(shapes, taken == 3 ? 2 : 1);
}

int ttrnn_rnn_wgrad(const ttrnn_rnn_desc* desc, int mats, const ttrnn_wgrad_args* a, void* workspace, size_t workspace_bytes,
                    void* stream) {
  RnnShape rs;
  int st = rnn_shape_init(&rs, desc);
  if (st != TTRNN_OK) return st;
  if (!a) return TTRNN_ERR_NULL;
  const TtShape* shapes[2] = {nullptr, nullptr};
  int hs;
  const int taken = wgrad_chain_mats(rs, desc->dtype, mats, shapes, &hs);
  if (taken == 0 || taken != mats) return TTRNN_ERR_UNSUPPORTED;
  if (rs.B == 0 || rs.T == 0) return TTRNN_OK;
  if (!a->out || !a->d_gates_hid || !a->packed_hid || !a->d_packed_hid) return TTRNN_ERR_NULL;
  if (taken == 3 && (!a->x || !a->packed_in || !a->d_packed_in || !a->d_gates_in)) return TTRNN_ERR_NULL;
  if (taken == 3 && a->d_gates_in != a->d_gates_hid) return TTRNN_ERR_BAD_DESC;      // one pass over dy = one buffer
  const float* packed[2]; const float* x[2]; const float* first[2]; int T[2]; const unsigned* xc[2]; int xn[2]; float* dp[2];
  const int n = taken == 3 ? 2 : 1;
  if (n == 2) {
    packed[0] = a->packed_in; x[0] = (const float*)a->x; first[0] = nullptr; T[0] = 0;
    xc[0] = reinterpret_cast<const unsigned*>(a->x_colmax); xn[0] = rs.in; dp[0] = a->d_packed_in;
  }
  packed[hs] = a->packed_hid; x[hs] = (const float*)a->out; first[hs] = (const float*)a->h0; T[hs] = rs.T;
  xc[hs] = reinterpret_cast<const unsigned*>(a->h_colmax); xn[hs] = rs.H; dp[hs] = a->d_packed_hid;
  return launch_c2w(shapes, n, (int64_t)rs.B * rs.T, packed, x, first, T, a->d_gates_hid,
                    xc, xn, reinterpret_cast<const unsigned*>(a->dy_colmax_hid), dp, n == 2 ? a->d_bias_in : nullptr, a->d_bias_hid,
                    workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
