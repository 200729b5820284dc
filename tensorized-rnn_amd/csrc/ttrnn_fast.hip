// ttrnn_fast.hip — shape-specialised persistent recurrent kernel (gfx950, fp32 MFMA).
//
// One workgroup (8 waves) owns one sample for the whole sequence:
//   * the hidden TT chain runs stage by stage on v_mfma_f32_16x16x4_f32 (exact fp32, same rate as
//     the fp32 VALU peak but one operand VGPR per 1024 MACs);  the MFMA "A" operand is the CORE
//     (output features on the 16 MFMA rows), the "B" operand the activations (chain rows on the 16
//     MFMA columns) — so each lane's 4 accumulator registers are 4 consecutive rank indices and a
//     stage result is stored with one ds_write_b128 per tile;
//   * every core fragment a wave needs is loaded ONCE into VGPRs (cfg2: 26 registers per lane) and
//     stays there for all timesteps: no weight traffic at all inside the time loop;
//   * chain intermediates ping-pong through LDS in an XOR-swizzled [row][K] image whose b128
//     fragment reads are bank-conflict-free (slot ^= f(row), see a_off);
//   * the cell state c lives in a register of the thread that owns the hidden unit, h goes straight
//     back into the LDS image the first chain stage reads;
//   * the non-recurrent input projection W_in x_t + b_in is hoisted out of the recurrence (one
//     batched TTLinear launch over all B*T rows on all CUs); the time loop prefetches its row for
//     step t+1 into registers while step t computes, and the per-step barriers are raw s_barrier +
//     lgkmcnt(0) so that prefetch (and the output stores) stay in flight across them.
//
// Replaces, for one layer: tensorized_rnn/lstm.py:23-32,123-133 and gru.py:33-44,124-134 with the
// hidden-weights chain of t3nsor/ops.py:78-93 (reference file:line).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"

namespace ttrnn {

__device__ __forceinline__ void store4(float* y, size_t idx, f32x4 v) { *reinterpret_cast<f32x4*>(y + idx) = v; }
__device__ __forceinline__ void store4(bf16_t* y, size_t idx, f32x4 v) {
  typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
  u16x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = f32_to_bf16(v[j]).v;
  *reinterpret_cast<u16x4*>(y + idx) = r;
}

// what a value becomes after a round trip through the storage type (h is fed back as stored)
template <typename TS>
__device__ __forceinline__ float round_storage(float v) {
  if constexpr (sizeof(TS) == 2) return bf16_to_f32(f32_to_bf16(v));
  else return v;
}

// ---- the persistent kernel ---------------------------------------------------------------------------
// gin: fp32 [B][T][G*H] = W_in x_t + b_in (hoisted).  One workgroup per sample.
template <class S, int CELL, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_rnn_fwd_fast(int B, int T, GinSrc gs,
                                                          const TS* __restrict__ h0, const TS* __restrict__ c0,
                                                          const float* __restrict__ packed_hid,
                                                          const TS* __restrict__ bias_hid, TS* __restrict__ out,
                                                          TS* __restrict__ hT, TS* __restrict__ cT,
                                                          float* __restrict__ reserve) {
  constexpr int D = S::D;
  constexpr int H = in_size_of<S>();
  constexpr int G = CELL == TTRNN_LSTM ? 4 : 3;
  constexpr int GH = G * H;
  static_assert(out_size_of<S>() == GH, "TT output size must be n_gates * hidden");
  if (gs.run_if) {                                       // queued as a fallback (GinSrc::run_if): workgroup-uniform
    if (*gs.run_if == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && gs.status) atomicAdd(gs.status + TTRNN_STAT_GUARD_TRIPS, 1u);
  }
  constexpr int HPT = (H + FAST_NT - 1) / FAST_NT;       // hidden units per thread
  constexpr int MID = maxmid_of<S>();
  using SL = St<S, D - 1>;                               // first stage executed (reads h)

  __shared__ __attribute__((aligned(16))) float hbuf[H];
  __shared__ __attribute__((aligned(16))) float bufA[MID];
  __shared__ __attribute__((aligned(16))) float bufB[D > 2 ? MID : 4];
  __shared__ __attribute__((aligned(16))) float gbuf[GH];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t b = blockIdx.x;

  // resident core fragments
  float w0[nwreg<S, 0>()];
  float w1[nwreg<S, (D > 1 ? 1 : 0)>()];
  float w2[nwreg<S, (D > 2 ? 2 : 0)>()];
  float w3[nwreg<S, (D > 3 ? 3 : 0)>()];
  load_wfrag<S, 0>(w0, packed_hid, wave, lane);
  if constexpr (D > 1) load_wfrag<S, 1>(w1, packed_hid, wave, lane);
  if constexpr (D > 2) load_wfrag<S, 2>(w2, packed_hid, wave, lane);
  if constexpr (D > 3) load_wfrag<S, 3>(w3, packed_hid, wave, lane);

  // per-thread recurrent state and constants for its hidden units
  // gin is gate-interleaved: [B][T][H][4], slot order i,g,f,o (LSTM) / r,z,n,- (GRU): one 16-byte load
  const float* __restrict__ gin = gs.gin;
  const TS* __restrict__ xs = reinterpret_cast<const TS*>(gs.x);
  const bool in1 = gs.in1 != 0;
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);
  float cst[HPT], hst[HPT], bh[HPT][G];
  f32x4 gi[HPT], vv[HPT], bb[HPT];
  XChunk<TS> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    const bool ok = hid < H;
    hst[u] = (ok && h0) ? ld(h0, b * H + hid) : 0.f;
    cst[u] = (ok && c0 && CELL == TTRNN_LSTM) ? ld(c0, b * H + hid) : 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) bh[u][g] = (ok && bias_hid) ? ld(bias_hid, g * H + hid) : 0.f;
    vv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    bb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    gi[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok && T > 0) {
      if (in1) {
        bb[u] = gin4[H + hid];
        vv[u] = gin4[hid] - bb[u];
      } else {
        gi[u] = gin4[(b * T) * H + hid];
      }
    }
    if (ok) hbuf[a_off<SL::K>(hid / SL::K, hid % SL::K)] = hst[u];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see k_lstm_fwd_fused
  lds_barrier();

  for (int t = 0; t < T; ++t) {
    // ---- hidden chain: stage D-1 .. 0 --------------------------------------------------------------
    if constexpr (D == 1) {
      run_stage<S, 0>(w0, hbuf, gbuf, wave, lane);
    } else if constexpr (D == 2) {
      run_stage<S, 1>(w1, hbuf, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 0>(w0, bufA, gbuf, wave, lane);
    } else if constexpr (D == 3) {
      run_stage<S, 2>(w2, hbuf, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 1>(w1, bufA, bufB, wave, lane);
      lds_barrier();
      run_stage<S, 0>(w0, bufB, gbuf, wave, lane);
    } else {
      run_stage<S, 3>(w3, hbuf, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 2>(w2, bufA, bufB, wave, lane);
      lds_barrier();
      run_stage<S, 1>(w1, bufB, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 0>(w0, bufA, gbuf, wave, lane);
    }
    lds_barrier();
    // ---- gates + state update (lstm.py:26-32 / gru.py:38-44) ----------------------------------------
    const size_t bt = b * T + t;
    const float xt = in1 ? xq.at(t) : 0.f;
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
      const int hid = tid + u * FAST_NT;
      if (hid < H) {
        float hy;
        if (in1) gi[u] = bb[u] + xt * vv[u];      // W_in x_t + b_in from the two unit rows (GinSrc)
        if constexpr (CELL == TTRNN_LSTM) {
          const float ig = fsigmoid(gi[u][0] + gbuf[hid] + bh[u][0]);
          const float fg = fsigmoid(gi[u][2] + gbuf[H + hid] + bh[u][1]);
          const float gg = ftanh(gi[u][1] + gbuf[2 * H + hid] + bh[u][2]);
          const float og = fsigmoid(gi[u][3] + gbuf[3 * H + hid] + bh[u][3]);
          const float cy = fg * cst[u] + ig * gg;
          hy = og * ftanh(cy);
          cst[u] = cy;
          if (reserve) {
            float* rv = reserve + res_gate(bt, H, hid);
            rv[0] = ig; rv[1] = gg; rv[2] = fg; rv[3] = og;
            reserve[res_cell((size_t)B * T, bt, H, hid)] = cy;
          }
        } else {
          const float hn = gbuf[2 * H + hid] + bh[u][2];
          const float rg = fsigmoid(gi[u][0] + gbuf[hid] + bh[u][0]);
          const float zg = fsigmoid(gi[u][1] + gbuf[H + hid] + bh[u][1]);
          const float ng = ftanh(gi[u][2] + rg * hn);
          hy = (1.0f - zg) * ng + zg * hst[u];
          if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
        }
        // the stored output is what the next step (and the next layer) sees: round once to the storage type
        if (out) st(out, bt * H + hid, hy);          // out == NULL: final state only (ttrnn_rnn_out_optional)
        hy = round_as(out, hy);
        hst[u] = hy;
        hbuf[a_off<SL::K>(hid / SL::K, hid % SL::K)] = hy;
        // prefetch the hoisted input projection of the next step; consumed one iteration later
        if (!in1 && t + 1 < T) gi[u] = gin4[(bt + 1) * H + hid];
      }
    }
    if (in1) xq.advance(xs, b * T, T, t, lane);
    lds_barrier();
  }
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    if (hid < H) {
      if (hT) st(hT, b * H + hid, hst[u]);
      if (CELL == TTRNN_LSTM && cT) st(cT, b * H + hid, cst[u]);
    }
  }
}

// ---- LSTM: last chain stage fused with the gate arithmetic --------------------------------------------
// o = gate*H + hid and o = i_0*ROWS_0 + row give i_0 = gate*P + par, hid = par*ROWS_0 + row (P = I_0/4).
// Loading the stage-0 core with its rows permuted — MFMA row p = 4*qq + j of m-tile mt <-> i_0 = j*P + (4*mt+qq) —
// puts the four gate pre-activations of ONE hidden unit into the four accumulator registers of ONE lane
// (lane (c, q): hid = (4*mt+q)*ROWS_0 + 16*rt + c).  The gate math then runs straight on the accumulators:
// no gate buffer in LDS and one barrier fewer per timestep; c and the hoisted gate inputs live in that lane.
template <class S>
constexpr bool lstm_fusable() { return S::D >= 2 && S::R[0] == 1 && S::I[0] % 4 == 0; }

// PAIR variant (P <= 2, one m-tile): MFMA row p = 4*qq + j holds, for pair = qq >> 1 and par = qq & 1, gate
// (pair ? f : i) in j = 0 and (pair ? o : g) in j = 1; rows j = 2, 3 are zero.  Lanes 0-31 then own (i, g) and
// lanes 32-63 own (f, o) of the same hidden unit: every lane evaluates two non-linearities instead of five and
// one v_permlane32_swap hands i*g to the lane that keeps c.
template <class S>
constexpr bool lstm_pair_mode() { return S::I[0] / 4 <= 2; }

template <class S, int NW_>
__device__ __forceinline__ void load_wfrag0_lstm(float (&w)[NW_], const float* packed, int wave, int lane) {
  using T = St<S, 0>;
  static_assert(NW_ == T::NWREG, "weight fragment array size");
  constexpr int P = S::I[0] / 4;
  const int r = lane & 15, q = lane >> 4;
  const int j = r & 3, qq = r >> 2;
#pragma unroll
  for (int x = 0; x < T::XM; ++x) {
    const int mt = T::SPLIT ? (wave % T::MT) : (wave + FAST_NW * x);
    int par = 4 * mt + qq;
    int i0 = j * P + par;
    if constexpr (lstm_pair_mode<S>()) {
      const int pair = qq >> 1;
      par = (j < 2) ? (qq & 1) : P;                                      // rows j = 2,3 -> zero
      const int gate = (j == 0) ? (pair ? 1 : 0) : (pair ? 3 : 2);
      i0 = gate * P + (qq & 1);
    }
#pragma unroll
    for (int u = 0; u < T::NU; ++u)
#pragma unroll
      for (int e = 0; e < T::WV; ++e) {
        const int kk = (4 * u + q) * T::WV + e;
        w[x * T::NSTEP + u * T::WV + e] = (mt < T::MT && par < P) ? packed[kk * T::M + i0] : 0.f;   // M = I_0
      }
  }
}

template <class S, bool DIAG, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_lstm_fwd_fused(int B, int T, GinSrc gs,
                                                            const TS* __restrict__ h0, const TS* __restrict__ c0,
                                                            const float* __restrict__ packed_hid,
                                                            const TS* __restrict__ bias_hid, TS* __restrict__ out,
                                                            TS* __restrict__ hT, TS* __restrict__ cT,
                                                            float* __restrict__ reserve) {
  constexpr int D = S::D;
  constexpr int H = in_size_of<S>();
  constexpr int GH = 4 * H;
  static_assert(out_size_of<S>() == GH, "TT output size must be 4 * hidden");
  if (gs.run_if) {                                       // queued as a fallback (GinSrc::run_if): workgroup-uniform
    if (*gs.run_if == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && gs.status) atomicAdd(gs.status + TTRNN_STAT_GUARD_TRIPS, 1u);
  }
  constexpr int MID = maxmid_of<S>();
  using SL = St<S, D - 1>;
  using T0 = St<S, 0>;
  constexpr int P = S::I[0] / 4;
  static_assert(P * T0::ROWS == H, "hidden index decomposition");

  __shared__ __attribute__((aligned(16))) float hbuf[H];
  __shared__ __attribute__((aligned(16))) float bufA[MID];
  __shared__ __attribute__((aligned(16))) float bufB[D > 2 ? MID : 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  float w0[nwreg<S, 0>()];
  float w1[nwreg<S, (D > 1 ? 1 : 0)>()];
  float w2[nwreg<S, (D > 2 ? 2 : 0)>()];
  float w3[nwreg<S, (D > 3 ? 3 : 0)>()];
  load_wfrag0_lstm<S>(w0, packed_hid, wave, lane);
  if constexpr (D > 1) load_wfrag<S, 1>(w1, packed_hid, wave, lane);
  if constexpr (D > 2) load_wfrag<S, 2>(w2, packed_hid, wave, lane);
  if constexpr (D > 3) load_wfrag<S, 3>(w3, packed_hid, wave, lane);

  // hidden units owned by this lane (one per stage-0 tile of the wave).  gin is gate-interleaved
  // [B][T][H][4] with slot order i,g,f,o.  PAIR: lanes 0-31 take slots (i,g), lanes 32-63 (f,o) and own c.
  constexpr bool PAIR = lstm_pair_mode<S>();
  constexpr int NG = PAIR ? 2 : 4;                  // gates evaluated per lane
  const int pair = PAIR ? (q >> 1) : 1;             // 1 = this lane owns c / h of its hidden unit
  const float sgn = (PAIR && pair == 0) ? 2.0f : 1.0f;   // second gate: tanh (= 2*sigmoid(2x) - 1) or sigmoid
  int hid[T0::XM][T0::YR];
  bool ok[T0::XM][T0::YR];
  const float* __restrict__ gin = gs.gin;
  const TS* __restrict__ xs = reinterpret_cast<const TS*>(gs.x);
  const bool in1 = gs.in1 != 0;
  float cst[T0::XM][T0::YR], hst[T0::XM][T0::YR], bh[T0::XM][T0::YR][NG], gi[T0::XM][T0::YR][NG];
  float vv[T0::XM][T0::YR][NG], bb[T0::XM][T0::YR][NG];
  XChunk<TS> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
#pragma unroll
  for (int x = 0; x < T0::XM; ++x) {
    const int mt = T0::SPLIT ? (wave % T0::MT) : (wave + FAST_NW * x);
#pragma unroll
    for (int y = 0; y < T0::YR; ++y) {
      const int rt = T0::SPLIT ? (wave / T0::MT + T0::G * y) : y;
      const int row = 16 * rt + c, par = PAIR ? (q & 1) : (4 * mt + q);
      ok[x][y] = mt < T0::MT && rt < T0::RT && row < T0::ROWS && par < P;
      hid[x][y] = ok[x][y] ? par * T0::ROWS + row : 0;
      const int hd = hid[x][y];
      hst[x][y] = (ok[x][y] && h0) ? ld(h0, b * H + hd) : 0.f;
      cst[x][y] = (ok[x][y] && c0) ? ld(c0, b * H + hd) : 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        // slot s of the interleaved layout <-> reference gate index: i,g,f,o -> 0,2,1,3
        const int slot = PAIR ? 2 * pair + g : g;
        const int gate = slot == 1 ? 2 : (slot == 2 ? 1 : slot);
        bh[x][y][g] = (ok[x][y] && bias_hid) ? ld(bias_hid, gate * H + hd) : 0.f;
        vv[x][y][g] = 0.f; bb[x][y][g] = 0.f; gi[x][y][g] = 0.f;
        if (ok[x][y] && T > 0) {
          if (in1) {
            bb[x][y][g] = gin[(H + hd) * 4 + slot];
            vv[x][y][g] = gin[hd * 4 + slot] - bb[x][y][g];
          } else {
            gi[x][y][g] = gin[((b * T) * H + hd) * 4 + slot];
          }
        }
      }
      if (ok[x][y] && pair) hbuf[a_off<SL::K>(hd / SL::K, hd % SL::K)] = hst[x][y];
    }
  }
  // All pre-loop global loads (core fragments, biases, first gate inputs) must be provably complete here:
  // otherwise hipcc guards the first use of a weight register inside the loop with s_waitcnt vmcnt(0), which in
  // steady state drains the gate-input prefetch issued a few hundred cycles earlier (one exposed HBM latency
  // per timestep).  0x0F70 = vmcnt(0) only.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  for (int t = 0; t < T; ++t) {
    const float* last;   // input image of stage 0
    if constexpr (D == 2) {
      run_stage<S, 1>(w1, hbuf, bufA, wave, lane);
      TT_STAMP(2)
      last = bufA;
    } else if constexpr (D == 3) {
      run_stage<S, 2>(w2, hbuf, bufA, wave, lane);
      TT_STAMP(0)
      lds_barrier();
      TT_STAMP(1)
      run_stage<S, 1>(w1, bufA, bufB, wave, lane);
      TT_STAMP(2)
      last = bufB;
    } else {
      run_stage<S, 3>(w3, hbuf, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 2>(w2, bufA, bufB, wave, lane);
      lds_barrier();
      run_stage<S, 1>(w1, bufB, bufA, wave, lane);
      last = bufA;
    }
    lds_barrier();
    TT_STAMP(3)
    f32x4 acc[T0::XM][T0::YR];
    stage_mma<S, 0>(w0, last, acc, wave, lane);
    if constexpr (DIAG) {
      asm volatile("" : "+v"(acc[0][0]));
    }
    TT_STAMP(4)
    const size_t bt = b * T + t;
    const float xt = in1 ? xq.at(t) : 0.f;
#pragma unroll
    for (int x = 0; x < T0::XM; ++x)
#pragma unroll
      for (int y = 0; y < T0::YR; ++y) {
        const int hd = hid[x][y];
        if (in1) {      // W_in x_t + b_in from the two unit rows (GinSrc)
#pragma unroll
          for (int g = 0; g < NG; ++g) gi[x][y][g] = bb[x][y][g] + xt * vv[x][y][g];
        }
        if constexpr (PAIR) {
          // lanes 0-31: (i, g); lanes 32-63: (f, o).  u = sigmoid(first), v = tanh|sigmoid(second)
          const float u = fsigmoid(acc[x][y][0] + gi[x][y][0] + bh[x][y][0]);                 // lstm.py:26-27
          const float a1 = acc[x][y][1] + gi[x][y][1] + bh[x][y][1];
          const float v = sgn * fsigmoid(sgn * a1) + (1.0f - sgn);                           // lstm.py:28-29
          const float prod = u * v;                                                          // i*g on lanes 0-31
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(prod), __float_as_uint(prod), false, false);
          const float ig_g = __uint_as_float(sw[0]);                                         // lanes 32-63 <- lanes 0-31
          const float cy = u * cst[x][y] + ig_g;                                             // lstm.py:31
          const float hy = round_storage<TS>(v * ftanh(cy));                                 // lstm.py:32
          if (reserve && ok[x][y]) {
            float* rv = reserve + res_gate(bt, H, hd) + 2 * pair;                            // i,g | f,o ; c
            rv[0] = u; rv[1] = v;
            if (pair) reserve[res_cell((size_t)B * T, bt, H, hd)] = cy;
          }
          if (ok[x][y] && pair) {
            cst[x][y] = cy;
            hst[x][y] = hy;
            hbuf[a_off<SL::K>(hd / SL::K, hd % SL::K)] = hy;
          }
          if (!in1 && ok[x][y] && t + 1 < T) {
            const f32x2 nx = *reinterpret_cast<const f32x2*>(gin + ((bt + 1) * H + hd) * 4 + 2 * pair);
            gi[x][y][0] = nx[0]; gi[x][y][1] = nx[1];
          }
        } else if (ok[x][y]) {
          const float ig = fsigmoid(acc[x][y][0] + gi[x][y][0] + bh[x][y][0]);     // lstm.py:26
          const float gg = ftanh(acc[x][y][2] + gi[x][y][1] + bh[x][y][1]);        // lstm.py:28
          const float fg = fsigmoid(acc[x][y][1] + gi[x][y][2] + bh[x][y][2]);     // lstm.py:27
          const float og = fsigmoid(acc[x][y][3] + gi[x][y][3] + bh[x][y][3]);     // lstm.py:29
          const float cy = fg * cst[x][y] + ig * gg;                               // lstm.py:31
          const float hy = round_storage<TS>(og * ftanh(cy));                      // lstm.py:32
          cst[x][y] = cy;
          hst[x][y] = hy;
          hbuf[a_off<SL::K>(hd / SL::K, hd % SL::K)] = hy;
          if (reserve) {
            float* rv = reserve + res_gate(bt, H, hd);
            rv[0] = ig; rv[1] = gg; rv[2] = fg; rv[3] = og;
            reserve[res_cell((size_t)B * T, bt, H, hd)] = cy;
          }
          if (!in1 && t + 1 < T) {
            const f32x4 nx = *reinterpret_cast<const f32x4*>(gin + ((bt + 1) * H + hd) * 4);
            gi[x][y][0] = nx[0]; gi[x][y][1] = nx[1]; gi[x][y][2] = nx[2]; gi[x][y][3] = nx[3];
          }
        }
      }
    if (in1) xq.advance(xs, b * T, T, t, lane);
    TT_STAMP(5)
    lds_barrier();
    TT_STAMP(6)
    // outputs[:, t, :] = h_t (lstm.py:133): h_t now sits complete in the LDS image the next step reads; the last
    // wave streams it out as whole 16-byte pieces per lane (one coalesced store per timestep instead of eight
    // half-empty ones).  The image is not overwritten before the next gate phase, three barriers away.
    // (out == NULL: the caller consumes only the final state — ttrnn_rnn_out_optional)
    if (wave == FAST_NW - 1 && out) {
#pragma unroll
      for (int h4 = lane; h4 < H / 4; h4 += 64) {
        const int hd0 = 4 * h4;
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hbuf + a_off<SL::K>(hd0 / SL::K, hd0 % SL::K));
        store4(out, bt * H + hd0, hv);
      }
    }
  }
#pragma unroll
  for (int x = 0; x < T0::XM; ++x)
#pragma unroll
    for (int y = 0; y < T0::YR; ++y)
      if (ok[x][y] && pair) {
        if (hT) st(hT, b * H + hid[x][y], hst[x][y]);
        if (cT) st(cT, b * H + hid[x][y], cst[x][y]);
      }
  if constexpr (DIAG) {
    // stamps leave through a buffer nothing else reads: the caller passes a scratch `reserve`
    if (lane == 0 && reserve && b < 8) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (b * FAST_NW + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
}

// ---- dispatch ------------------------------------------------------------------------------------------
template <class S, int CELL, typename TS>
static int launch_one_t(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                        const void* bias_hid, void* out, void* hT, void* cT, float* reserve, hipStream_t stream) {
  static_assert(shape_ok_recurrent<S>(), "shape not supported by the MFMA path");
  const TS* bh = rs.has_bias_hid ? (const TS*)bias_hid : (const TS*)nullptr;
  if constexpr (CELL == TTRNN_LSTM && lstm_fusable<S>()) {
    const bool diag_on = opt(OPT_DIAG) != 0;
    if (diag_on && reserve)   // diagnostic build: phase stamps overwrite the reserve buffer
      hipLaunchKernelGGL((k_lstm_fwd_fused<S, true, TS>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin,
                         (const TS*)h0, (const TS*)c0, packed_hid, bh, (TS*)out, (TS*)hT, (TS*)cT, reserve);
    else
      hipLaunchKernelGGL((k_lstm_fwd_fused<S, false, TS>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin,
                         (const TS*)h0, (const TS*)c0, packed_hid, bh, (TS*)out, (TS*)hT, (TS*)cT, reserve);
    return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((k_rnn_fwd_fast<S, CELL, TS>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin,
                     (const TS*)h0, (const TS*)c0, packed_hid, bh, (TS*)out, (TS*)hT, (TS*)cT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

template <class S, int CELL>
static int launch_one(const RnnShape& rs, int dtype, GinSrc gin, const void* h0, const void* c0,
                      const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT, float* reserve,
                      hipStream_t stream) {
  return dtype == TTRNN_F32
             ? launch_one_t<S, CELL, float>(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, stream)
             : launch_one_t<S, CELL, bf16_t>(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, stream);
}

bool fast_rnn_fwd_available(const RnnShape& rs, int dtype) {
  if ((dtype != TTRNN_F32 && dtype != TTRNN_BF16) || rs.B < 1 || rs.T < 1) return false;
  if (rs.cell == TTRNN_LSTM)
    return shape_matches<ShpH256R8L>(rs.hid_s) || shape_matches<ShpH256R16L>(rs.hid_s) ||
           shape_matches<ShpH128R4L>(rs.hid_s);
  return shape_matches<ShpH256R8G>(rs.hid_s) || shape_matches<ShpH256R16G>(rs.hid_s);
}

int launch_rnn_fwd_fast(const RnnShape& rs, int dtype, GinSrc gin, const void* h0, const void* c0,
                        const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT, float* reserve,
                        hipStream_t stream) {
#define TT_TRY(SHAPE, CELL)                                                                         \
  if (rs.cell == CELL && shape_matches<SHAPE>(rs.hid_s))                                            \
    return launch_one<SHAPE, CELL>(rs, dtype, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, stream)
  TT_TRY(ShpH256R8L, TTRNN_LSTM);
  TT_TRY(ShpH256R16L, TTRNN_LSTM);
  TT_TRY(ShpH128R4L, TTRNN_LSTM);
  TT_TRY(ShpH256R8G, TTRNN_GRU);
  TT_TRY(ShpH256R16G, TTRNN_GRU);
#undef TT_TRY
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
