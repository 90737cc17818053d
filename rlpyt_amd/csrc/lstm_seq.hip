// No-grad forward of a single-layer LSTM over a whole sequence: the target-network, warm-up and
// double-DQN passes of R2D1's update (rlpyt/algos/dqn/r2d1.py:199-224 -- `agent.target(...)`,
// the `torch.no_grad()` warm-up of both networks, `agent(*target_inputs)` for the double-DQN
// argmax -- each through torch.nn.LSTM in rlpyt/models/dqn/atari_r2d1_model.py:61-63): 250 of
// the 335 LSTM time steps of one update of BASELINE config #5 (sequences [40 + 80 + 5, 64]).
//
// Through the library RNN a time step was ~5 launches (recurrent GEMM 11 us, gate kernel, copies,
// ~30 us in all; profiles/r5_r2d1_region.txt).  Here the input projection x W_ih^T + b_ih + b_hh of
// ALL time steps is one library GEMM done by the caller, and a time step is ONE launch:
//   workgroup = (4 hidden units, 32 sequences); its 16 rows of W_hh -- row i = (unit i >> 2,
//   gate i & 3), so that after v_mfma_f32_16x16x4_f32 a lane holds all four gates of one (unit,
//   sequence) pair -- are the A operand, h_{t-1} of its 32 sequences the B operand, K = H split
//   over the 4 waves (all 24 16-byte loads of a wave in flight at once), partial tiles summed through
//   LDS in a fixed order, then the cell arithmetic of lstm_cell_kernel (step.hip):
//   c' = sigmoid(f) c + sigmoid(i) tanh(g), h' = sigmoid(o) tanh(c'); c is updated in place (every
//   (sequence, unit) pair has one owner), h_t goes to out[t] and is the next launch's B operand --
//   the kernel boundary is the grid-wide barrier.  W_hh (4 MB at H = 512) stays in the L2s: block ->
//   XCD assignment is the same in every launch.
//
// Round 6: the same sequence UNDER AUTOGRAD (the online network's 85-step training pass of the update,
// rlpyt/algos/dqn/r2d1.py:286-334 through atari_r2d1_model.py:61-63; the library RNN spent ~6.6 ms of a
// 24 ms update there: two GEMM launches + a gate kernel per step forward, three launches per step back).
//   forward  = the same step kernel, which also stores the activated gates (i, f, g, o as one float4 per
//              (sequence, unit)) and c_t of every step;
//   backward = ONE launch per step, last to first (lstm_seq_bwd_step_kernel): a workgroup owns 16 hidden
//              units x 16 sequences; phase A is the recurrent gradient dh_rec = dgates_{t+1} W_hh for its
//              units (A operand = 16 rows of W_hh^T, contraction over the 4H gate columns split over 8
//              waves, all loads of a wave in flight at once), phase B the cell's pointwise backward for
//              its 256 (sequence, unit) pairs -> dgates_t [B, 4H] in the module's gate order, dc in place.
//              The kernel boundary is the grid-wide barrier between dgates_{t+1} and its consumers.
//   The four large products (input projection, dx, dW_ih, dW_hh) and the bias sums run once over all
//   time steps in the caller (library GEMMs on [T B, .] matrices).
#include "common.h"

namespace rlpyt {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_s(float x) { return 1.f / (1.f + expf(-x)); }

template <int HG>   // H = 64 * HG
__global__ __launch_bounds__(256) void lstm_seq_step_kernel(const float* __restrict__ xproj,
                                                            const float* __restrict__ w_hh,
                                                            const float* __restrict__ h_prev,
                                                            float* __restrict__ c,
                                                            float* __restrict__ h_out, int B,
                                                            float* __restrict__ gates_out,
                                                            float* __restrict__ c_out) {
  constexpr int H = 64 * HG;
  __shared__ f32x4 red[4][2][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int u0 = blockIdx.x * 4, b0 = blockIdx.y * 32;
  const int k0 = wave * (H / 4) + 4 * kq;
  const float* __restrict__ wrow = w_hh + (int64_t)((j & 3) * H + u0 + (j >> 2)) * H + k0;
  const float* __restrict__ hr0 = h_prev + (int64_t)min(b0 + j, B - 1) * H + k0;
  const float* __restrict__ hr1 = h_prev + (int64_t)min(b0 + 16 + j, B - 1) * H + k0;
  // the cell's own inputs (threads 0..127: tile = tid >> 6, sequence j, unit kq), requested now
  const int cb = min(b0 + 16 * (tid >> 6 & 1) + j, B - 1), cu = u0 + kq;
  float xg[4], c_old;
#pragma unroll
  for (int r = 0; r < 4; ++r) xg[r] = xproj[(int64_t)cb * (4 * H) + r * H + cu];
  c_old = c[(int64_t)cb * H + cu];
  f32x4 a[HG], v0[HG], v1[HG];
#pragma unroll
  for (int g = 0; g < HG; ++g) {
    a[g] = *reinterpret_cast<const f32x4*>(wrow + 16 * g);
    v0[g] = *reinterpret_cast<const f32x4*>(hr0 + 16 * g);
    v1[g] = *reinterpret_cast<const f32x4*>(hr1 + 16 * g);
  }
  // every load above is issued before the first MFMA (left alone, hipcc keeps two or three of the 24
  // row segments in flight and pays a latency trip per batch of them, and sinks the cell's inputs
  // into the branch behind the MFMAs: one more trip); the empty asm is the use that keeps them here
  asm volatile("" ::"v"(xg[0]), "v"(xg[1]), "v"(xg[2]), "v"(xg[3]), "v"(c_old));
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < HG; ++g) {
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][sp], v0[g][sp], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][sp], v1[g][sp], acc1, 0, 0, 0);
    }
  }
  // D[row i = 4 kq + r][column j]: lane (j, kq) holds gates r = i, f, g, o of unit kq, sequence j
  red[wave][0][lane] = acc0;
  red[wave][1][lane] = acc1;
  __syncthreads();
  if (tid >= 128) return;
  const int tile = tid >> 6;
  const f32x4 s = ((red[0][tile][lane] + red[1][tile][lane]) + red[2][tile][lane]) + red[3][tile][lane];
  if (b0 + 16 * tile + j >= B) return;
  const float gi = s[0] + xg[0], gf = s[1] + xg[1], gg = s[2] + xg[2], go = s[3] + xg[3];
  const float si = sigmoid_s(gi), sf = sigmoid_s(gf), tg = tanhf(gg), so = sigmoid_s(go);
  const float c1 = sf * c_old + si * tg;
  const float h1 = so * tanhf(c1);
  c[(int64_t)cb * H + cu] = c1;
  h_out[(int64_t)cb * H + cu] = h1;
  if (gates_out != nullptr) {          // training pass: what the backward step needs (uniform branch)
    *reinterpret_cast<f32x4*>(gates_out + ((int64_t)cb * H + cu) * 4) = f32x4{si, sf, tg, so};
    c_out[(int64_t)cb * H + cu] = c1;
  }
}

// One backward time step (see the header): grid = (H / 16, ceil(B / 16)), 512 threads.
//   dgn    dgates_{t+1} [B, 4H] (nullptr at the last step: no recurrent gradient yet)
//   w_t    W_hh^T [H, 4H]
//   dout   dL/d out_t [B, H] (nullable), dhn: extra gradient into h_T at the last step (nullable)
//   gates  activated gates of step t [B, H, 4], c_t, c_prev [B, H]
//   dc     [B, H] in / out: dL/dc_t in, dL/dc_{t-1} out
//   dg     dgates_t [B, 4H] out (nullptr: the launch after step 0 -- only dh_rec -> dh0)
template <int HG>   // H = 64 * HG
__global__ __launch_bounds__(512) void lstm_seq_bwd_step_kernel(
    const float* __restrict__ dgn, const float* __restrict__ w_t, const float* __restrict__ dout,
    const float* __restrict__ dhn, const float* __restrict__ gates, const float* __restrict__ c_t,
    const float* __restrict__ c_prev, float* __restrict__ dc, float* __restrict__ dg,
    float* __restrict__ dh0, int B) {
  constexpr int H = 64 * HG, G4 = 4 * H;
  constexpr int NG = G4 / 8 / 16;                     // 16-column groups of a wave's contraction slice
  __shared__ f32x4 red[8][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int u0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  // phase B's own inputs (threads 0..255: pair (unit 4 kq + r, sequence j), r = tid >> 6), requested now
  const int r = tid >> 6 & 3;
  const int pb = min(b0 + j, B - 1), pu = u0 + 4 * kq + r;
  const int64_t pi = (int64_t)pb * H + pu;
  f32x4 gt = {0.f, 0.f, 0.f, 0.f};
  float ct = 0.f, cp = 0.f, dcv = 0.f, dh = 0.f;
  if (tid < 256) {
    if (dg != nullptr) {
      gt = *reinterpret_cast<const f32x4*>(gates + pi * 4);
      ct = c_t[pi];
      cp = c_prev[pi];
      dcv = dc[pi];
    }
    if (dout != nullptr) dh = dout[pi];
    if (dhn != nullptr) dh += dhn[pi];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (dgn != nullptr) {                               // uniform
    const int n0 = wave * (G4 / 8) + 4 * kq;
    const float* __restrict__ wrow = w_t + (int64_t)(u0 + j) * G4 + n0;
    const float* __restrict__ grow = dgn + (int64_t)min(b0 + j, B - 1) * G4 + n0;
    f32x4 a[NG], v[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      a[g] = *reinterpret_cast<const f32x4*>(wrow + 16 * g);
      v[g] = *reinterpret_cast<const f32x4*>(grow + 16 * g);
    }
    asm volatile("" ::"v"(gt[0]), "v"(gt[1]), "v"(gt[2]), "v"(gt[3]), "v"(ct), "v"(cp), "v"(dcv), "v"(dh));
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][sp], v[g][sp], acc, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g + 1][sp], v[g + 1][sp], acc_b, 0, 0, 0);
      }
    }
    acc += acc_b;
  }
  // D[row = unit 4 kq + r'][column = sequence j]: lane (j, kq) holds units 4 kq .. 4 kq + 3 of sequence j
  red[wave][lane] = acc;
  __syncthreads();
  if (tid >= 256 || b0 + j >= B) return;
  float rec = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) rec += red[w][lane][r];      // fixed order
  dh += rec;
  if (dg == nullptr) {
    if (dh0 != nullptr) dh0[pi] = dh;
    return;
  }
  const float gi = gt[0], gf = gt[1], gg = gt[2], go = gt[3];
  const float tc = tanhf(ct);
  const float dcc = dcv + dh * go * (1.f - tc * tc);
  float* __restrict__ row = dg + (int64_t)pb * G4 + pu;
  row[0] = dcc * gg * gi * (1.f - gi);
  row[H] = dcc * cp * gf * (1.f - gf);
  row[2 * H] = dcc * gi * (1.f - gg * gg);
  row[3 * H] = dh * tc * go * (1.f - go);
  dc[pi] = dcc * gf;
}

}  // namespace
}  // namespace rlpyt

extern "C" int rlpyt_lstm_seq_f32(const float* xproj, const float* w_hh, const float* h0, float* c,
                                  float* out, int T, int B, int H, rlpyt_stream_t stream) {
  RL_CHECK_ARG(T >= 0 && B > 0, RLPYT_EINVAL, "rlpyt_lstm_seq_f32: bad sizes");
  if (T == 0) return RLPYT_OK;
  RL_CHECK_ARG(xproj && w_hh && h0 && c && out, RLPYT_EINVAL, "rlpyt_lstm_seq_f32: null pointer");
  RL_CHECK_ARG(H == 256 || H == 512, RLPYT_ESHAPE, "rlpyt_lstm_seq_f32: H must be 256 or 512 (H=%d)", H);
  RL_CHECK_ARG(RL_ALIGNED16(w_hh) && RL_ALIGNED16(h0) && RL_ALIGNED16(out), RLPYT_ESHAPE,
               "rlpyt_lstm_seq_f32: w_hh / h0 / out must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(H / 4), (unsigned)((B + 31) / 32));
  const int64_t BH = (int64_t)B * H;
  for (int t = 0; t < T; ++t) {
    const float* hp = t == 0 ? h0 : out + (int64_t)(t - 1) * BH;
    const float* xp = xproj + (int64_t)t * B * 4 * H;
    if (H == 512)
      RL_LAUNCH((rlpyt::lstm_seq_step_kernel<8>), grid, dim3(256), 0, s, xp, w_hh, hp, c, out + t * BH, B,
                (float*)nullptr, (float*)nullptr);
    else
      RL_LAUNCH((rlpyt::lstm_seq_step_kernel<4>), grid, dim3(256), 0, s, xp, w_hh, hp, c, out + t * BH, B,
                (float*)nullptr, (float*)nullptr);
    RL_LAUNCH_CHECK();
  }
  return RLPYT_OK;
}

extern "C" int rlpyt_lstm_seq_train_f32(const float* xproj, const float* w_hh, const float* h0, float* c,
                                        float* out, float* gates, float* c_all, int T, int B, int H,
                                        rlpyt_stream_t stream) {
  RL_CHECK_ARG(T >= 0 && B > 0, RLPYT_EINVAL, "rlpyt_lstm_seq_train_f32: bad sizes");
  if (T == 0) return RLPYT_OK;
  RL_CHECK_ARG(xproj && w_hh && h0 && c && out && gates && c_all, RLPYT_EINVAL,
               "rlpyt_lstm_seq_train_f32: null pointer");
  RL_CHECK_ARG(H == 256 || H == 512, RLPYT_ESHAPE, "rlpyt_lstm_seq_train_f32: H must be 256 or 512 (H=%d)", H);
  RL_CHECK_ARG(RL_ALIGNED16(w_hh) && RL_ALIGNED16(h0) && RL_ALIGNED16(out) && RL_ALIGNED16(gates),
               RLPYT_ESHAPE, "rlpyt_lstm_seq_train_f32: w_hh / h0 / out / gates must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(H / 4), (unsigned)((B + 31) / 32));
  const int64_t BH = (int64_t)B * H;
  for (int t = 0; t < T; ++t) {
    const float* hp = t == 0 ? h0 : out + (int64_t)(t - 1) * BH;
    const float* xp = xproj + (int64_t)t * B * 4 * H;
    if (H == 512)
      RL_LAUNCH((rlpyt::lstm_seq_step_kernel<8>), grid, dim3(256), 0, s, xp, w_hh, hp, c, out + t * BH, B,
                gates + 4 * t * BH, c_all + t * BH);
    else
      RL_LAUNCH((rlpyt::lstm_seq_step_kernel<4>), grid, dim3(256), 0, s, xp, w_hh, hp, c, out + t * BH, B,
                gates + 4 * t * BH, c_all + t * BH);
    RL_LAUNCH_CHECK();
  }
  return RLPYT_OK;
}

extern "C" int rlpyt_lstm_seq_bwd_f32(const float* dout, const float* dhn, const float* gates,
                                      const float* c_all, const float* c0, const float* w_hh_t, float* dc,
                                      float* dgates, float* dh0, int T, int B, int H,
                                      rlpyt_stream_t stream) {
  RL_CHECK_ARG(T >= 0 && B > 0, RLPYT_EINVAL, "rlpyt_lstm_seq_bwd_f32: bad sizes");
  if (T == 0) return RLPYT_OK;
  RL_CHECK_ARG(gates && c_all && c0 && w_hh_t && dc && dgates, RLPYT_EINVAL,
               "rlpyt_lstm_seq_bwd_f32: null pointer");
  RL_CHECK_ARG(H == 256 || H == 512, RLPYT_ESHAPE, "rlpyt_lstm_seq_bwd_f32: H must be 256 or 512 (H=%d)", H);
  RL_CHECK_ARG(RL_ALIGNED16(w_hh_t) && RL_ALIGNED16(gates) && RL_ALIGNED16(dgates), RLPYT_ESHAPE,
               "rlpyt_lstm_seq_bwd_f32: w_hh_t / gates / dgates must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(H / 16), (unsigned)((B + 15) / 16));
  const int64_t BH = (int64_t)B * H;
  // t = T - 1 .. 0, then one launch for dh0 (skipped when the caller does not want it)
  for (int t = T - 1; t >= (dh0 != nullptr ? -1 : 0); --t) {
    const float* dgn = t == T - 1 ? nullptr : dgates + (int64_t)(t + 1) * 4 * BH;
    const float* dout_t = (t >= 0 && dout != nullptr) ? dout + (int64_t)t * BH : nullptr;
    const float* dhn_t = t == T - 1 ? dhn : nullptr;
    const float* g_t = t >= 0 ? gates + (int64_t)t * 4 * BH : nullptr;
    const float* c_t = t >= 0 ? c_all + (int64_t)t * BH : nullptr;
    const float* c_p = t > 0 ? c_all + (int64_t)(t - 1) * BH : c0;
    float* dg_t = t >= 0 ? dgates + (int64_t)t * 4 * BH : nullptr;
    if (H == 512)
      RL_LAUNCH((rlpyt::lstm_seq_bwd_step_kernel<8>), grid, dim3(512), 0, s, dgn, w_hh_t, dout_t, dhn_t, g_t,
                c_t, c_p, dc, dg_t, dh0, B);
    else
      RL_LAUNCH((rlpyt::lstm_seq_bwd_step_kernel<4>), grid, dim3(512), 0, s, dgn, w_hh_t, dout_t, dhn_t, g_t,
                c_t, c_p, dc, dg_t, dh0, B);
    RL_LAUNCH_CHECK();
  }
  return RLPYT_OK;
}
