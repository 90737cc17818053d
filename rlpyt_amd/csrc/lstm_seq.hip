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
                                                            float* __restrict__ h_out, int B) {
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
  const float c1 = sigmoid_s(gf) * c_old + sigmoid_s(gi) * tanhf(gg);
  const float h1 = sigmoid_s(go) * tanhf(c1);
  c[(int64_t)cb * H + cu] = c1;
  h_out[(int64_t)cb * H + cu] = h1;
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
      RL_LAUNCH((rlpyt::lstm_seq_step_kernel<8>), grid, dim3(256), 0, s, xp, w_hh, hp, c, out + t * BH, B);
    else
      RL_LAUNCH((rlpyt::lstm_seq_step_kernel<4>), grid, dim3(256), 0, s, xp, w_hh, hp, c, out + t * BH, B);
    RL_LAUNCH_CHECK();
  }
  return RLPYT_OK;
}
