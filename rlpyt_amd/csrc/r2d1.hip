// R2D1 sequence TD loss (forward + backward) and sequence priorities for gfx950.
// Reference: rlpyt/algos/dqn/r2d1.py:298-345 -- value rescaling h / h^-1, (double-)max
// target, squared or Huber loss with per-sequence importance weights, masked mean over
// [T,B], and priorities eta*max_t + (1-eta)*mean_t of the masked |TD| per sequence.
// Layout [T, B, A] / [T, B]; one workgroup per column b (lanes run over time), so the
// per-sequence max/mean are workgroup reductions; the global masked mean is finished by a
// one-workgroup kernel that also scales the gradients' normaliser.
#include "common.h"

namespace rlpyt {
namespace {

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }
// r2d1.py:336-339
__device__ __forceinline__ float value_scale(float x, float eps) {
  return sgn(x) * (sqrtf(fabsf(x) + 1.f) - 1.f) + eps * x;
}
// r2d1.py:341-345
__device__ __forceinline__ float inv_value_scale(float z, float eps) {
  const float t = (sqrtf(1.f + 4.f * eps * (fabsf(z) + 1.f + eps)) - 1.f) / (2.f * eps);
  return sgn(z) * (t * t - 1.f);
}

struct SeqWs {
  double loss_num[4096];
  double valid_sum[4096];
};

// grid = B workgroups; block = 256 lanes striding over T.
__global__ __launch_bounds__(256) void r2d1_column_kernel(
    const float* __restrict__ qs, const float* __restrict__ target_qs,
    const float* __restrict__ next_qs, const int64_t* __restrict__ action,
    const float* __restrict__ return_, const uint8_t* __restrict__ done_n,
    const float* __restrict__ valid, const float* __restrict__ is_weights, int T, int B, int A,
    float disc_n, float delta_clip, float eps, float pri_eta, float* __restrict__ td_valid,
    float* __restrict__ priorities, float* __restrict__ grad_qs, SeqWs* __restrict__ ws) {
  __shared__ double scratch[3 * 16];
  __shared__ float smax[16];
  const int b = blockIdx.x;
  const float isw = is_weights ? is_weights[b] : 1.f;
  double acc[3] = {0, 0, 0};  // sum loss*valid, sum valid, sum td*valid
  float mx = 0.f;             // max over t of td*valid (values are >= 0)
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const int64_t i = (int64_t)t * B + b;
    const float* q_row = qs + i * A;
    const float* tq_row = target_qs + i * A;
    const int a = (int)action[i];
    float tq;
    if (next_qs != nullptr) {
      const float* nq = next_qs + i * A;
      int best = 0;
      float bvv = nq[0];
      for (int j = 1; j < A; ++j)
        if (nq[j] > bvv) { bvv = nq[j]; best = j; }
      tq = tq_row[best];
    } else {
      tq = tq_row[0];
      for (int j = 1; j < A; ++j) tq = fmaxf(tq, tq_row[j]);
    }
    const float nd = 1.f - (float)(done_n[i] ? 1 : 0);
    const float y = value_scale(return_[i] + nd * disc_n * inv_value_scale(tq, eps), eps);
    const float delta = y - q_row[a];
    const float ad = fabsf(delta);
    float loss = 0.5f * delta * delta, dl = delta;
    if (delta_clip > 0.f && !(ad <= delta_clip)) {
      loss = delta_clip * (ad - delta_clip / 2);
      dl = delta_clip * sgn(delta);
    }
    loss *= isw;
    const float v = valid[i];
    const float td = delta_clip > 0.f ? fminf(ad, delta_clip) : ad;
    const float tdv = td * v;
    td_valid[i] = tdv;
    // un-normalised gradient; the finalize kernel's normaliser is applied by grad_scale
    for (int j = 0; j < A; ++j) grad_qs[i * A + j] = (j == a) ? (-v * isw * dl) : 0.f;
    acc[0] += (double)(loss * v);
    acc[1] += (double)v;
    acc[2] += (double)tdv;
    mx = fmaxf(mx, tdv);
  }
  block_sum<3>(acc, scratch);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = smax[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, smax[w]);
    const float mean_d = (float)(acc[2] / acc[1]);  // valid_mean(td, valid, dim=0)
    priorities[b] = pri_eta * m + (1.f - pri_eta) * mean_d;
    ws->loss_num[b] = acc[0];
    ws->valid_sum[b] = acc[1];
  }
}

__global__ __launch_bounds__(256) void r2d1_finalize_kernel(SeqWs* __restrict__ ws, int B,
                                                            float* __restrict__ out) {
  __shared__ double scratch[2 * 16];
  double acc[2] = {0, 0};
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    acc[0] += ws->loss_num[b];
    acc[1] += ws->valid_sum[b];
  }
  block_sum<2>(acc, scratch);
  if (threadIdx.x == 0) {
    out[0] = (float)(acc[0] / acc[1]);
    out[1] = (float)(1.0 / acc[1]);  // gradient normaliser
  }
}

__global__ __launch_bounds__(256) void scale_by_device_scalar_kernel(float* __restrict__ x,
                                                                     int64_t n,
                                                                     const float* __restrict__ s) {
  const float k = s[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    x[i] *= k;
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

extern "C" int64_t rlpyt_r2d1_loss_workspace_bytes(void) { return (int64_t)sizeof(SeqWs); }

extern "C" int rlpyt_r2d1_loss_fwd_bwd_f32(
    const float* qs, const float* target_qs, const float* next_qs, const int64_t* action,
    const float* return_, const uint8_t* done_n, const float* valid, const float* is_weights,
    int T, int B, int A, float disc_n, float delta_clip, float value_scale_eps, float pri_eta,
    float* out_scalars, float* td_abs_valid, float* priorities, float* grad_qs, void* workspace,
    rlpyt_stream_t stream) {
  RL_CHECK_ARG(qs && target_qs && action && return_ && done_n && valid && out_scalars &&
                   td_abs_valid && priorities && grad_qs && workspace,
               RLPYT_EINVAL, "rlpyt_r2d1_loss_fwd_bwd_f32: null pointer");
  RL_CHECK_ARG(T > 0 && B > 0 && B <= 4096 && A > 0, RLPYT_ESHAPE,
               "rlpyt_r2d1_loss_fwd_bwd_f32: need T>0, 0<B<=4096, A>0");
  RL_CHECK_ARG(value_scale_eps > 0.f, RLPYT_EINVAL,
               "rlpyt_r2d1_loss_fwd_bwd_f32: value_scale_eps must be > 0");
  hipStream_t s = (hipStream_t)stream;
  SeqWs* ws = (SeqWs*)workspace;
  RL_LAUNCH(r2d1_column_kernel, dim3(B), dim3(256), 0, s, qs, target_qs, next_qs,
                     action, return_, done_n, valid, is_weights, T, B, A, disc_n, delta_clip,
                     value_scale_eps, pri_eta, td_abs_valid, priorities, grad_qs, ws);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(r2d1_finalize_kernel, dim3(1), dim3(256), 0, s, ws, B, out_scalars);
  RL_LAUNCH_CHECK();
  const int64_t n = (int64_t)T * B * A;
  RL_LAUNCH(scale_by_device_scalar_kernel,
                     dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 2048)), dim3(256), 0, s,
                     grad_qs, n, out_scalars + 1);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
