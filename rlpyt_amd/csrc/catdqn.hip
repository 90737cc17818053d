// Categorical (distributional) DQN loss, forward + backward, for gfx950.
// Reference: rlpyt/algos/dqn/cat_dqn.py:34-93 -- Bellman-shifted atom grid, projection of the
// target distribution onto the fixed grid ([B,P,P'] coefficient tensor in the reference, never
// materialised here), cross-entropy loss with IS weights, KL divergence for the priorities.
//
// Mapping: one wavefront per sample, lane i = atom i (P <= 64; the reference's default is 51).
// The P x P' projection is P' rounds of two cross-lane broadcasts; the per-action expected
// values that pick the greedy next action are wave reductions.  A row of P probabilities is
// one coalesced load.  Loss partials per workgroup in f64 -> one-workgroup finalize
// (deterministic, no atomics).  Built with -ffp-contract=off so `ret + nd * (z * disc)` keeps
// the reference's operation boundaries.
#include <algorithm>

#include "common.h"

namespace rlpyt {
namespace {

constexpr int kCatBlock = 256;
constexpr int kCatWaves = kCatBlock / kWave;
constexpr int kCatMaxGrid = 1024;
constexpr float kEpsCatDqn = 1e-6f;  // cat_dqn.py:8

struct CatWs {
  double denom;                  // sum(valid) or M
  double part[kCatMaxGrid];      // per-workgroup sums of the weighted losses
};

__global__ __launch_bounds__(kCatBlock) void cat_denom_kernel(const float* __restrict__ valid,
                                                              int64_t M, CatWs* __restrict__ ws) {
  __shared__ double scratch[16];
  double acc[1] = {0.0};
  if (valid != nullptr) {
    for (int64_t i = threadIdx.x; i < M; i += blockDim.x) acc[0] += (double)valid[i];
    block_sum<1>(acc, scratch);
  } else {
    acc[0] = (double)M;
  }
  if (threadIdx.x == 0) ws->denom = acc[0];
}

__global__ __launch_bounds__(kCatBlock) void cat_dqn_loss_kernel(
    const float* __restrict__ ps, const float* __restrict__ target_ps,
    const float* __restrict__ next_ps, const int64_t* __restrict__ action,
    const float* __restrict__ return_, const uint8_t* __restrict__ done_n,
    const float* __restrict__ is_weights, const float* __restrict__ valid,
    const float* __restrict__ z, int64_t M, int A, int P, float v_min, float v_max,
    float delta_z, float disc_n, float* __restrict__ kl_div, float* __restrict__ grad_ps,
    CatWs* __restrict__ ws) {
  __shared__ double wsum[kCatWaves];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool on = lane < P;
  const float zi = on ? z[lane] : 0.f;
  const float inv_denom = (float)(1.0 / ws->denom);
  const float* __restrict__ sel = next_ps != nullptr ? next_ps : target_ps;  // cat_dqn.py:63-69
  double acc = 0.0;
  for (int64_t m = (int64_t)blockIdx.x * kCatWaves + wid; m < M;
       m += (int64_t)gridDim.x * kCatWaves) {
    const int a = (int)action[m];
    const float nd = 1.0f - (done_n[m] ? 1.0f : 0.0f);
    const float ret = return_[m];
    // cat_dqn.py:45-48: next_z = clamp(ret + (1-done_n) * (z * disc^n), V_min, V_max)
    float nz = ret + nd * (zi * disc_n);
    nz = fminf(fmaxf(nz, v_min), v_max);
    // greedy next action on expected values (first maximum wins, as torch.argmax)
    int best = 0;
    float best_q = 0.f;
    for (int j = 0; j < A; ++j) {
      const float pj = on ? sel[((size_t)m * A + j) * P + lane] : 0.f;
      const float q = wave_sum(pj * zi);
      if (j == 0 || q > best_q) { best_q = q; best = j; }
    }
    const float tp = on ? target_ps[((size_t)m * A + best) * P + lane] : 0.f;
    // cat_dqn.py:50-57,72: target_p[i] = sum_j tp[j] * clamp(1 - |next_z[j] - z[i]| / dz, 0, 1)
    float t = 0.f;
    for (int j = 0; j < P; ++j) {
      const float nzj = __shfl(nz, j, kWave);
      const float tpj = __shfl(tp, j, kWave);
      float c = 1.0f - fabsf(nzj - zi) / delta_z;
      c = fminf(fmaxf(c, 0.f), 1.f);
      t += tpj * c;
    }
    if (!on) t = 0.f;
    // cat_dqn.py:73-76: p = clamp(ps[a], EPS, 1); losses = -sum(target_p * log p)
    const float p_raw = on ? ps[((size_t)m * A + a) * P + lane] : 1.f;
    const float p = fminf(fmaxf(p_raw, kEpsCatDqn), 1.f);
    const float lp = logf(p);
    float loss = -wave_sum(t * lp);
    const float isw = is_weights != nullptr ? is_weights[m] : 1.f;  // cat_dqn.py:78-79
    loss *= isw;
    // cat_dqn.py:81-84: KL(target || p) with both clamped, result clamped to [EPS, 1/EPS]
    const float tc = fminf(fmaxf(t, kEpsCatDqn), 1.f);
    float kl = wave_sum(on ? tc * (logf(tc) - lp) : 0.f);
    kl = fminf(fmaxf(kl, kEpsCatDqn), 1.0f / kEpsCatDqn);
    const float vm = valid != nullptr ? valid[m] : 1.f;  // cat_dqn.py:86-91
    if (valid != nullptr) kl *= vm;
    if (lane == 0) {
      kl_div[m] = kl;
      acc += (double)(loss * vm);
    }
    // dLoss/dps: through log and the clamp (gradient passes where EPS <= p_raw <= 1)
    const float w = isw * vm * inv_denom;
    const bool pass = p_raw >= kEpsCatDqn && p_raw <= 1.f;
    for (int j = 0; j < A; ++j)
      if (on) grad_ps[((size_t)m * A + j) * P + lane] = (j == a && pass) ? -(t / p) * w : 0.f;
  }
  if (lane == 0) wsum[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < kCatWaves; ++w) s += wsum[w];
    ws->part[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kCatBlock) void cat_dqn_finalize_kernel(
    const CatWs* __restrict__ ws, int n_part, float* __restrict__ out) {
  __shared__ double scratch[16];
  double acc[1] = {0.0};
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) acc[0] += ws->part[i];
  block_sum<1>(acc, scratch);
  if (threadIdx.x == 0) {
    out[0] = (float)(acc[0] / ws->denom);
    out[1] = (float)ws->denom;
  }
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

extern "C" int64_t rlpyt_cat_dqn_loss_workspace_bytes(void) { return (int64_t)sizeof(CatWs); }

extern "C" int rlpyt_cat_dqn_loss_fwd_bwd_f32(
    const float* ps, const float* target_ps, const float* next_ps, const int64_t* action,
    const float* return_, const uint8_t* done_n, const float* is_weights, const float* valid,
    const float* z, int64_t M, int A, int P, float v_min, float v_max, float disc_n,
    float* out_scalars, float* kl_div, float* grad_ps, void* workspace, rlpyt_stream_t stream) {
  RL_CHECK_ARG(ps && target_ps && action && return_ && done_n && z && out_scalars && kl_div &&
                   grad_ps && workspace,
               RLPYT_EINVAL, "rlpyt_cat_dqn_loss_fwd_bwd_f32: null pointer");
  RL_CHECK_ARG(M > 0 && A > 0 && P >= 2 && P <= kWave, RLPYT_ESHAPE,
               "rlpyt_cat_dqn_loss_fwd_bwd_f32: need M>0, A>0, 2<=P<=64 (M=%ld A=%d P=%d)",
               (long)M, A, P);
  RL_CHECK_ARG(v_max > v_min, RLPYT_EINVAL, "rlpyt_cat_dqn_loss_fwd_bwd_f32: V_max <= V_min");
  CatWs* ws = reinterpret_cast<CatWs*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  // cat_dqn.py:42: delta_z in double, used as a float32 divisor by the tensor ops
  const float delta_z = (float)(((double)v_max - (double)v_min) / (double)(P - 1));
  const int grid = (int)std::min<int64_t>(ceil_div(M, kCatWaves), kCatMaxGrid);
  RL_LAUNCH(cat_denom_kernel, dim3(1), dim3(kCatBlock), 0, s, valid, M, ws);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(cat_dqn_loss_kernel, dim3(grid), dim3(kCatBlock), 0, s, ps, target_ps,
                     next_ps, action, return_, done_n, is_weights, valid, z, M, A, P, v_min,
                     v_max, delta_z, disc_n, kl_div, grad_ps, ws);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(cat_dqn_finalize_kernel, dim3(1), dim3(kCatBlock), 0, s, ws, grid,
                     out_scalars);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
