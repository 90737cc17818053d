// Observation running-mean/std kernels for gfx950 (MuJoCo-style normalisation).
// Reference: rlpyt/models/running_mean_std.py:21-45 (update: batch mean / biased var over the
// T*B leading rows, optional all-reduce average, Chan parallel merge into running
// mean/var/count) and rlpyt/models/pg/mujoco_ff_model.py:68-73 (normalise + clip).
// [n, D] row-major: lanes run along D (coalesced), rows are strided over workgroups.
#include "common.h"

namespace rlpyt {
namespace {

constexpr int kRmsRowsPerBlock = 256;

// partial[blockIdx.y][0/1][d] = sum x, sum x^2 over this block's rows (f64)
__global__ __launch_bounds__(256) void rms_partial_kernel(const float* __restrict__ x,
                                                          int64_t n, int64_t D,
                                                          double* __restrict__ partial) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const int64_t r0 = (int64_t)blockIdx.y * kRmsRowsPerBlock;
  const int64_t r1 = min(n, r0 + kRmsRowsPerBlock);
  double s = 0.0, ss = 0.0;
  for (int64_t r = r0; r < r1; ++r) {
    const double v = (double)x[r * D + d];
    s += v;
    ss += v * v;
  }
  partial[((int64_t)blockIdx.y * 2 + 0) * D + d] = s;
  partial[((int64_t)blockIdx.y * 2 + 1) * D + d] = ss;
}

__global__ __launch_bounds__(256) void rms_finalize_kernel(const double* __restrict__ partial,
                                                           int n_part, int64_t n, int64_t D,
                                                           float* __restrict__ mean,
                                                           float* __restrict__ var) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  double s = 0.0, ss = 0.0;
  for (int p = 0; p < n_part; ++p) {
    s += partial[((int64_t)p * 2 + 0) * D + d];
    ss += partial[((int64_t)p * 2 + 1) * D + d];
  }
  const double m = s / (double)n;
  double v = ss / (double)n - m * m;  // biased variance (unbiased=False)
  if (v < 0.0) v = 0.0;
  mean[d] = (float)m;
  var[d] = (float)v;
}

// Chan merge (running_mean_std.py:34-44); count stays a device scalar.
__global__ __launch_bounds__(256) void rms_merge_kernel(float* __restrict__ mean,
                                                        float* __restrict__ var,
                                                        const float* __restrict__ count,
                                                        const float* __restrict__ bmean,
                                                        const float* __restrict__ bvar,
                                                        float bcount, int64_t D) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const float c = count[0];
  if (c == 0.f) {
    mean[d] = bmean[d];
    var[d] = bvar[d];
  } else {
    const float delta = bmean[d] - mean[d];
    const float total = c + bcount;
    const float m_a = var[d] * c;
    const float m_b = bvar[d] * bcount;
    const float M2 = m_a + m_b + delta * delta * c * bcount / total;
    mean[d] = mean[d] + delta * bcount / total;
    var[d] = M2 / total;
  }
}

__global__ void rms_count_kernel(float* __restrict__ count, float bcount) {
  count[0] += bcount;
}

__global__ __launch_bounds__(256) void rms_normalize_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ var,
                                                            float* __restrict__ out, int64_t n,
                                                            int64_t D, float var_clip,
                                                            float obs_clip) {
  const int64_t total = n * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t d = i % D;
    float v = var[d];
    if (var_clip > 0.f) v = fmaxf(v, var_clip);
    float y = (x[i] - mean[d]) / sqrtf(v);
    out[i] = fminf(fmaxf(y, -obs_clip), obs_clip);
  }
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

extern "C" int64_t rlpyt_obs_rms_workspace_bytes(int64_t n, int64_t D) {
  return ceil_div(n, kRmsRowsPerBlock) * 2 * D * (int64_t)sizeof(double);
}

extern "C" int rlpyt_obs_batch_stats_f32(const float* x, int64_t n, int64_t D, float* mean,
                                         float* var, void* workspace, rlpyt_stream_t stream) {
  RL_CHECK_ARG(n > 0 && D > 0, RLPYT_EINVAL, "rlpyt_obs_batch_stats_f32: need n>0, D>0");
  RL_CHECK_ARG(x && mean && var && workspace, RLPYT_EINVAL,
               "rlpyt_obs_batch_stats_f32: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int n_part = (int)ceil_div(n, kRmsRowsPerBlock);
  RL_CHECK_ARG(n_part <= 65535, RLPYT_ESHAPE, "rlpyt_obs_batch_stats_f32: n too large");
  const unsigned gx = (unsigned)ceil_div(D, 256);
  RL_LAUNCH(rms_partial_kernel, dim3(gx, n_part), dim3(256), 0, s, x, n, D,
                     (double*)workspace);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(rms_finalize_kernel, dim3(gx), dim3(256), 0, s, (const double*)workspace,
                     n_part, n, D, mean, var);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_obs_rms_merge_f32(float* mean, float* var, float* count,
                                       const float* batch_mean, const float* batch_var,
                                       float batch_count, int64_t D, rlpyt_stream_t stream) {
  RL_CHECK_ARG(mean && var && count && batch_mean && batch_var && D > 0, RLPYT_EINVAL,
               "rlpyt_obs_rms_merge_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  RL_LAUNCH(rms_merge_kernel, dim3((unsigned)ceil_div(D, 256)), dim3(256), 0, s, mean,
                     var, count, batch_mean, batch_var, batch_count, D);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(rms_count_kernel, dim3(1), dim3(1), 0, s, count, batch_count);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_obs_normalize_f32(const float* x, const float* mean, const float* var,
                                       float* out, int64_t n, int64_t D, float var_clip,
                                       float obs_clip, rlpyt_stream_t stream) {
  RL_CHECK_ARG(n >= 0 && D > 0, RLPYT_EINVAL, "rlpyt_obs_normalize_f32: bad sizes");
  if (n == 0) return RLPYT_OK;
  RL_CHECK_ARG(x && mean && var && out, RLPYT_EINVAL, "rlpyt_obs_normalize_f32: null pointer");
  const unsigned g = (unsigned)std::min<int64_t>(ceil_div(n * D, 256), 256 * 16);
  RL_LAUNCH(rms_normalize_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, mean,
                     var, out, n, D, var_clip, obs_clip);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
