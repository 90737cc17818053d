// Gradient clipping + Adam in two launches for a whole model (multi-tensor, one table by value).
//
// Reference sequence replaced (host side of every minibatch update):
//   rlpyt/algos/pg/ppo.py:100-104, rlpyt/algos/pg/a2c.py:52-56, rlpyt/algos/dqn/dqn.py:176-180:
//     grad_norm = torch.nn.utils.clip_grad_norm_(agent.parameters(), clip_grad_norm)
//     optimizer.step()                      (torch.optim.Adam, rlpyt/algos/pg/base.py:34-37)
// which on the device is ~12 launches (per-tensor norms, stack, norm, clip coefficient, scale,
// multi-tensor Adam) for 7.1 MB of parameters: launch-bound, ~0.3 ms per minibatch.  Here:
//   1. clip_adam_norm_kernel: sum of squares of all gradients, f64 partial per workgroup;
//   2. clip_adam_apply_kernel: every workgroup re-reduces the partials (fixed order, no atomics:
//      deterministic), forms coef = min(1, max_norm / (norm + 1e-6)) exactly as clip_grad_norm_,
//      and applies Adam to its elements with the clipped gradient (the gradient tensors themselves
//      are left untouched: nothing reads them afterwards, zero_grad(set_to_none) follows).
// Adam arithmetic = torch.optim.Adam (amsgrad=False, maximize=False):
//   m = m + (1-b1)(g-m);  v = b2 v + (1-b2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// HBM-bound: 5 arrays read + 3 written = 32 B per parameter (+4 B for the norm pass).
#include "common.h"

namespace rlpyt {
namespace {

struct AdamTable {
  rlpyt_adam_tensor t[RLPYT_ADAM_MAX_TENSORS];
  int n;
  // optional transposed mirror of ONE [rows, cols] tensor (the update trunk's weight: its input
  // gradient GEMM reads W^T, ops._LinearNoBias): tensor `mirror_k` is walked in 32 x 32 tiles and
  // every tile is also written, through LDS, to mirror_pt[cols, rows] -- the new W^T comes out of the
  // launch that makes the new W instead of a 7 MB transposing copy per minibatch.  mirror_k < 0: none.
  int mirror_k, mirror_rows, mirror_cols;
  float* mirror_pt;
};

constexpr int kOptGrid = 512, kOptBlock = 256;

__global__ __launch_bounds__(kOptBlock) void clip_adam_norm_kernel(AdamTable tab,
                                                                   double* __restrict__ partial) {
  __shared__ double scratch[16];
  double acc[1] = {0.0};
  const int64_t gtid = (int64_t)blockIdx.x * kOptBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kOptBlock;
  for (int k = 0; k < tab.n; ++k) {
    const float* __restrict__ g = tab.t[k].g;
    // 16-byte path only where the tensor allows it (gradients may be views at odd offsets of a
    // packed buffer, e.g. the fused head kernel's parameter gradients)
    const int64_t n = tab.t[k].n;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(g) & 15) == 0 ? n >> 2 : 0;
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
    float s = 0.f;
    for (int64_t i = gtid; i < n4; i += gsize) {
      const float4 x = g4[i];
      s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    }
    for (int64_t i = (n4 << 2) + gtid; i < n; i += gsize) s += g[i] * g[i];
    acc[0] += (double)s;
  }
  block_sum<1>(acc, scratch);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(kOptBlock) void clip_adam_apply_kernel(
    AdamTable tab, const double* __restrict__ partial, int n_partial, float lr_over_bc1,
    float inv_sqrt_bc2, float beta1, float beta2, float eps, float weight_decay, float max_norm,
    float* __restrict__ grad_norm_out, const float* __restrict__ hyper,
    int64_t* __restrict__ tick_ctr) {
  // hyper != NULL (captured update graphs): {lr / bc1, 1 / sqrt(bc2)} of THIS update come from
  // device memory (rlpyt_update_tick put them there from a host-computed table), so one captured
  // launch serves every update; tick_ctr: the update counter that table is indexed by, advanced
  // here -- the last launch of an update -- by one thread (nobody reads it in this kernel)
  if (hyper != nullptr) {
    lr_over_bc1 = hyper[0];
    inv_sqrt_bc2 = hyper[1];
  }
  if (tick_ctr != nullptr && blockIdx.x == 0 && threadIdx.x == 0) tick_ctr[0] += 1;
  __shared__ double scratch[16];
  __shared__ float coef_s;
  double acc[1] = {0.0};
  for (int i = threadIdx.x; i < n_partial; i += kOptBlock) acc[0] += partial[i];
  block_sum<1>(acc, scratch);
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(acc[0]);
    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    coef_s = max_norm > 0.f ? fminf(max_norm / (total + 1e-6f), 1.f) : 1.f;
    if (blockIdx.x == 0 && grad_norm_out != nullptr) *grad_norm_out = total;
  }
  __syncthreads();
  const float coef = coef_s;
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  const int64_t gtid = (int64_t)blockIdx.x * kOptBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kOptBlock;
  auto upd = [&](float g, float& p, float& m, float& v) {
    g *= coef;
    if (weight_decay != 0.f) g += weight_decay * p;
    m = m + omb1 * (g - m);
    v = beta2 * v + omb2 * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p -= lr_over_bc1 * (m / denom);
  };
  if (tab.mirror_k >= 0) {
    // 32 x 32 tiles of the mirrored tensor: thread = (row tid >> 3, float4 tid & 7); 8 threads read /
    // write 128 contiguous bytes of a row of W, and after the exchange of a row of W^T
    __shared__ float tile[32][33];
    const rlpyt_adam_tensor t = tab.t[tab.mirror_k];
    const int R = tab.mirror_rows, C = tab.mirror_cols;
    const int tiles_c = C >> 5, n_tiles = (R >> 5) * tiles_c;
    const int tr = threadIdx.x >> 3, tc = threadIdx.x & 7;
    for (int tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
      const int r0 = (tile_id / tiles_c) << 5, c0 = (tile_id % tiles_c) << 5;
      const int64_t e = (int64_t)(r0 + tr) * C + c0 + 4 * tc;
      const float4 g = *reinterpret_cast<const float4*>(t.g + e);
      float4 p = *reinterpret_cast<const float4*>(t.p + e);
      float4 m = *reinterpret_cast<const float4*>(t.m + e);
      float4 v = *reinterpret_cast<const float4*>(t.v + e);
      upd(g.x, p.x, m.x, v.x);
      upd(g.y, p.y, m.y, v.y);
      upd(g.z, p.z, m.z, v.z);
      upd(g.w, p.w, m.w, v.w);
      *reinterpret_cast<float4*>(t.p + e) = p;
      *reinterpret_cast<float4*>(t.m + e) = m;
      *reinterpret_cast<float4*>(t.v + e) = v;
      __syncthreads();                       // (the previous tile's transposed reads are done)
      tile[tr][4 * tc + 0] = p.x;
      tile[tr][4 * tc + 1] = p.y;
      tile[tr][4 * tc + 2] = p.z;
      tile[tr][4 * tc + 3] = p.w;
      __syncthreads();
      const float4 q = {tile[4 * tc + 0][tr], tile[4 * tc + 1][tr], tile[4 * tc + 2][tr],
                        tile[4 * tc + 3][tr]};
      *reinterpret_cast<float4*>(tab.mirror_pt + (int64_t)(c0 + tr) * R + r0 + 4 * tc) = q;
    }
  }
  for (int k = 0; k < tab.n; ++k) {
    if (k == tab.mirror_k) continue;
    const rlpyt_adam_tensor t = tab.t[k];
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                       reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    const int64_t n4 = vec ? t.n >> 2 : 0;
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(t.g);
    float4* __restrict__ p4 = reinterpret_cast<float4*>(t.p);
    float4* __restrict__ m4 = reinterpret_cast<float4*>(t.m);
    float4* __restrict__ v4 = reinterpret_cast<float4*>(t.v);
    for (int64_t i = gtid; i < n4; i += gsize) {
      const float4 g = g4[i];
      float4 p = p4[i], m = m4[i], v = v4[i];
      upd(g.x, p.x, m.x, v.x);
      upd(g.y, p.y, m.y, v.y);
      upd(g.z, p.z, m.z, v.z);
      upd(g.w, p.w, m.w, v.w);
      p4[i] = p;
      m4[i] = m;
      v4[i] = v;
    }
    for (int64_t i = (n4 << 2) + gtid; i < t.n; i += gsize) {
      float p = t.p[i], m = t.m[i], v = t.v[i];
      upd(t.g[i], p, m, v);
      t.p[i] = p;
      t.m[i] = m;
      t.v[i] = v;
    }
  }
}

// First launch of a captured update: row `cur = *ctr` of the host-computed per-update table becomes
// the update's hyper-parameters, the update's minibatch indices idx_all[cur * M : (cur + 1) * M]
// are copied to the fixed address the captured kernels read, and `cur` is published as a device
// index (for in-graph index_copy_ of the diagnostics row).  ctr itself is advanced by the update's
// last launch (clip_adam_apply_kernel).
__global__ __launch_bounds__(256) void update_tick_kernel(
    const int64_t* __restrict__ ctr, const float* __restrict__ table, int n_rows, int n_cols,
    float* __restrict__ hyper_cur, const int64_t* __restrict__ idx_all,
    int64_t* __restrict__ idx_static, int64_t M, int64_t* __restrict__ tick_idx) {
  const int64_t cur = min(max(ctr[0], (int64_t)0), (int64_t)n_rows - 1);
  if (blockIdx.x == 0 && threadIdx.x < n_cols) hyper_cur[threadIdx.x] = table[cur * n_cols + threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0 && tick_idx != nullptr) tick_idx[0] = cur;
  if (idx_all != nullptr) {
    const int64_t* __restrict__ src = idx_all + cur * M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
         i += (int64_t)gridDim.x * blockDim.x)
      idx_static[i] = src[i];
  }
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

extern "C" int rlpyt_update_tick(const int64_t* ctr, const float* table, int n_rows, int n_cols,
                                 float* hyper_cur, const int64_t* idx_all, int64_t* idx_static,
                                 int64_t M, int64_t* tick_idx, rlpyt_stream_t stream) {
  RL_CHECK_ARG(ctr && table && hyper_cur && n_rows > 0 && n_cols > 0 && n_cols <= 64, RLPYT_EINVAL,
               "rlpyt_update_tick: bad arguments");
  RL_CHECK_ARG((idx_all == nullptr) == (idx_static == nullptr) && M >= 0, RLPYT_EINVAL,
               "rlpyt_update_tick: idx_all and idx_static go together");
  const int grid = idx_all != nullptr ? (int)std::min<int64_t>(std::max<int64_t>(ceil_div(M, 256), 1), 64) : 1;
  RL_LAUNCH(update_tick_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ctr, table, n_rows,
            n_cols, hyper_cur, idx_all, idx_static, M, tick_idx);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int64_t rlpyt_clip_adam_workspace_bytes(void) { return (int64_t)kOptGrid * sizeof(double); }

extern "C" int rlpyt_clip_adam_step_f32(const rlpyt_adam_tensor* tensors_host, int n_tensors,
                                        double lr, double beta1, double beta2, double eps,
                                        double weight_decay, int64_t step, double max_norm,
                                        void* workspace, float* grad_norm_out,
                                        rlpyt_stream_t stream) {
  return rlpyt_clip_adam_step_dev_f32(tensors_host, n_tensors, lr, beta1, beta2, eps, weight_decay,
                                      step, max_norm, workspace, grad_norm_out, nullptr, nullptr,
                                      stream);
}

extern "C" int rlpyt_clip_adam_step_dev_f32(const rlpyt_adam_tensor* tensors_host, int n_tensors,
                                            double lr, double beta1, double beta2, double eps,
                                            double weight_decay, int64_t step, double max_norm,
                                            void* workspace, float* grad_norm_out,
                                            const float* hyper_dev, int64_t* tick_ctr,
                                            rlpyt_stream_t stream) {
  return rlpyt_clip_adam_step_mirror_f32(tensors_host, n_tensors, lr, beta1, beta2, eps, weight_decay,
                                         step, max_norm, workspace, grad_norm_out, hyper_dev, tick_ctr,
                                         -1, nullptr, 0, 0, stream);
}

extern "C" int rlpyt_clip_adam_step_mirror_f32(const rlpyt_adam_tensor* tensors_host, int n_tensors,
                                               double lr, double beta1, double beta2, double eps,
                                               double weight_decay, int64_t step, double max_norm,
                                               void* workspace, float* grad_norm_out,
                                               const float* hyper_dev, int64_t* tick_ctr,
                                               int mirror_index, float* mirror_pt, int64_t mirror_rows,
                                               int64_t mirror_cols, rlpyt_stream_t stream) {
  RL_CHECK_ARG(tensors_host != nullptr && workspace != nullptr, RLPYT_EINVAL,
               "rlpyt_clip_adam_step_f32: null pointer");
  RL_CHECK_ARG(n_tensors > 0 && n_tensors <= RLPYT_ADAM_MAX_TENSORS, RLPYT_ESHAPE,
               "rlpyt_clip_adam_step_f32: %d tensors (1..%d per call)", n_tensors,
               RLPYT_ADAM_MAX_TENSORS);
  RL_CHECK_ARG(step >= 1, RLPYT_EINVAL, "rlpyt_clip_adam_step_f32: step is 1-based (got %ld)",
               (long)step);
  AdamTable tab;
  tab.n = n_tensors;
  tab.mirror_k = -1;
  tab.mirror_rows = tab.mirror_cols = 0;
  tab.mirror_pt = nullptr;
  if (mirror_index >= 0) {
    RL_CHECK_ARG(mirror_index < n_tensors && mirror_pt != nullptr, RLPYT_EINVAL,
                 "rlpyt_clip_adam_step_mirror_f32: mirror_index %d outside the table or null mirror",
                 mirror_index);
    const rlpyt_adam_tensor& t = tensors_host[mirror_index];
    RL_CHECK_ARG(mirror_rows > 0 && mirror_cols > 0 && mirror_rows % 32 == 0 && mirror_cols % 32 == 0 &&
                     mirror_rows * mirror_cols == t.n && mirror_rows < (1 << 30) && mirror_cols < (1 << 30),
                 RLPYT_ESHAPE, "rlpyt_clip_adam_step_mirror_f32: the mirrored tensor must be [rows, cols] "
                 "with both multiples of 32 and rows * cols == n (rows=%ld cols=%ld n=%ld)",
                 (long)mirror_rows, (long)mirror_cols, (long)t.n);
    RL_CHECK_ARG(((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                   reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v) |
                   reinterpret_cast<uintptr_t>(mirror_pt)) & 15) == 0,
                 RLPYT_ESHAPE, "rlpyt_clip_adam_step_mirror_f32: the mirrored tensor, its gradient / "
                 "state and the mirror must be 16-byte aligned");
    tab.mirror_k = mirror_index;
    tab.mirror_rows = (int)mirror_rows;
    tab.mirror_cols = (int)mirror_cols;
    tab.mirror_pt = mirror_pt;
  }
  int64_t total = 0;
  for (int k = 0; k < n_tensors; ++k) {
    const rlpyt_adam_tensor& t = tensors_host[k];
    RL_CHECK_ARG(t.p && t.g && t.m && t.v && t.n > 0, RLPYT_EINVAL,
                 "rlpyt_clip_adam_step_f32: tensor %d has a null pointer or no elements", k);
    RL_CHECK_ARG(((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                   reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 3) == 0,
                 RLPYT_ESHAPE, "rlpyt_clip_adam_step_f32: tensor %d is not 4-byte aligned", k);
    tab.t[k] = t;
    total += t.n;
  }
  hipStream_t s = (hipStream_t)stream;
  const int grid = (int)std::min<int64_t>(ceil_div(total, (int64_t)kOptBlock * 4), kOptGrid);
  double* partial = reinterpret_cast<double*>(workspace);
  // bias corrections in double on the host, as torch does for a host-side step count
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  RL_LAUNCH(clip_adam_norm_kernel, dim3(grid), dim3(kOptBlock), 0, s, tab, partial);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(clip_adam_apply_kernel, dim3(grid), dim3(kOptBlock), 0, s, tab, partial, grid,
            (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)beta1, (float)beta2, (float)eps,
            (float)weight_decay, (float)max_norm, grad_norm_out, hyper_dev, tick_ctr);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
