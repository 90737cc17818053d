// Per-time-step device work of the HBM-resident sampler that is not the model itself
// (roles of rlpyt/samplers/parallel/gpu/collectors.py:30-47 -- writing row t of the sample
// batch -- and of rlpyt/agents/pg/categorical.py:34-43 + distributions/categorical.py:28-31
// -- policy head, softmax, action sampling).  Both kernels read the time index from device
// memory so that ONE captured hipGraph serves every step of the batch.
#include "common.h"

namespace rlpyt {
namespace {

// Copy n_entries byte ranges: dst_e + (t + dt_e) * row_stride_e + col_off_e <- src_e[0:nbytes_e].
// grid = (chunks, n_entries); 16-byte lanes when everything is 16-byte aligned.  An entry with
// zero_where writes zeros instead of the source for every unit u (unit_bytes bytes each) with
// zero_where[u] != 0: the wait-reset collector's blank rows of finished environments
// (rlpyt/samplers/parallel/gpu/collectors.py:85-91) without a masking launch per leaf.
__global__ __launch_bounds__(256) void commit_rows_kernel(const rlpyt_row_copy* __restrict__ table,
                                                          const int64_t* __restrict__ t_dev) {
  const rlpyt_row_copy e = table[blockIdx.y];
  const int64_t t = (t_dev != nullptr ? *t_dev : 0) + e.dt;
  char* __restrict__ dst = static_cast<char*>(e.dst) + t * e.row_stride_bytes + e.col_off_bytes;
  const char* __restrict__ src = static_cast<const char*>(e.src);
  const uint8_t* __restrict__ zw = e.zero_where;
  const int64_t ub = e.unit_bytes;
  const int64_t n = e.nbytes;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
  const bool wide = (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) &&
                    (zw == nullptr || (ub & 15) == 0);
  // (unit lookups in 32-bit arithmetic: a 64-bit division per lane tripled this kernel's time; entries are
  //  far below 4 GB)
  const uint32_t ub32 = (uint32_t)ub;
  if (wide) {
    const int64_t n16 = n >> 4;
    const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
    uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
    if (zw == nullptr) {
      for (int64_t i = tid; i < n16; i += nthr) d4[i] = s4[i];
      for (int64_t i = (n16 << 4) + tid; i < n; i += nthr) dst[i] = src[i];
    } else {
      for (int64_t i = tid; i < n16; i += nthr) {
        uint4 v = s4[i];
        if (zw[((uint32_t)i << 4) / ub32]) v = uint4{0u, 0u, 0u, 0u};
        d4[i] = v;
      }
      for (int64_t i = (n16 << 4) + tid; i < n; i += nthr) dst[i] = zw[(uint32_t)i / ub32] ? 0 : src[i];
    }
  } else if (zw == nullptr) {
    for (int64_t i = tid; i < n; i += nthr) dst[i] = src[i];
  } else {
    for (int64_t i = tid; i < n; i += nthr) dst[i] = zw[(uint32_t)i / ub32] ? 0 : src[i];
  }
}

// One wave per row: logits = h . w_pi^T + b_pi, prob = softmax(logits), value = h . w_v + b_v,
// action = inverse-CDF sample of prob at the given uniform (first a with cumsum > u).
template <int AMAX>
__global__ __launch_bounds__(256) void categorical_head_kernel(
    const float* __restrict__ h, const float* __restrict__ w_pi, const float* __restrict__ b_pi,
    const float* __restrict__ w_v, const float* __restrict__ b_v,
    const float* __restrict__ uniforms, const int64_t* __restrict__ u_row_dev, int64_t n, int K,
    int A, float* __restrict__ prob, float* __restrict__ value, int64_t* __restrict__ action) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* __restrict__ hr = h + row * K;
  float acc[AMAX + 1];
#pragma unroll
  for (int a = 0; a <= AMAX; ++a) acc[a] = 0.f;
  const float* __restrict__ wv_ = w_v != nullptr ? w_v : w_pi;   // always a valid address
  const float vscale = w_v != nullptr ? 1.f : 0.f;
  for (int k = lane; k < K; k += 64) {
    const float x = hr[k];
    // unconditional clamped loads (a >= A re-reads row A-1, its result is ignored below):
    // predicated loads would serialize into one L2 round trip each
#pragma unroll
    for (int a = 0; a < AMAX; ++a) acc[a] = fmaf(x, w_pi[min(a, A - 1) * K + k], acc[a]);
    acc[AMAX] = fmaf(x, wv_[k] * vscale, acc[AMAX]);
  }
#pragma unroll
  for (int a = 0; a <= AMAX; ++a) acc[a] = wave_sum(acc[a]);
  if (lane == 0) {
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      if (a < A) {
        acc[a] += b_pi[a];
        mx = fmaxf(mx, acc[a]);
      }
    float den = 0.f;
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      if (a < A) {
        acc[a] = expf(acc[a] - mx);
        den += acc[a];
      }
    const float inv = 1.f / den;
    float cum = 0.f;
    int pick = -1, last_pos = 0;
    const int64_t urow = u_row_dev != nullptr ? *u_row_dev : 0;
    const float u = uniforms != nullptr ? uniforms[urow * n + row] : 0.f;
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      if (a < A) {
        const float p = acc[a] * inv;
        prob[row * A + a] = p;
        cum += p;
        if (p > 0.f) last_pos = a;
        if (pick < 0 && cum > u) pick = a;
      }
    if (value != nullptr) value[row] = acc[AMAX] + (b_v != nullptr ? b_v[0] : 0.f);
    if (action != nullptr) action[row] = pick >= 0 ? pick : last_pos;  // u ~ 1: rounding guard
  }
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

extern "C" int rlpyt_commit_rows(const rlpyt_row_copy* table_dev, int n_entries,
                                 int64_t max_entry_bytes, const int64_t* t_dev,
                                 rlpyt_stream_t stream) {
  RL_CHECK_ARG(n_entries >= 0 && n_entries <= 64, RLPYT_EINVAL, "rlpyt_commit_rows: 0..64 entries");
  if (n_entries == 0) return RLPYT_OK;
  RL_CHECK_ARG(table_dev != nullptr, RLPYT_EINVAL, "rlpyt_commit_rows: null table");
  RL_CHECK_ARG(max_entry_bytes >= 0 && max_entry_bytes < (1LL << 31), RLPYT_ESHAPE,
               "rlpyt_commit_rows: entries must be smaller than 2 GB");
  const int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(ceil_div(max_entry_bytes, 256 * 16), 1024));
  RL_LAUNCH(commit_rows_kernel, dim3((unsigned)chunks, (unsigned)n_entries), dim3(256), 0,
                     (hipStream_t)stream, table_dev, t_dev);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_categorical_head_f32(const float* h, const float* w_pi, const float* b_pi,
                                          const float* w_v, const float* b_v,
                                          const float* uniforms, const int64_t* u_row_dev,
                                          int64_t n, int K, int A, float* prob, float* value,
                                          int64_t* action, rlpyt_stream_t stream) {
  RL_CHECK_ARG(n >= 0 && K > 0 && A > 0 && A <= 32, RLPYT_EINVAL,
               "rlpyt_categorical_head_f32: need 0 < A <= 32, K > 0");
  if (n == 0) return RLPYT_OK;
  RL_CHECK_ARG(h && w_pi && b_pi && prob, RLPYT_EINVAL, "rlpyt_categorical_head_f32: null pointer");
  RL_CHECK_ARG((action == nullptr) || (uniforms != nullptr), RLPYT_EINVAL,
               "rlpyt_categorical_head_f32: sampling needs uniforms");
  RL_CHECK_ARG((value == nullptr) == (w_v == nullptr), RLPYT_EINVAL,
               "rlpyt_categorical_head_f32: value and w_v go together");
  const dim3 grid((unsigned)ceil_div(n, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (A <= 8)
    RL_LAUNCH((categorical_head_kernel<8>), grid, block, 0, s, h, w_pi, b_pi, w_v, b_v,
                       uniforms, u_row_dev, n, K, A, prob, value, action);
  else
    RL_LAUNCH((categorical_head_kernel<32>), grid, block, 0, s, h, w_pi, b_pi, w_v, b_v,
                       uniforms, u_row_dev, n, K, A, prob, value, action);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

// --------------------------------------------------------------------------------------
// Frame-stack push: observations of frame-stacked envs (rlpyt/envs/atari/atari_env.py:
// 115-118 keeps obs = the last C frames, newest last) differ from the previous step's only
// by one frame, so the host uploads just that frame (C x less PCIe traffic) and the stack is
// rebuilt in HBM:  obs[t, lo+b] = slot[b] >= 0 ? full_rows[slot[b]]            (env was reset /
//                                                                               first step of a batch)
//                               : concat(obs[t-1, lo+b, 1:], new_frame[b]).
// Also mirrors the row into `stage` (nullable) for agents that want a fixed-address input.
// One workgroup-row per env (grid.y), 16-byte lanes; t is read from device memory.
namespace rlpyt {
namespace {
__global__ __launch_bounds__(256) void frame_push_kernel(
    uint8_t* __restrict__ obs, const int64_t* __restrict__ t_dev, int64_t B, int64_t lo, int C,
    int64_t HW, const uint8_t* __restrict__ new_frame, const uint8_t* __restrict__ full_rows,
    const int32_t* __restrict__ slot, uint8_t* __restrict__ stage, float* __restrict__ reward_rows,
    const float* __restrict__ reward_src, uint8_t* __restrict__ done_rows,
    const uint8_t* __restrict__ done_src) {
  const int64_t b = blockIdx.y;
  const int64_t t = *t_dev;
  if (reward_rows != nullptr && blockIdx.x == 0 && b == 0) {
    // the step's scalar rows ride along: all_reward[t, lo:lo+Bg], all_done[t, lo:lo+Bg]
    for (int i = threadIdx.x; i < (int)gridDim.y; i += blockDim.x) {
      reward_rows[t * B + lo + i] = reward_src[i];
      done_rows[t * B + lo + i] = done_src[i];
    }
  }
  const int64_t row_bytes = (int64_t)C * HW;
  uint8_t* __restrict__ dst = obs + (t * B + lo + b) * row_bytes;
  uint8_t* __restrict__ dst2 = stage != nullptr ? stage + b * row_bytes : nullptr;
  const int sl = slot[b];
  const int64_t n16 = row_bytes >> 4, hw16 = HW >> 4;
  const uint4* __restrict__ full = reinterpret_cast<const uint4*>(full_rows + (int64_t)(sl < 0 ? 0 : sl) * row_bytes);
  const uint4* __restrict__ prev = reinterpret_cast<const uint4*>(obs + ((t - 1) * B + lo + b) * row_bytes);
  const uint4* __restrict__ nf = reinterpret_cast<const uint4*>(new_frame + b * HW);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint4 v;
    if (sl >= 0) v = full[i];
    else if (i < n16 - hw16) v = prev[i + hw16];
    else v = nf[i - (n16 - hw16)];
    reinterpret_cast<uint4*>(dst)[i] = v;
    if (dst2 != nullptr) reinterpret_cast<uint4*>(dst2)[i] = v;
  }
}
}  // namespace
}  // namespace rlpyt

extern "C" int rlpyt_frame_push(uint8_t* obs, const int64_t* t_dev, int64_t B, int64_t lo,
                                int64_t Bg, int C, int64_t HW, const uint8_t* new_frame,
                                const uint8_t* full_rows, const int32_t* slot, uint8_t* stage,
                                float* reward_rows, const float* reward_src, uint8_t* done_rows,
                                const uint8_t* done_src, rlpyt_stream_t stream) {
  RL_CHECK_ARG(B > 0 && lo >= 0 && Bg >= 0 && lo + Bg <= B && C > 0 && HW > 0, RLPYT_EINVAL,
               "rlpyt_frame_push: bad sizes");
  if (Bg == 0) return RLPYT_OK;
  RL_CHECK_ARG(obs && t_dev && new_frame && full_rows && slot, RLPYT_EINVAL,
               "rlpyt_frame_push: null pointer");
  RL_CHECK_ARG((reward_rows == nullptr) || (reward_src && done_rows && done_src), RLPYT_EINVAL,
               "rlpyt_frame_push: reward/done rows go together");
  RL_CHECK_ARG(HW % 16 == 0 && ((reinterpret_cast<uintptr_t>(obs) | reinterpret_cast<uintptr_t>(new_frame) |
                                 reinterpret_cast<uintptr_t>(full_rows) | reinterpret_cast<uintptr_t>(stage)) & 15) == 0,
               RLPYT_ESHAPE, "rlpyt_frame_push: H*W must be a multiple of 16 and buffers 16-byte aligned");
  const unsigned gx = (unsigned)std::min<int64_t>(ceil_div((int64_t)C * HW / 16, 256), 4);
  RL_LAUNCH(frame_push_kernel, dim3(gx, (unsigned)Bg), dim3(256), 0, (hipStream_t)stream, obs,
                     t_dev, B, lo, C, HW, new_frame, full_rows, slot, stage, reward_rows, reward_src,
                     done_rows, done_src);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

// --------------------------------------------------------------------------------------
// Small-batch fully connected layer + ReLU for the sampling forward (the 3456 -> 512 trunk of
// rlpyt/models/pg/atari_ff_model.py:52-55 via rlpyt/models/mlp.py at M = 128..256 rows):
//   y[m, n] = relu(sum_k x[m, k] w[n, k] + b[n]).
// At this size a library GEMM is latency-bound (~20 us); here the K range is split over
// KSPLIT workgroups per 16-wide column tile so that all CUs stream a slice of the 7 MB weight
// matrix at once, on fp32 MFMA (A = w rows, B = x rows, both read as float4 along K with the
// K order inside 16 permuted identically), and a second tiny kernel sums the KSPLIT partials
// in a fixed order, adds the bias and applies the ReLU (deterministic, no atomics).
namespace rlpyt {
namespace {
typedef float fc_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kFcKSplit = 8;

template <int MT>  // m-tiles of 16 rows per wave (4 waves): M <= 64 * MT
__global__ __launch_bounds__(256) void fc_small_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       float* __restrict__ partial, int M, int N,
                                                       int K, int kchunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int nt = blockIdx.x, ks = blockIdx.y;
  const int k0 = ks * kchunk;
  const int k1 = min(k0 + kchunk, K);
  const float* __restrict__ wrow = w + (int64_t)(nt * 16 + j) * K + 4 * kq;
  const float* xrow[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = min((wave + 4 * t) * 16 + j, M - 1);
    xrow[t] = x + (int64_t)m * K + 4 * kq;
  }
  fc_f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = fc_f32x4{0.f, 0.f, 0.f, 0.f};
  // K-groups of 16 fetched together: the loads of a slab are independent, the slabs are not -- a
  // workgroup's K chunk (432 at the trunk's 3456 / 8) is a chain of chunk / (16 U) L2 round trips,
  // and at 64 rows the kernel is nothing but that chain (U = 3: 9 trips, 9.9 us per launch)
  constexpr int U = MT == 1 ? 9 : (MT == 2 ? 6 : 3);
  int k = k0;
  for (; k + 16 * U <= k1; k += 16 * U) {
    fc_f32x4 a[U], b[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = *reinterpret_cast<const fc_f32x4*>(wrow + k + 16 * u);
#pragma unroll
      for (int t = 0; t < MT; ++t) b[u][t] = *reinterpret_cast<const fc_f32x4*>(xrow[t] + k + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp)
#pragma unroll
        for (int t = 0; t < MT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][sp], b[u][t][sp], acc[t], 0, 0, 0);
  }
  for (; k < k1; k += 16) {
    const fc_f32x4 a = *reinterpret_cast<const fc_f32x4*>(wrow + k);
    fc_f32x4 b[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) b[t] = *reinterpret_cast<const fc_f32x4*>(xrow[t] + k);
#pragma unroll
    for (int sp = 0; sp < 4; ++sp)
#pragma unroll
      for (int t = 0; t < MT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sp], b[t][sp], acc[t], 0, 0, 0);
  }
  // D[row = n_local = 4*kq + r][col = m_local = j]
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = (wave + 4 * t) * 16 + j;
    if (m < M)
      *reinterpret_cast<fc_f32x4*>(partial + ((int64_t)ks * M + m) * N + nt * 16 + 4 * kq) = acc[t];
  }
}

__global__ __launch_bounds__(256) void fc_small_finish_kernel(const float* __restrict__ partial,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ y, int64_t MN,
                                                              int N, int ksplit, int relu) {
  const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e >= MN) return;
  fc_f32x4 s = *reinterpret_cast<const fc_f32x4*>(partial + e);
  for (int k = 1; k < ksplit; ++k) {
    const fc_f32x4 v = *reinterpret_cast<const fc_f32x4*>(partial + (int64_t)k * MN + e);
    s += v;
  }
  const int n = (int)(e % N);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = s[r] + (bias != nullptr ? bias[n + r] : 0.f);
    s[r] = relu ? fmaxf(v, 0.f) : v;
  }
  *reinterpret_cast<fc_f32x4*>(y + e) = s;
}

// One LSTM cell step behind the split-K GEMM above (the per-time-step sampling forward of the
// recurrent agents, rlpyt/models/dqn/atari_r2d1_model.py:61-63 / torch.nn.LSTM with T = 1):
//   gates[b, :] = sum_s partial[s][b][:] + b_ih + b_hh      (partials of [x | h] [W_ih | W_hh]^T)
//   i, f, g, o = gates[0:H], [H:2H], [2H:3H], [3H:4H]        (torch gate order)
//   c' = sigmoid(f) c + sigmoid(i) tanh(g);   h' = sigmoid(o) tanh(c')
// One thread per (row, hidden unit), all its partials (<= 8 slices x 4 gates, clamped addresses,
// masked sums: no branch between the loads) in flight at once; slices summed in slice order ->
// deterministic.  (Until round 5: one thread per FOUR units with the slice loop around dependent
// 16-byte loads -- 24 workgroups and eight latency trips for B = 48, H = 512: 10.8 us in the R2D1
// rollout's step graph, profiles/r5_r2d1_region_fused_convs.txt.)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ partial, int ksplit,
                                                        const float* __restrict__ b_ih,
                                                        const float* __restrict__ b_hh,
                                                        const float* __restrict__ c_prev,
                                                        float* __restrict__ h_out,
                                                        float* __restrict__ c_out, int64_t B, int H) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // element of [B, H]
  if (e >= B * H) return;
  const int64_t b = e / H;
  const int j = (int)(e - b * H);
  const int64_t BN = B * 4 * (int64_t)H;
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < ksplit; s0 += 8) {
    float pv[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        pv[q][u] = partial[(int64_t)min(s0 + u, ksplit - 1) * BN + b * 4 * (int64_t)H + (int64_t)q * H + j];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 8; ++u) g[q] += (s0 + u < ksplit ? pv[q][u] : 0.f);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) g[q] = (g[q] + b_ih[q * H + j]) + b_hh[q * H + j];   // (x W_ih^T + h W_hh^T + b_ih) + b_hh
  const float c1 = sigmoid_f(g[1]) * c_prev[e] + sigmoid_f(g[0]) * tanhf(g[2]);
  c_out[e] = c1;
  h_out[e] = sigmoid_f(g[3]) * tanhf(c1);
}
}  // namespace
}  // namespace rlpyt

// Inputs of one recurrent sampling step, assembled in ONE launch (rlpyt/agents/dqn/r2d1_agent.py:23-40
// with the collector's reset handling, rlpyt/samplers/parallel/gpu/action_server.py:49-53 and
// rlpyt/agents/base.py:283-297): row b of
//   xh = [ act(feat[b]) | onehot(prev_action[b]) | prev_reward[b] | h[b] | 0-pad ]   ([B, Kp], the
//        [x | h] row rlpyt_fc_small_f32 multiplies with [W_ih | W_hh])
// where for an environment that was reset before this step (done[b]) the previous action is the
// null action 0, the previous reward 0 and the recurrent state zero; prev_h / prev_c receive the state
// the step starts from (what the reference stores as agent_info.prev_rnn_state) and c is zeroed IN
// PLACE for reset environments (h needs no write-back: the cell kernel overwrites it).  Replaces
// where x 2 + fills + one-hot scatter + mask casts + two state multiplies + concatenation + two
// state clones of the eager path (14 launches).  One workgroup per row.
namespace rlpyt {
namespace {
__global__ __launch_bounds__(256) void rnn_step_inputs_kernel(
    const float* __restrict__ feat, int F, int relu, const int64_t* __restrict__ action, int A,
    const float* __restrict__ reward, const uint8_t* __restrict__ done, const float* __restrict__ h,
    float* __restrict__ c, int H, float* __restrict__ xh, int Kp, float* __restrict__ prev_h,
    float* __restrict__ prev_c) {
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const bool keep = done == nullptr || done[b] == 0;
  float* __restrict__ row = xh + b * Kp;
  for (int k = tid; k < F; k += 256) {
    const float v = feat[b * F + k];
    row[k] = relu ? fmaxf(v, 0.f) : v;
  }
  const int64_t act = keep ? action[b] : 0;
  for (int a = tid; a < A; a += 256) row[F + a] = (a == act) ? 1.f : 0.f;
  if (tid == 0) row[F + A] = keep ? reward[b] : 0.f;
  const int h0 = F + A + 1;
  for (int k = tid; k < H; k += 256) {
    const float hv = keep ? h[b * H + k] : 0.f, cv = keep ? c[b * H + k] : 0.f;
    row[h0 + k] = hv;
    prev_h[b * H + k] = hv;
    prev_c[b * H + k] = cv;
    if (!keep) c[b * H + k] = 0.f;
  }
  for (int k = h0 + H + tid; k < Kp; k += 256) row[k] = 0.f;
}
}  // namespace
}  // namespace rlpyt

extern "C" int rlpyt_rnn_step_inputs_f32(const float* feat, int F, int relu, const int64_t* action,
                                         int A, const float* reward, const uint8_t* done,
                                         const float* h, float* c, int H, float* xh, int Kp,
                                         float* prev_h, float* prev_c, int64_t B,
                                         rlpyt_stream_t stream) {
  RL_CHECK_ARG(feat && action && reward && h && c && xh && prev_h && prev_c, RLPYT_EINVAL,
               "rlpyt_rnn_step_inputs_f32: null pointer");
  RL_CHECK_ARG(B > 0 && F > 0 && A > 0 && H > 0 && Kp >= F + A + 1 + H, RLPYT_ESHAPE,
               "rlpyt_rnn_step_inputs_f32: need B, F, A, H > 0 and Kp >= F + A + 1 + H (F=%d A=%d H=%d Kp=%d)",
               F, A, H, Kp);
  RL_LAUNCH(rlpyt::rnn_step_inputs_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, feat,
            F, relu, action, A, reward, done, h, c, H, xh, Kp, prev_h, prev_c);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

// Epsilon-greedy action selection of the DQN-family agents' sampling step
// (rlpyt/distributions/epsilon_greedy.py:17-29: argmax, then with probability epsilon a uniformly
// random action) from ONE pre-drawn uniform per environment and step: u < eps -> the action
// floor(u / eps * A) (u / eps is uniform on [0, 1) given u < eps), else the first maximal Q.
// One thread per environment; replaces argmax + rand + randint + compare + where (5 launches).
namespace rlpyt {
namespace {
__global__ __launch_bounds__(256) void eps_greedy_kernel(const float* __restrict__ q, int A,
                                                         const float* __restrict__ eps, int eps_stride,
                                                         const float* __restrict__ uniforms,
                                                         const int64_t* __restrict__ t_dev, int64_t n,
                                                         int64_t* __restrict__ action) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const float* row = q + b * A;
  int best = 0;
  float vb = row[0];
  for (int a = 1; a < A; ++a) {
    const float v = row[a];
    if (v > vb) { vb = v; best = a; }       // first maximum, as torch.argmax
  }
  const int64_t t = t_dev ? t_dev[0] : 0;
  const float u = uniforms[t * n + b], e = eps[b * eps_stride];
  int a_rand = (int)(u / e * (float)A);
  a_rand = a_rand < A ? a_rand : A - 1;
  action[b] = u < e ? a_rand : best;
}
}  // namespace
}  // namespace rlpyt

extern "C" int rlpyt_eps_greedy_f32(const float* q, int64_t n, int A, const float* eps, int eps_stride,
                                    const float* uniforms, const int64_t* t_dev, int64_t* action,
                                    rlpyt_stream_t stream) {
  RL_CHECK_ARG(q && eps && uniforms && action, RLPYT_EINVAL, "rlpyt_eps_greedy_f32: null pointer");
  RL_CHECK_ARG(n > 0 && A > 0 && (eps_stride == 0 || eps_stride == 1), RLPYT_ESHAPE,
               "rlpyt_eps_greedy_f32: need n > 0, A > 0, eps_stride 0 | 1 (n=%ld A=%d stride=%d)",
               (long)n, A, eps_stride);
  RL_LAUNCH(rlpyt::eps_greedy_kernel, dim3((unsigned)rlpyt::ceil_div(n, 256)), dim3(256), 0,
            (hipStream_t)stream, q, A, eps, eps_stride, uniforms, t_dev, n, action);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_lstm_cell_f32(const float* partial, int ksplit, const float* b_ih,
                                   const float* b_hh, const float* c_prev, float* h_out,
                                   float* c_out, int64_t B, int H, rlpyt_stream_t stream) {
  RL_CHECK_ARG(partial && b_ih && b_hh && c_prev && h_out && c_out, RLPYT_EINVAL,
               "rlpyt_lstm_cell_f32: null pointer");
  RL_CHECK_ARG(B > 0 && H > 0 && H % 4 == 0 && ksplit > 0, RLPYT_ESHAPE,
               "rlpyt_lstm_cell_f32: need B > 0, H a positive multiple of 4, ksplit > 0 (B=%ld H=%d "
               "ksplit=%d)", (long)B, H, ksplit);
  RL_CHECK_ARG(((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(b_ih) |
                 reinterpret_cast<uintptr_t>(b_hh) | reinterpret_cast<uintptr_t>(c_prev) |
                 reinterpret_cast<uintptr_t>(h_out) | reinterpret_cast<uintptr_t>(c_out)) & 15) == 0,
               RLPYT_ESHAPE, "rlpyt_lstm_cell_f32: buffers must be 16-byte aligned");
  RL_LAUNCH(rlpyt::lstm_cell_kernel, dim3((unsigned)rlpyt::ceil_div(B * H, 256)), dim3(256), 0,
            (hipStream_t)stream, partial, ksplit, b_ih, b_hh, c_prev, h_out, c_out, B, H);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int64_t rlpyt_fc_small_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (int64_t)rlpyt::kFcKSplit * M * N * (int64_t)sizeof(float);
}

extern "C" int rlpyt_fc_small_ksplit(int K) {
  if (K <= 0) return 0;
  const int kchunk = (int)(rlpyt::ceil_div(rlpyt::ceil_div(K, rlpyt::kFcKSplit), 16) * 16);
  return (int)rlpyt::ceil_div(K, kchunk);
}

extern "C" int rlpyt_fc_small_f32(const float* x, const float* w, const float* bias, float* y,
                                  int M, int N, int K, int relu, float* workspace,
                                  rlpyt_stream_t stream) {
  RL_CHECK_ARG(x && w && workspace, RLPYT_EINVAL, "rlpyt_fc_small_f32: null pointer");
  RL_CHECK_ARG(M > 0 && M <= 256 && N > 0 && N % 16 == 0 && K > 0 && K % 16 == 0, RLPYT_ESHAPE,
               "rlpyt_fc_small_f32: need 0 < M <= 256, N %% 16 == 0, K %% 16 == 0 (M=%d N=%d K=%d)",
               M, N, K);
  RL_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) |
                 reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
               RLPYT_ESHAPE, "rlpyt_fc_small_f32: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int kchunk = (int)(ceil_div(ceil_div(K, rlpyt::kFcKSplit), 16) * 16);
  const int ksplit = (int)ceil_div(K, kchunk);
  const dim3 grid((unsigned)(N / 16), (unsigned)ksplit);
  if (M <= 64)
    RL_LAUNCH((rlpyt::fc_small_kernel<1>), grid, dim3(256), 0, s, x, w, workspace, M, N, K, kchunk);
  else if (M <= 128)
    RL_LAUNCH((rlpyt::fc_small_kernel<2>), grid, dim3(256), 0, s, x, w, workspace, M, N, K, kchunk);
  else
    RL_LAUNCH((rlpyt::fc_small_kernel<4>), grid, dim3(256), 0, s, x, w, workspace, M, N, K, kchunk);
  RL_LAUNCH_CHECK();
  if (y == nullptr) return RLPYT_OK;   // caller consumes the split-K partials itself
  const int64_t MN = (int64_t)M * N;
  RL_LAUNCH(rlpyt::fc_small_finish_kernel, dim3((unsigned)ceil_div(MN / 4, 256)), dim3(256), 0,
                     s, workspace, bias, y, MN, N, ksplit, relu);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

// --------------------------------------------------------------------------------------
// Rollout chain, trunk + head (round 4; fc_small_kernel stays for M > 64 callers and the LSTM gate
// GEMM, its one-wave-per-row head kernel of round 3 is gone).
//
// Why: round 3's split (16 columns x K/8 per workgroup) made every CU ingest the x slab of its K
// slice for ALL rows plus its W rows once per WAVE -- 56 MB of L2 -> L1 traffic per launch for
// 7 MB of weights, 8.4 us, bound by the ~10..30 B/clk a CU can pull through its L1, not by MFMA
// (1.4 us) or HBM.  The head then read 8 partials x 512 per row on 16 CUs (139 KB each, 6.2 us).
// Here a workgroup owns 64 columns x 128 K for a block of 64 rows: W rows are loaded once per
// workgroup (a wave owns 16 columns, all four 16-row tiles), the x slab is 32 KB (L1-resident, the
// four waves share it), 64 KB ingest per CU instead of 218 KB, all 40 loads of a wave in flight at
// once (one latency trip), and the head runs one WORKGROUP per row (4 waves x K/4 each, 64 CUs).
// Partial layout [ksplit][M][N] as before; fixed summation order -> deterministic.
namespace rlpyt {
namespace {
constexpr int kRfcKc = 128;       // K slice per workgroup
constexpr int kRfcMaxSplit = 32;  // K <= 4096

__global__ __launch_bounds__(256) void rollout_fc_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         float* __restrict__ partial, int M, int N,
                                                         int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  // XCD-aware (column block, K slice) map (round 6): workgroups go round-robin over the 8 XCDs in
  // launch order, so with (blockIdx.x, blockIdx.y) = (column block, slice) XCD c ran column block c of
  // EVERY slice and each XCD's L2 fetched the whole x for itself (8 x 0.88 MB: PMC traffic 1.54 x the
  // algorithmic bytes, profiles/r5_rollout_pmc.json with the calibrated x2 of r6_fetch_calibration.json).
  // Here the units are numbered slice-major and XCD c takes units [c n/8, (c+1) n/8): the column blocks
  // of one slice share their x slab through one L2.
  int cb = blockIdx.x, ks = blockIdx.y;
  {
    const int total = gridDim.x * gridDim.y;
    if ((total & 7) == 0) {
      const int L = blockIdx.x + gridDim.x * blockIdx.y;
      const int u = (L & 7) * (total >> 3) + (L >> 3);
      ks = u / (int)gridDim.x;
      cb = u - ks * (int)gridDim.x;
    }
  }
  const int n0 = cb * 64 + wave * 16;
  const int k0 = ks * kRfcKc;
  const int kn = min(kRfcKc, K - k0);                 // multiple of 16, >= 16
  const int m0 = blockIdx.z * 64;
  const float* __restrict__ wrow = w + (int64_t)(n0 + j) * K + k0 + 4 * kq;
  const float* xrow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = min(m0 + 16 * t + j, M - 1);        // clamped: rows >= M are computed, not stored
    xrow[t] = x + (int64_t)m * K + k0 + 4 * kq;
  }
  fc_f32x4 a[8], b[8][4];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int kk = min(16 * g, kn - 16);              // short last slice: re-read, MFMAs skipped
    a[g] = *reinterpret_cast<const fc_f32x4*>(wrow + kk);
#pragma unroll
    for (int t = 0; t < 4; ++t) b[g][t] = *reinterpret_cast<const fc_f32x4*>(xrow[t] + kk);
  }
  fc_f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = fc_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (16 * g < kn) {
#pragma unroll
      for (int sp = 0; sp < 4; ++sp)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][sp], b[g][t][sp], acc[t], 0, 0, 0);
    }
  }
  // D[row = n_local = 4*kq + r][col = m_local = j]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = m0 + 16 * t + j;
    if (m < M)
      *reinterpret_cast<fc_f32x4*>(partial + ((int64_t)ks * M + m) * N + n0 + 4 * kq) = acc[t];
  }
}

// One workgroup per row: wave w finishes the trunk for k in [w K/4, (w+1) K/4) (sum of the ksplit
// partials in slice order, + bias, ReLU), takes its share of the 7 head dot products, the four
// shares meet in LDS and thread 0 does softmax + inverse-CDF draw + the step's row writes (as
// rlpyt_categorical_head_f32).  bootstrap_out != NULL: value only, written to bootstrap_out[row] (the
// bootstrap-value pass after the last step of a batch).
template <int KW>   // trunk width = 256 * KW
__global__ __launch_bounds__(256) void rollout_head_kernel(
    const float* __restrict__ partial, int ksplit, const float* __restrict__ fc_bias,
    const float* __restrict__ w_pi, const float* __restrict__ b_pi, const float* __restrict__ w_v,
    const float* __restrict__ b_v, const float* __restrict__ uniforms,
    const int64_t* __restrict__ t_dev, int64_t n, int A, float* __restrict__ prob_rows,
    float* __restrict__ value_rows, int64_t* __restrict__ action_rows, int64_t B, int64_t lo,
    int64_t* __restrict__ action_out, float* __restrict__ bootstrap_out) {
  constexpr int K = 256 * KW;
  constexpr int AMAX = 8;
  __shared__ float red[4][AMAX + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = blockIdx.x;
  // unconditional clamped loads, masked sums: predicated loads compile to a branch and an
  // s_waitcnt vmcnt(0) EACH on gfx950 (~60 serialized L2 round trips in the first version)
  float hv[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) hv[i] = 0.f;
  const int kb = wave * 64 * KW + lane;
#pragma unroll
  for (int s0 = 0; s0 < kRfcMaxSplit; s0 += 8) {
    if (s0 < ksplit) {                                 // uniform
      float pv[KW][8];
#pragma unroll
      for (int i = 0; i < KW; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          pv[i][u] = partial[((int64_t)min(s0 + u, ksplit - 1) * n + row) * K + kb + 64 * i];
#pragma unroll
      for (int i = 0; i < KW; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) hv[i] += (s0 + u < ksplit ? pv[i][u] : 0.f);
    }
  }
  float wp[AMAX][KW], wvv[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int k = kb + 64 * i;
    hv[i] = fmaxf(hv[i] + fc_bias[k], 0.f);
#pragma unroll
    for (int a = 0; a < AMAX; ++a) wp[a][i] = w_pi[min(a, A - 1) * K + k];
    wvv[i] = w_v[k];
  }
  float acc[AMAX + 1];
#pragma unroll
  for (int a = 0; a <= AMAX; ++a) acc[a] = 0.f;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
#pragma unroll
    for (int a = 0; a < AMAX; ++a) acc[a] = fmaf(hv[i], wp[a][i], acc[a]);
    acc[AMAX] = fmaf(hv[i], wvv[i], acc[AMAX]);
  }
#pragma unroll
  for (int a = 0; a <= AMAX; ++a) acc[a] = wave_sum(acc[a]);
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a <= AMAX; ++a) red[wave][a] = acc[a];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int a = 0; a <= AMAX; ++a) acc[a] = ((red[0][a] + red[1][a]) + red[2][a]) + red[3][a];
  const float val = acc[AMAX] + b_v[0];
  if (bootstrap_out != nullptr) {
    bootstrap_out[row] = val;
    return;
  }
  const int64_t t = *t_dev;
  const float u = uniforms[t * n + row];
  float mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      acc[a] += b_pi[a];
      mx = fmaxf(mx, acc[a]);
    }
  float den = 0.f;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      acc[a] = expf(acc[a] - mx);
      den += acc[a];
    }
  const float inv = 1.f / den;
  float cum = 0.f;
  int pick = -1, last_pos = 0;
  float* __restrict__ pr = prob_rows + (t * B + lo + row) * A;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      const float p = acc[a] * inv;
      pr[a] = p;
      cum += p;
      if (p > 0.f) last_pos = a;
      if (pick < 0 && cum > u) pick = a;
    }
  value_rows[t * B + lo + row] = val;
  const int64_t act = pick >= 0 ? pick : last_pos;
  action_rows[(t + 1) * B + lo + row] = act;
  action_out[row] = act;
}

}  // namespace
}  // namespace rlpyt

// Q-value head of the DQN-family sampling / target passes behind a split-K hidden layer
// (rlpyt/models/mlp.py:24-31 as the `head` of rlpyt/models/dqn/atari_dqn_model.py:50-51 and
// atari_r2d1_model.py:44-45: Linear -> ReLU -> Linear): one workgroup per row finishes the hidden layer
// (sum of the ksplit partials of rlpyt_fc_small_f32 in slice order, + bias, ReLU) and takes the A output
// dot products -- the structure of rollout_head_kernel without the sampling.  Replaces library GEMM +
// clamp + library GEMM (20-25 us for 8..48 rows) by the split-K kernel + this one.
namespace rlpyt {
namespace {
template <int KW>   // hidden width = 256 * KW
__global__ __launch_bounds__(256) void q_head_kernel(const float* __restrict__ partial, int ksplit,
                                                     const float* __restrict__ b_hidden,
                                                     const float* __restrict__ w_out,
                                                     const float* __restrict__ b_out, int64_t n, int A,
                                                     float* __restrict__ q, float* __restrict__ h_out) {
  constexpr int K = 256 * KW;
  constexpr int AMAX = 18;                             // the full Atari action set
  __shared__ float red[4][AMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = blockIdx.x;
  float hv[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) hv[i] = 0.f;
  const int kb = wave * 64 * KW + lane;
  for (int s0 = 0; s0 < ksplit; s0 += 8) {
    float pv[KW][8];
#pragma unroll
    for (int i = 0; i < KW; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        pv[i][u] = partial[((int64_t)min(s0 + u, ksplit - 1) * n + row) * K + kb + 64 * i];
#pragma unroll
    for (int i = 0; i < KW; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u) hv[i] += (s0 + u < ksplit ? pv[i][u] : 0.f);
  }
  float acc[AMAX];
#pragma unroll
  for (int a = 0; a < AMAX; ++a) acc[a] = 0.f;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int k = kb + 64 * i;
    hv[i] = fmaxf(hv[i] + b_hidden[k], 0.f);
    if (h_out != nullptr) h_out[row * K + k] = hv[i];   // (kept for the backward pass: rlpyt_q_head_train_f32)
#pragma unroll
    for (int a = 0; a < AMAX; ++a) acc[a] = fmaf(hv[i], w_out[(int64_t)min(a, A - 1) * K + k], acc[a]);
  }
#pragma unroll
  for (int a = 0; a < AMAX; ++a) acc[a] = wave_sum(acc[a]);
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < AMAX; ++a) red[wave][a] = acc[a];
  }
  __syncthreads();
  const int a = threadIdx.x;
  if (a < A) q[row * A + a] = (((red[0][a] + red[1][a]) + red[2][a]) + red[3][a]) + b_out[a];
}

// Backward of that head up to the hidden layer's pre-activation (round 6): from dq [n, A] and the kept
// hidden activations h [n, K]
//   dw_out [A, K] = dq^T h,  db_out [A] = column sums of dq,
//   dh [n, K] = (dq w_out) * (h > 0)   (= the gradient at the hidden layer's pre-activation),
//   db_hidden [K] = column sums of dh
// -- what autograd runs as fill + fill + multiply + two GEMMs of < 6 us + a column-sum reduction + the ReLU's
// threshold kernel + another reduction (8 launches, ~45 us inside a captured DQN update).  A workgroup owns
// 64 hidden units (one per lane), its sixteen waves take every sixteenth row (their h values requested up
// front) and meet in LDS in wave order: deterministic (8 us; a four-wave version with the loads inside
// the row loop took 21).  The hidden layer's own weight / input gradients (dh^T x, dh W) stay GEMMs of the caller.
constexpr int kQbAmax = 18, kQbRows = 256, kQbWaves = 16, kQbRpw = kQbRows / kQbWaves;
__global__ __launch_bounds__(kQbWaves * 64) void q_head_bwd_kernel(const float* __restrict__ dq,
                                                                  const float* __restrict__ h,
                                                                  const float* __restrict__ w_out, int n,
                                                                  int K, int A, float* __restrict__ dw_out,
                                                                  float* __restrict__ db_out,
                                                                  float* __restrict__ dh,
                                                                  float* __restrict__ db_hidden) {
  __shared__ float sdq[kQbRows * kQbAmax];
  __shared__ float red[kQbWaves][kQbAmax + 1][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x * 64 + lane;
  // this wave's rows wave, wave + 16, ...: their hidden activations requested up front (one round trip)
  float hv[kQbRpw];
#pragma unroll
  for (int i = 0; i < kQbRpw; ++i) {
    const int m = wave + kQbWaves * i;
    hv[i] = m < n ? h[(int64_t)m * K + k] : 0.f;
  }
  for (int i = tid; i < n * A; i += kQbWaves * 64) sdq[i] = dq[i];
  float w2[kQbAmax], dw[kQbAmax];
#pragma unroll
  for (int a = 0; a < kQbAmax; ++a) {
    w2[a] = a < A ? w_out[(int64_t)a * K + k] : 0.f;
    dw[a] = 0.f;
  }
  float db = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kQbRpw; ++i) {
    const int m = wave + kQbWaves * i;
    if (m < n) {                                               // (uniform per wave)
      const float* __restrict__ d = sdq + m * A;
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < kQbAmax; ++a)
        if (a < A) {
          const float g = d[a];
          s = fmaf(g, w2[a], s);
          dw[a] = fmaf(g, hv[i], dw[a]);
        }
      const float g1 = hv[i] > 0.f ? s : 0.f;
      dh[(int64_t)m * K + k] = g1;
      db += g1;
    }
  }
#pragma unroll
  for (int a = 0; a < kQbAmax; ++a) red[wave][a][lane] = dw[a];
  red[wave][kQbAmax][lane] = db;
  __syncthreads();
  // the 16 wave partials of every (output | hidden bias, column) summed in wave order
  for (int it = tid; it < (A + 1) * 64; it += kQbWaves * 64) {
    const int a = it >> 6, c = it & 63, slot = a < A ? a : kQbAmax;
    float s = red[0][slot][c];
#pragma unroll
    for (int w = 1; w < kQbWaves; ++w) s += red[w][slot][c];
    if (a < A) dw_out[(int64_t)a * K + blockIdx.x * 64 + c] = s;
    else db_hidden[blockIdx.x * 64 + c] = s;
  }
  if (blockIdx.x == 0 && tid < A) {                            // (after the barrier: sdq is complete)
    float s = 0.f;
    for (int m = 0; m < n; ++m) s += sdq[m * A + tid];
    db_out[tid] = s;
  }
}
}  // namespace
}  // namespace rlpyt

extern "C" int rlpyt_q_head_f32(const float* partial, int ksplit, const float* b_hidden,
                                const float* w_out, const float* b_out, int64_t n, int K, int A, float* q,
                                rlpyt_stream_t stream) {
  return rlpyt_q_head_train_f32(partial, ksplit, b_hidden, w_out, b_out, n, K, A, q, nullptr, stream);
}

// ... and the variant that also keeps the hidden activations h [n, K] for rlpyt_q_head_bwd_f32
extern "C" int rlpyt_q_head_train_f32(const float* partial, int ksplit, const float* b_hidden,
                                      const float* w_out, const float* b_out, int64_t n, int K, int A,
                                      float* q, float* h_out, rlpyt_stream_t stream) {
  RL_CHECK_ARG(partial && b_hidden && w_out && b_out && q, RLPYT_EINVAL, "rlpyt_q_head_f32: null pointer");
  RL_CHECK_ARG(n > 0 && ksplit > 0 && A > 0 && A <= 18 && (K == 512 || K == 256), RLPYT_ESHAPE,
               "rlpyt_q_head_f32: need n > 0, 0 < A <= 18, K in {256, 512} (K=%d A=%d)", K, A);
  hipStream_t s = (hipStream_t)stream;
  if (K == 512)
    RL_LAUNCH((rlpyt::q_head_kernel<2>), dim3((unsigned)n), dim3(256), 0, s, partial, ksplit, b_hidden,
              w_out, b_out, n, A, q, h_out);
  else
    RL_LAUNCH((rlpyt::q_head_kernel<1>), dim3((unsigned)n), dim3(256), 0, s, partial, ksplit, b_hidden,
              w_out, b_out, n, A, q, h_out);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_q_head_bwd_f32(const float* dq, const float* h, const float* w_out, int64_t n, int K,
                                    int A, float* dw_out, float* db_out, float* dh, float* db_hidden,
                                    rlpyt_stream_t stream) {
  RL_CHECK_ARG(dq && h && w_out && dw_out && db_out && dh && db_hidden, RLPYT_EINVAL,
               "rlpyt_q_head_bwd_f32: null pointer");
  RL_CHECK_ARG(n > 0 && n <= rlpyt::kQbRows && A > 0 && A <= rlpyt::kQbAmax && K > 0 && K % 64 == 0,
               RLPYT_ESHAPE, "rlpyt_q_head_bwd_f32: need 0 < n <= 256, 0 < A <= 18, K %% 64 == 0 (n=%d K=%d A=%d)",
               (int)n, K, A);
  RL_LAUNCH(rlpyt::q_head_bwd_kernel, dim3((unsigned)(K / 64)), dim3(rlpyt::kQbWaves * 64), 0, (hipStream_t)stream, dq, h,
            w_out, (int)n, K, A, dw_out, db_out, dh, db_hidden);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_rollout_fc_ksplit(int K) {
  return K > 0 ? (int)rlpyt::ceil_div(K, rlpyt::kRfcKc) : 0;
}

extern "C" int64_t rlpyt_rollout_fc_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)rlpyt_rollout_fc_ksplit(K) * M * N * (int64_t)sizeof(float);
}

extern "C" int rlpyt_rollout_fc_f32(const float* x, const float* w, float* partial, int M, int N,
                                    int K, rlpyt_stream_t stream) {
  RL_CHECK_ARG(x && w && partial, RLPYT_EINVAL, "rlpyt_rollout_fc_f32: null pointer");
  RL_CHECK_ARG(M > 0 && M <= 1024 && N > 0 && N % 64 == 0 && K > 0 && K % 16 == 0 &&
                   K <= rlpyt::kRfcKc * rlpyt::kRfcMaxSplit,
               RLPYT_ESHAPE,
               "rlpyt_rollout_fc_f32: need 0 < M <= 1024, N %% 64 == 0, K %% 16 == 0, K <= 4096 "
               "(M=%d N=%d K=%d)", M, N, K);
  RL_CHECK_ARG(RL_ALIGNED16(x) && RL_ALIGNED16(w) && RL_ALIGNED16(partial), RLPYT_ESHAPE,
               "rlpyt_rollout_fc_f32: buffers must be 16-byte aligned");
  const dim3 grid((unsigned)(N / 64), (unsigned)rlpyt_rollout_fc_ksplit(K),
                  (unsigned)rlpyt::ceil_div(M, 64));
  RL_LAUNCH(rlpyt::rollout_fc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, partial, M, N, K);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_rollout_head_f32(const float* partial, int ksplit, const float* fc_bias,
                                      const float* w_pi, const float* b_pi, const float* w_v,
                                      const float* b_v, const float* uniforms,
                                      const int64_t* t_dev, int64_t n, int K, int A,
                                      float* prob_rows, float* value_rows, int64_t* action_rows,
                                      int64_t B, int64_t lo, int64_t* action_out,
                                      float* bootstrap_out, rlpyt_stream_t stream) {
  RL_CHECK_ARG(partial && fc_bias && w_pi && b_pi && w_v && b_v, RLPYT_EINVAL,
               "rlpyt_rollout_head_f32: null pointer");
  RL_CHECK_ARG(bootstrap_out != nullptr ||
                   (uniforms && t_dev && prob_rows && value_rows && action_rows && action_out),
               RLPYT_EINVAL, "rlpyt_rollout_head_f32: null pointer (step mode needs every output)");
  RL_CHECK_ARG(n > 0 && ksplit > 0 && ksplit <= rlpyt::kRfcMaxSplit && A > 0 && A <= 8 &&
                   (K == 512 || K == 256) && lo >= 0 && (bootstrap_out != nullptr || lo + n <= B),
               RLPYT_ESHAPE, "rlpyt_rollout_head_f32: need 0<A<=8, K in {256,512}, ksplit<=32, lo+n<=B");
  const dim3 grid((unsigned)n), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (K == 512)
    RL_LAUNCH((rlpyt::rollout_head_kernel<2>), grid, block, 0, s, partial, ksplit, fc_bias, w_pi,
              b_pi, w_v, b_v, uniforms, t_dev, n, A, prob_rows, value_rows, action_rows, B, lo,
              action_out, bootstrap_out);
  else
    RL_LAUNCH((rlpyt::rollout_head_kernel<1>), grid, block, 0, s, partial, ksplit, fc_bias, w_pi,
              b_pi, w_v, b_v, uniforms, t_dev, n, A, prob_rows, value_rows, action_rows, B, lo,
              action_out, bootstrap_out);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
