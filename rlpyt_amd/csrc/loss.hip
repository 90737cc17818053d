// Fused forward+backward loss kernels (PPO / A2C / DQN) and advantage normalisation
// for gfx950.  Replaces ~15 elementwise/reduction torch ops per minibatch that the
// reference runs on CPU tensors (rlpyt/algos/pg/ppo.py:133-153, a2c.py:85-101,
// dqn/dqn.py:231-263, pg/base.py:65-73).
//
// Data movement: the [M,A] probability tile of a workgroup is staged through LDS with
// flat coalesced loads (a lane-per-sample read of A consecutive floats would issue A
// strided dword loads per wave); gradients leave through the same tile.  Reductions are
// deterministic: per-workgroup partials in f64 -> a one-workgroup finalize kernel.
#include "common.h"

namespace rlpyt {
namespace {

constexpr int kLossBlock = 256;
constexpr int kMaxLossGrid = 1024;
constexpr float kEpsCat = 1e-8f;  // rlpyt/distributions/categorical.py:9

constexpr int kMaxLossPart = 2048;  // partial rows: <= kMaxLossGrid workgroups, or one per wave of
                                    // the head kernel (512 workgroups x 4 waves)
struct LossWs {            // layout of the caller's workspace
  double valid_part[kMaxLossGrid];
  double part[kMaxLossPart][6];
};

__global__ __launch_bounds__(kLossBlock) void valid_partial_kernel(
    const float* __restrict__ valid, int64_t M, double* __restrict__ part) {
  __shared__ double scratch[16];
  double acc[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
       i += (int64_t)gridDim.x * blockDim.x)
    acc[0] += (double)valid[i];
  block_sum<1>(acc, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}

// Sum `n` partials (n <= kMaxLossGrid) redundantly inside a workgroup.
__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int n,
                                               double* scratch) {
  double acc[1] = {0.0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc[0] += part[i];
  block_sum<1>(acc, scratch);
  __shared__ double bcast;
  if (threadIdx.x == 0) bcast = acc[0];
  __syncthreads();
  return bcast;
}

// MODE 0: PPO clipped surrogate; MODE 1: A2C log-likelihood.
template <int MODE>
__global__ __launch_bounds__(kLossBlock) void pg_loss_kernel(
    const float* __restrict__ prob_new, const float* __restrict__ value,
    const float* __restrict__ prob_old, const int64_t* __restrict__ action,
    const float* __restrict__ advantage, const float* __restrict__ return_,
    const float* __restrict__ valid, int64_t M, int A, float ratio_clip, float c_v, float c_e,
    float* __restrict__ grad_prob, float* __restrict__ grad_value, LossWs* __restrict__ ws,
    int n_valid_part) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [kLossBlock * A]
  __shared__ double scratch[6 * 16];
  // Normaliser of valid_mean (utils/tensor.py:39-46): sum(valid) or M.
  double denom = (double)M;
  if (valid != nullptr) denom = sum_partials(ws->valid_part, n_valid_part, scratch);
  const float inv = (float)(1.0 / denom);

  double acc[6] = {0, 0, 0, 0, 0, 0};  // surr|logliA, verr, H, exp(H), count, unused
  const int64_t n_tiles = ceil_div(M, kLossBlock);
  for (int64_t tile_i = blockIdx.x; tile_i < n_tiles; tile_i += gridDim.x) {
    const int64_t m0 = tile_i * kLossBlock;
    const int rows = (int)min((int64_t)kLossBlock, M - m0);
    const int n_el = rows * A;
    __syncthreads();
    for (int e = threadIdx.x; e < n_el; e += kLossBlock) tile[e] = prob_new[m0 * A + e];
    __syncthreads();
    const int row = threadIdx.x;
    float g_sel = 0.f;   // extra gradient on the selected action's probability
    int a_sel = -1;
    float w = 0.f;
    if (row < rows) {
      const int64_t m = m0 + row;
      const float vmask = valid ? valid[m] : 1.0f;
      w = vmask * inv;
      a_sel = (int)action[m];
      const float adv = advantage[m];
      const float p_sel = tile[row * A + a_sel];
      float pi_term;
      if (MODE == 0) {
        // categorical.py:40-43 ; ppo.py:136-143
        const float den = prob_old[m * A + a_sel] + kEpsCat;
        const float ratio = (p_sel + kEpsCat) / den;
        const float lo = 1.0f - ratio_clip, hi = 1.0f + ratio_clip;
        const float clipped = fminf(fmaxf(ratio, lo), hi);
        const float s1 = ratio * adv, s2 = clipped * adv;
        pi_term = fminf(s1, s2);
        // d min(s1,s2)/d ratio with torch's tie rule (ties split the gradient) and
        // clamp's inclusive pass-through.
        const bool inside = (ratio >= lo) && (ratio <= hi);
        float dr;
        if (s1 < s2) dr = adv;
        else if (s1 > s2) dr = inside ? adv : 0.f;
        else dr = 0.5f * adv + (inside ? 0.5f * adv : 0.f);
        g_sel = -w * dr / den;
      } else {
        // categorical.py:36-38 ; a2c.py:87-88
        const float logli = logf(p_sel + kEpsCat);
        pi_term = logli * adv;
        g_sel = -w * adv / (p_sel + kEpsCat);
      }
      // value loss 0.5*(V-R)^2 (ppo.py:145-146)
      const float verr_d = value[m] - return_[m];
      const float verr = 0.5f * verr_d * verr_d;
      grad_value[m] = c_v * w * verr_d;
      // entropy -sum p log(p+eps) (categorical.py:32-34) and its gradient.
      float H = 0.f;
      for (int j = 0; j < A; ++j) {
        const float p = tile[row * A + j];
        const float lp = logf(p + kEpsCat);
        H -= p * lp;
        // d(-c_e * w * H)/dp_j = c_e * w * (log(p+eps) + p/(p+eps))
        float gj = c_e * w * (lp + p / (p + kEpsCat));
        if (j == a_sel) gj += g_sel;
        tile[row * A + j] = gj;
      }
      acc[0] += (double)(vmask * pi_term);
      acc[1] += (double)(vmask * verr);
      acc[2] += (double)(vmask * H);
      acc[3] += (double)(vmask * expf(H));
      acc[4] += (double)vmask;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_el; e += kLossBlock) grad_prob[m0 * A + e] = tile[e];
  }
  block_sum<6>(acc, scratch);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) ws->part[blockIdx.x][k] = acc[k];
  }
}

__global__ __launch_bounds__(kLossBlock) void pg_loss_finalize_kernel(
    const LossWs* __restrict__ ws, int n_part, int64_t M, int has_valid, int mode, float c_v,
    float c_e, float* __restrict__ out) {
  __shared__ double scratch[6 * 16];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] += ws->part[i][k];
  }
  block_sum<6>(acc, scratch);
  if (threadIdx.x == 0) {
    const double denom = has_valid ? acc[4] : (double)M;
    const float pi_loss = -(float)(acc[0] / denom);
    const float value_loss = c_v * (float)(acc[1] / denom);
    const float entropy = (float)(acc[2] / denom);
    const float perplexity = (float)(acc[3] / denom);
    const float entropy_loss = -c_e * entropy;
    out[0] = pi_loss + value_loss + entropy_loss;  // ppo.py:151
    out[1] = pi_loss;
    out[2] = value_loss;
    out[3] = entropy;
    out[4] = perplexity;
    (void)mode;
  }
}

// ---------------------------------------------------------------------------------------
// DQN TD / Huber loss (dqn.py:231-263).  One lane per sample; A <= a few dozen.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kLossBlock) void dqn_loss_kernel(
    const float* __restrict__ qs, const float* __restrict__ target_qs,
    const float* __restrict__ next_qs, const int64_t* __restrict__ action,
    const float* __restrict__ return_, const uint8_t* __restrict__ done_n,
    const float* __restrict__ is_weights, int64_t M, int A, float disc_n, float delta_clip,
    float* __restrict__ td_abs, float* __restrict__ grad_qs, LossWs* __restrict__ ws) {
  __shared__ double scratch[2 * 16];
  double acc[2] = {0, 0};
  const float invM = 1.0f / (float)M;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M;
       m += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)action[m];
    const float q = qs[m * A + a];
    float tq;
    if (next_qs != nullptr) {  // double DQN: argmax of the online net (first max wins)
      int best = 0;
      float bv = next_qs[m * A];
      for (int j = 1; j < A; ++j) {
        const float x = next_qs[m * A + j];
        if (x > bv) { bv = x; best = j; }
      }
      tq = target_qs[m * A + best];
    } else {
      tq = target_qs[m * A];
      for (int j = 1; j < A; ++j) tq = fmaxf(tq, target_qs[m * A + j]);
    }
    const float y = return_[m] + (1.0f - (float)(done_n[m] ? 1 : 0)) * (disc_n * tq);
    const float delta = y - q;
    const float ad = fabsf(delta);
    float loss = 0.5f * delta * delta;
    float dl_ddelta = delta;
    if (delta_clip > 0.f && !(ad <= delta_clip)) {
      loss = delta_clip * (ad - delta_clip / 2);
      dl_ddelta = delta_clip * (delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f));
    }
    const float isw = is_weights ? is_weights[m] : 1.0f;
    loss *= isw;
    td_abs[m] = delta_clip > 0.f ? fminf(fmaxf(ad, 0.f), delta_clip) : ad;
    for (int j = 0; j < A; ++j) grad_qs[m * A + j] = (j == a) ? (-invM * isw * dl_ddelta) : 0.f;
    acc[0] += (double)loss;
    acc[1] += (double)ad;
  }
  block_sum<2>(acc, scratch);
  if (threadIdx.x == 0) {
    ws->part[blockIdx.x][0] = acc[0];
    ws->part[blockIdx.x][1] = acc[1];
  }
}

__global__ __launch_bounds__(kLossBlock) void dqn_loss_finalize_kernel(
    const LossWs* __restrict__ ws, int n_part, int64_t M, float* __restrict__ out) {
  __shared__ double scratch[2 * 16];
  double acc[2] = {0, 0};
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
    acc[0] += ws->part[i][0];
    acc[1] += ws->part[i][1];
  }
  block_sum<2>(acc, scratch);
  if (threadIdx.x == 0) {
    out[0] = (float)(acc[0] / (double)M);
    out[1] = (float)(acc[1] / (double)M);
  }
}

// ---------------------------------------------------------------------------------------
// Advantage normalisation (pg/base.py:65-73): masked mean, UNBIASED std, in place.
// Three deterministic passes; partials in f64.
// ---------------------------------------------------------------------------------------
constexpr int kNormMaxGrid = 1024;
struct NormWs {
  double sum[kNormMaxGrid];
  double cnt[kNormMaxGrid];
  double ssq[kNormMaxGrid];
};

__global__ __launch_bounds__(256) void norm_pass1_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ valid,
                                                         int64_t n, NormWs* __restrict__ ws) {
  __shared__ double scratch[2 * 16];
  double acc[2] = {0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const bool ok = valid ? (valid[i] > 0.f) : true;
    if (ok) { acc[0] += (double)x[i]; acc[1] += 1.0; }
  }
  block_sum<2>(acc, scratch);
  if (threadIdx.x == 0) { ws->sum[blockIdx.x] = acc[0]; ws->cnt[blockIdx.x] = acc[1]; }
}

__global__ __launch_bounds__(256) void norm_pass2_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ valid,
                                                         int64_t n, NormWs* __restrict__ ws,
                                                         int n_part) {
  __shared__ double scratch[2 * 16];
  __shared__ double s_mean;
  double acc[2] = {0, 0};
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) { acc[0] += ws->sum[i]; acc[1] += ws->cnt[i]; }
  block_sum<2>(acc, scratch);
  if (threadIdx.x == 0) s_mean = acc[0] / acc[1];
  __syncthreads();
  // torch computes mean in fp32 and the deviations against that fp32 mean.
  const float mean = (float)s_mean;
  double ss[1] = {0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const bool ok = valid ? (valid[i] > 0.f) : true;
    if (ok) { const double dlt = (double)x[i] - (double)mean; ss[0] += dlt * dlt; }
  }
  block_sum<1>(ss, scratch);
  if (threadIdx.x == 0) ws->ssq[blockIdx.x] = ss[0];
}

__global__ __launch_bounds__(256) void norm_pass3_kernel(float* __restrict__ x, int64_t n,
                                                         const NormWs* __restrict__ ws,
                                                         int n_part, float eps,
                                                         float* __restrict__ stats_out) {
  __shared__ double scratch[3 * 16];
  __shared__ float s_mean, s_den;
  double acc[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
    acc[0] += ws->sum[i]; acc[1] += ws->cnt[i]; acc[2] += ws->ssq[i];
  }
  block_sum<3>(acc, scratch);
  if (threadIdx.x == 0) {
    const float mean = (float)(acc[0] / acc[1]);
    const float stdv = (float)sqrt(acc[2] / (acc[1] - 1.0));  // unbiased (torch .std())
    // Python max(adv_std, 1e-6): NaN std stays NaN.
    const float den = (eps > stdv) ? eps : stdv;
    s_mean = mean; s_den = den;
    if (stats_out != nullptr && blockIdx.x == 0) {
      stats_out[0] = mean; stats_out[1] = stdv; stats_out[2] = (float)acc[1];
    }
  }
  __syncthreads();
  const float mean = s_mean, den = s_den;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    x[i] = (x[i] - mean) / den;
}


// ======================================================================================
// PPO loss with the policy / value heads fused in (rlpyt/models/pg/atari_ff_model.py:56-58 +
// rlpyt/algos/pg/ppo.py:133-153, forward AND backward):
//   logits = h Wpi^T + bpi ; pi = softmax(logits) ; v = h Wv^T + bv ; PPO loss(pi, v, ...)
//   -> loss scalars, dL/dh [M,K], per-wave partial sums of dL/dWpi, dL/dbpi, dL/dWv, dL/dbv.
// Replaces 2 forward GEMMs, softmax, the prob-level loss kernel, softmax backward, 4 backward
// GEMMs and 2 bias reductions by one pass over h (16 B of HBM traffic per element of h) plus a
// partial-sum reduction.  One wave per row: lane l owns h[l + 64 i]; the (A+1) x K head weights
// and the (A+1) x K weight-gradient accumulators stay in VGPRs for the wave's lifetime.
// Loss arithmetic is statement-for-statement that of pg_loss_kernel<0> above.
// ======================================================================================
constexpr int kHeadAMax = 8;
constexpr int kHeadWavesPerBlock = 4;

template <int KI, int AM, bool TB>  // K = 64 * KI; AM = action slots held in registers (>= A): 4, 6
                                    // or 8; TB = trunk bias + ReLU fused in (compile-time: as a runtime
                                    // flag its selects cost 18 us at M = 8192)
// Two 4-wave workgroups per CU need <= 256 VGPRs per wave: with AM = 8 for every A the 2 x 8 x KI
// weight and weight-gradient registers pushed the kernel to 1 wave per SIMD (measured 2x slower).
__global__ __launch_bounds__(64 * kHeadWavesPerBlock, 2) void ppo_head_loss_kernel(
    const float* __restrict__ h, const float* __restrict__ w_pi, const float* __restrict__ b_pi,
    const float* __restrict__ w_v, const float* __restrict__ b_v,
    const float* __restrict__ prob_old, const int64_t* __restrict__ action,
    const float* __restrict__ advantage, const float* __restrict__ return_,
    const float* __restrict__ valid, int64_t M, int A, float ratio_clip, float c_v, float c_e,
    float* __restrict__ grad_h, float* __restrict__ wpart, LossWs* __restrict__ ws,
    int n_valid_part, const int64_t* __restrict__ flat_idx, int T, int64_t B,
    const float* __restrict__ trunk_bias, const float* __restrict__ ratio_clip_dev) {
  // ratio_clip_dev != NULL (captured update graphs): the clip range of this update is read from
  // device memory, so that one captured launch serves the whole schedule
  if (ratio_clip_dev != nullptr) ratio_clip = *ratio_clip_dev;
  // trunk_bias != NULL: ``h`` is the trunk's PRE-activation without its bias, z = x W^T; the kernel
  // applies h = relu(z + b) while loading the row, returns dL/dz (masked by h > 0) in grad_h and
  // the bias gradient sum_m dL/dz[m] as K more floats of the partial row -- the trunk's bias add,
  // ReLU, ReLU backward and bias-gradient reduction (4 launches, ~45 us at M = 8192) disappear.
  __shared__ double scratch[6 * 16];
  constexpr int K = 64 * KI;
  constexpr bool has_tb = TB;
  double denom = (double)M;
  if (valid != nullptr) denom = sum_partials(ws->valid_part, n_valid_part, scratch);
  const float inv = (float)(1.0 / denom);
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * kHeadWavesPerBlock + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * kHeadWavesPerBlock;

  float wp[AM][KI], wv[KI], gw[AM][KI], gwv[KI];
#pragma unroll
  for (int i = 0; i < KI; ++i) {
#pragma unroll
    for (int a = 0; a < AM; ++a) {
      // unconditional clamped load + select: predicated loads compile to one branch and one
      // s_waitcnt vmcnt(0) each, i.e. ~60 serialized L2 round trips per wave
      const float wl = w_pi[min(a, A - 1) * K + lane + 64 * i];
      wp[a][i] = a < A ? wl : 0.f;
      gw[a][i] = 0.f;
    }
    wv[i] = w_v[lane + 64 * i];
    gwv[i] = 0.f;
  }
  float bp[AM], gb[AM];
#pragma unroll
  for (int a = 0; a < AM; ++a) {
    const float bl = b_pi[min(a, A - 1)];
    bp[a] = a < A ? bl : 0.f;
    gb[a] = 0.f;
  }
  const float bv = b_v[0];
  float gbv = 0.f;
  float tbv[KI], gtb[KI];
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    tbv[i] = has_tb ? trunk_bias[lane + 64 * i] : 0.f;
    gtb[i] = 0.f;
  }
  double acc[5] = {0, 0, 0, 0, 0};  // surrogate, value err, H, exp(H), count

  // Software pipeline over the wave's rows.  A row's critical path used to hold THREE dependent
  // global loads (flat_idx[m] -> action[r] -> prob_old[r, a_sel]: ~5 us per row, the whole kernel
  // ran at 5 % of HBM peak): now the row index is fetched two rows ahead and every per-sample
  // scalar (action, advantage, return, valid, the whole prob_old row) one row ahead, together
  // with the next row of h, so a row only waits for arithmetic.
  struct RowIn {
    int64_t r;
    int a_sel;
    float adv, ret, vmask, po[AM];
  };
  auto row_of = [&](int64_t m) -> int64_t {      // [T,B] row of sample m (ppo.py:94-95)
    if (flat_idx == nullptr) return m;
    const int64_t idx = flat_idx[m];
    return (idx % T) * B + (idx / T);
  };
  auto fetch_row = [&](int64_t r, RowIn& in) {
    in.r = r;
    in.a_sel = (int)action[r];
    in.adv = advantage[r];
    in.ret = return_[r];
    in.vmask = valid ? valid[r] : 1.0f;
#pragma unroll
    for (int a = 0; a < AM; ++a) in.po[a] = prob_old[r * A + min(a, A - 1)];
  };
  float hn[KI];   // next row of h
  RowIn nx{};
  int64_t r_nn = 0;  // row index of the row after next
  if (wave_id < M) {
#pragma unroll
    for (int i = 0; i < KI; ++i) hn[i] = h[wave_id * K + lane + 64 * i];
    fetch_row(row_of(wave_id), nx);
    if (wave_id + n_waves < M) r_nn = row_of(wave_id + n_waves);
  }
  for (int64_t m = wave_id; m < M; m += n_waves) {
    float hv[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) hv[i] = has_tb ? fmaxf(hn[i] + tbv[i], 0.f) : hn[i];
    const RowIn in = nx;
    if (m + n_waves < M) {
#pragma unroll
      for (int i = 0; i < KI; ++i) hn[i] = h[(m + n_waves) * K + lane + 64 * i];
      fetch_row(r_nn, nx);
      if (m + 2 * n_waves < M) r_nn = row_of(m + 2 * n_waves);
    }
    float lg[AM], vsum = 0.f;
#pragma unroll
    for (int a = 0; a < AM; ++a) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < KI; ++i) t = fmaf(hv[i], wp[a][i], t);
      lg[a] = t;
    }
#pragma unroll
    for (int i = 0; i < KI; ++i) vsum = fmaf(hv[i], wv[i], vsum);
#pragma unroll
    for (int a = 0; a < AM; ++a) lg[a] = wave_sum(lg[a]) + bp[a];
    const float val = wave_sum(vsum) + bv;
    // softmax (every lane redundantly: the scalars are needed by all lanes below)
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < AM; ++a)
      if (a < A) mx = fmaxf(mx, lg[a]);
    float p[AM], den = 0.f;
#pragma unroll
    for (int a = 0; a < AM; ++a) {
      p[a] = a < A ? expf(lg[a] - mx) : 0.f;
      den += p[a];
    }
    const float rden = 1.f / den;
#pragma unroll
    for (int a = 0; a < AM; ++a) p[a] *= rden;
    // ---- PPO loss terms and dL/dp, dL/dv (as pg_loss_kernel<0>)
    const float vmask = in.vmask;
    const float w = vmask * inv;
    const int a_sel = in.a_sel;
    const float adv = in.adv;
    float p_sel = 0.f, po_sel = 0.f;
#pragma unroll
    for (int a = 0; a < AM; ++a)
      if (a == a_sel) {
        p_sel = p[a];
        po_sel = in.po[a];
      }
    const float den_o = po_sel + kEpsCat;
    const float ratio = (p_sel + kEpsCat) / den_o;
    const float lo = 1.0f - ratio_clip, hi = 1.0f + ratio_clip;
    const float clipped = fminf(fmaxf(ratio, lo), hi);
    const float s1 = ratio * adv, s2 = clipped * adv;
    const float pi_term = fminf(s1, s2);
    const bool inside = (ratio >= lo) && (ratio <= hi);
    float dr;
    if (s1 < s2) dr = adv;
    else if (s1 > s2) dr = inside ? adv : 0.f;
    else dr = 0.5f * adv + (inside ? 0.5f * adv : 0.f);
    const float g_sel = -w * dr / den_o;
    const float verr_d = val - in.ret;
    const float verr = 0.5f * verr_d * verr_d;
    const float dv = c_v * w * verr_d;
    float H = 0.f, gp[AM], dot = 0.f;
#pragma unroll
    for (int a = 0; a < AM; ++a) {
      gp[a] = 0.f;
      if (a < A) {
        const float lp = logf(p[a] + kEpsCat);
        H -= p[a] * lp;
        gp[a] = c_e * w * (lp + p[a] / (p[a] + kEpsCat));
        if (a == a_sel) gp[a] += g_sel;
        dot += gp[a] * p[a];
      }
    }
    // softmax backward: dL/dlogit_a = p_a (g_a - sum_b g_b p_b)
    float dl[AM];
#pragma unroll
    for (int a = 0; a < AM; ++a) dl[a] = p[a] * (gp[a] - dot);
    if (lane == 0) {
      acc[0] += (double)(vmask * pi_term);
      acc[1] += (double)(vmask * verr);
      acc[2] += (double)(vmask * H);
      acc[3] += (double)(vmask * expf(H));
      acc[4] += (double)vmask;
    }
    // ---- dL/dh and weight-gradient accumulation
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      float g = dv * wv[i];
#pragma unroll
      for (int a = 0; a < AM; ++a) {
        g = fmaf(dl[a], wp[a][i], g);
        gw[a][i] = fmaf(dl[a], hv[i], gw[a][i]);
      }
      gwv[i] = fmaf(dv, hv[i], gwv[i]);
      if (has_tb) {
        g = hv[i] > 0.f ? g : 0.f;
        gtb[i] += g;
      }
      grad_h[m * K + lane + 64 * i] = g;
    }
#pragma unroll
    for (int a = 0; a < AM; ++a) gb[a] += dl[a];
    gbv += dv;
  }
  // weight-gradient partials [A*K dWpi | K dWv | A dbpi | 1 dbv]: waves 1..3 hand theirs to
  // wave 0 through LDS, one partial row per workgroup leaves for the reduction kernel
  extern __shared__ float hred[];   // [3][(AM + 2) * K + AM + 1]
  constexpr int kRedStride = (AM + 2) * K + AM + 1;
  constexpr int kTbOff = (AM + 1) * K + AM + 1;   // trunk-bias gradient inside an LDS row
  const int wv_i = threadIdx.x >> 6;
  if (wv_i > 0) {
    float* r = hred + (wv_i - 1) * kRedStride;
#pragma unroll
    for (int i = 0; i < KI; ++i) {
#pragma unroll
      for (int a = 0; a < AM; ++a) r[a * K + lane + 64 * i] = gw[a][i];
      r[AM * K + lane + 64 * i] = gwv[i];
      r[kTbOff + lane + 64 * i] = gtb[i];
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < AM; ++a) r[(AM + 1) * K + a] = gb[a];
      r[(AM + 1) * K + AM] = gbv;
    }
  }
  __syncthreads();
  if (wv_i == 0) {
    const int part = A * K + K + A + 1 + (has_tb ? K : 0);
    float* out = wpart + (int64_t)blockIdx.x * part;
    if (has_tb) {
#pragma unroll
      for (int i = 0; i < KI; ++i) {
        float v = gtb[i];
#pragma unroll
        for (int w3 = 0; w3 < 3; ++w3) v += hred[w3 * kRedStride + kTbOff + lane + 64 * i];
        out[A * K + K + A + 1 + lane + 64 * i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < KI; ++i) {
#pragma unroll
      for (int a = 0; a < AM; ++a) {
        if (a < A) {
          float v = gw[a][i];
#pragma unroll
          for (int w3 = 0; w3 < 3; ++w3) v += hred[w3 * kRedStride + a * K + lane + 64 * i];
          out[a * K + lane + 64 * i] = v;
        }
      }
      float v = gwv[i];
#pragma unroll
      for (int w3 = 0; w3 < 3; ++w3) v += hred[w3 * kRedStride + AM * K + lane + 64 * i];
      out[A * K + lane + 64 * i] = v;
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < AM; ++a) {
        if (a < A) {
          float v = gb[a];
#pragma unroll
          for (int w3 = 0; w3 < 3; ++w3) v += hred[w3 * kRedStride + (AM + 1) * K + a];
          out[A * K + K + a] = v;
        }
      }
      float v = gbv;
#pragma unroll
      for (int w3 = 0; w3 < 3; ++w3) v += hred[w3 * kRedStride + (AM + 1) * K + AM];
      out[A * K + K + A] = v;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) ws->part[wave_id][k] = acc[k];
    ws->part[wave_id][5] = 0.0;
  }
}

// Weight-gradient partial rows: out[e] = sum over the per-workgroup partial rows (fixed order -> deterministic).
// 64 elements per workgroup, the rows split over 16 waves with 4 independent loads in flight each:
// the reduction is a chain of dependent 256-byte loads per wave (round 2 start: 4 waves x 2 in
// flight = 64 dependent rounds, 21 us for 7 MB); now 8 rounds.
// ... and the loss scalars in the SAME launch (round 2: head_reduce_kernel + pg_loss_finalize_kernel):
// workgroups [0, nb) reduce the partial rows, the LAST workgroup sums the per-wave loss partials
// (f64, fixed order) and writes the five scalars -- the arithmetic of pg_loss_finalize_kernel, one
// launch (~6 us on a 97 %-busy update) less.
__global__ __launch_bounds__(1024) void head_reduce_finalize_kernel(
    const float* __restrict__ wpart, int n_rows, int part, float* __restrict__ out_w,
    const LossWs* __restrict__ ws, int n_part, int64_t M, int has_valid, float c_v, float c_e,
    float* __restrict__ out_scalars) {
  __shared__ float red[16][64];
  __shared__ double scratch[6 * 16];
  if (blockIdx.x == gridDim.x - 1) {
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += ws->part[i][k];
    }
    block_sum<6>(acc, scratch);
    if (threadIdx.x == 0) {
      const double denom = has_valid ? acc[4] : (double)M;
      const float pi_loss = -(float)(acc[0] / denom);
      const float value_loss = c_v * (float)(acc[1] / denom);
      const float entropy = (float)(acc[2] / denom);
      const float perplexity = (float)(acc[3] / denom);
      out_scalars[0] = pi_loss + value_loss - c_e * entropy;  // ppo.py:151
      out_scalars[1] = pi_loss;
      out_scalars[2] = value_loss;
      out_scalars[3] = entropy;
      out_scalars[4] = perplexity;
    }
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (e < part) {
    int g = wave;
    for (; g + 48 < n_rows; g += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += wpart[(int64_t)(g + 16 * u) * part + e];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (g + 16 * u < n_rows) s[u] += wpart[(int64_t)(g + 16 * u) * part + e];
  }
  red[wave][lane] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (wave == 0 && e < part) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][lane];
    out_w[e] = v;
  }
}

constexpr int kHeadGrid = 512;  // one weight-gradient partial row per workgroup; two 4-wave
                                // workgroups per CU (the kernel is latency-bound: 8 waves per CU halve it)
}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

extern "C" int64_t rlpyt_pg_loss_workspace_bytes(int64_t M) {
  (void)M;
  return (int64_t)sizeof(LossWs);
}
extern "C" int64_t rlpyt_adv_normalize_workspace_bytes(int64_t n) {
  (void)n;
  return (int64_t)sizeof(NormWs);
}

namespace {
template <int MODE>
int launch_pg_loss(const float* prob_new, const float* value, const float* prob_old,
                   const int64_t* action, const float* advantage, const float* return_,
                   const float* valid, int64_t M, int A, float ratio_clip, float c_v, float c_e,
                   float* out_scalars, float* grad_prob, float* grad_value, void* workspace,
                   hipStream_t s) {
  LossWs* ws = reinterpret_cast<LossWs*>(workspace);
  const int grid = (int)std::min<int64_t>(ceil_div(M, kLossBlock), kMaxLossGrid);
  int n_valid_part = 0;
  if (valid != nullptr) {
    n_valid_part = grid;
    RL_LAUNCH(valid_partial_kernel, dim3(grid), dim3(kLossBlock), 0, s, valid, M,
                       ws->valid_part);
    RL_LAUNCH_CHECK();
  }
  const size_t lds = (size_t)kLossBlock * A * sizeof(float);
  RL_LAUNCH((pg_loss_kernel<MODE>), dim3(grid), dim3(kLossBlock), lds, s, prob_new,
                     value, prob_old, action, advantage, return_, valid, M, A, ratio_clip, c_v,
                     c_e, grad_prob, grad_value, ws, n_valid_part);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(pg_loss_finalize_kernel, dim3(1), dim3(kLossBlock), 0, s, ws, grid, M,
                     valid != nullptr ? 1 : 0, MODE, c_v, c_e, out_scalars);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
}  // namespace

extern "C" int rlpyt_ppo_loss_fwd_bwd_f32(const float* prob_new, const float* value,
                                          const float* prob_old, const int64_t* action,
                                          const float* advantage, const float* return_,
                                          const float* valid, int64_t M, int A,
                                          float ratio_clip, float value_loss_coeff,
                                          float entropy_loss_coeff, float* out_scalars,
                                          float* grad_prob, float* grad_value, void* workspace,
                                          rlpyt_stream_t stream) {
  RL_CHECK_ARG(prob_new && value && prob_old && action && advantage && return_ && out_scalars &&
                   grad_prob && grad_value && workspace,
               RLPYT_EINVAL, "rlpyt_ppo_loss_fwd_bwd_f32: null pointer");
  RL_CHECK_ARG(M > 0 && A > 0 && A <= 160, RLPYT_ESHAPE,
               "rlpyt_ppo_loss_fwd_bwd_f32: need M>0, 0<A<=160 (M=%ld A=%d)", (long)M, A);
  return launch_pg_loss<0>(prob_new, value, prob_old, action, advantage, return_, valid, M, A,
                           ratio_clip, value_loss_coeff, entropy_loss_coeff, out_scalars,
                           grad_prob, grad_value, workspace, (hipStream_t)stream);
}

extern "C" int rlpyt_a2c_loss_fwd_bwd_f32(const float* prob, const float* value,
                                          const int64_t* action, const float* advantage,
                                          const float* return_, const float* valid, int64_t M,
                                          int A, float value_loss_coeff,
                                          float entropy_loss_coeff, float* out_scalars,
                                          float* grad_prob, float* grad_value, void* workspace,
                                          rlpyt_stream_t stream) {
  RL_CHECK_ARG(prob && value && action && advantage && return_ && out_scalars && grad_prob &&
                   grad_value && workspace,
               RLPYT_EINVAL, "rlpyt_a2c_loss_fwd_bwd_f32: null pointer");
  RL_CHECK_ARG(M > 0 && A > 0 && A <= 160, RLPYT_ESHAPE,
               "rlpyt_a2c_loss_fwd_bwd_f32: need M>0, 0<A<=160 (M=%ld A=%d)", (long)M, A);
  return launch_pg_loss<1>(prob, value, nullptr, action, advantage, return_, valid, M, A, 0.f,
                           value_loss_coeff, entropy_loss_coeff, out_scalars, grad_prob,
                           grad_value, workspace, (hipStream_t)stream);
}

extern "C" int rlpyt_dqn_loss_fwd_bwd_f32(const float* qs, const float* target_qs,
                                          const float* next_qs, const int64_t* action,
                                          const float* return_, const uint8_t* done_n,
                                          const float* is_weights, int64_t M, int A,
                                          float disc_n, float delta_clip, float* out_scalars,
                                          float* td_abs, float* grad_qs, void* workspace,
                                          rlpyt_stream_t stream) {
  RL_CHECK_ARG(qs && target_qs && action && return_ && done_n && out_scalars && td_abs &&
                   grad_qs && workspace,
               RLPYT_EINVAL, "rlpyt_dqn_loss_fwd_bwd_f32: null pointer");
  RL_CHECK_ARG(M > 0 && A > 0, RLPYT_ESHAPE, "rlpyt_dqn_loss_fwd_bwd_f32: need M>0, A>0");
  LossWs* ws = reinterpret_cast<LossWs*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  const int grid = (int)std::min<int64_t>(ceil_div(M, kLossBlock), kMaxLossGrid);
  RL_LAUNCH(dqn_loss_kernel, dim3(grid), dim3(kLossBlock), 0, s, qs, target_qs, next_qs,
                     action, return_, done_n, is_weights, M, A, disc_n, delta_clip, td_abs,
                     grad_qs, ws);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(dqn_loss_finalize_kernel, dim3(1), dim3(kLossBlock), 0, s, ws, grid, M,
                     out_scalars);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_adv_normalize_f32(float* advantage, const float* valid, int64_t n,
                                       float eps, void* workspace, float* stats_out,
                                       rlpyt_stream_t stream) {
  RL_CHECK_ARG(advantage && workspace, RLPYT_EINVAL, "rlpyt_adv_normalize_f32: null pointer");
  RL_CHECK_ARG(n >= 0, RLPYT_EINVAL, "rlpyt_adv_normalize_f32: negative n");
  if (n == 0) return RLPYT_OK;
  NormWs* ws = reinterpret_cast<NormWs*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  const int grid = (int)std::min<int64_t>(ceil_div(n, 256 * 4), kNormMaxGrid);
  RL_LAUNCH(norm_pass1_kernel, dim3(grid), dim3(256), 0, s, advantage, valid, n, ws);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(norm_pass2_kernel, dim3(grid), dim3(256), 0, s, advantage, valid, n, ws,
                     grid);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(norm_pass3_kernel, dim3(grid), dim3(256), 0, s, advantage, n, ws, grid, eps,
                     stats_out);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int64_t rlpyt_ppo_head_loss_workspace_bytes(int K, int A) {
  if (K <= 0 || A <= 0) return 0;
  const int64_t part = (int64_t)A * K + K + A + 1 + K;   // (+ K: trunk-bias gradient)
  // LossWs | per-wave weight-gradient partials
  return (int64_t)((sizeof(LossWs) + 255) / 256 * 256) +
         (int64_t)kHeadGrid * part * (int64_t)sizeof(float);
}

extern "C" int rlpyt_ppo_trunk_head_loss_fwd_bwd_f32(
    const float* h, const float* trunk_bias, const float* w_pi, const float* b_pi,
    const float* w_v, const float* b_v, const float* prob_old, const int64_t* action,
    const float* advantage, const float* return_, const float* valid, const int64_t* flat_idx,
    int T, int64_t B, int64_t M, int K, int A, float ratio_clip, float value_loss_coeff,
    float entropy_loss_coeff, float* out_scalars, float* grad_h, float* grad_params,
    void* workspace, rlpyt_stream_t stream) {
  return rlpyt_ppo_trunk_head_loss_fwd_bwd_dev_f32(
      h, trunk_bias, w_pi, b_pi, w_v, b_v, prob_old, action, advantage, return_, valid, flat_idx, T,
      B, M, K, A, ratio_clip, nullptr, value_loss_coeff, entropy_loss_coeff, out_scalars, grad_h,
      grad_params, workspace, stream);
}

extern "C" int rlpyt_ppo_trunk_head_loss_fwd_bwd_dev_f32(
    const float* h, const float* trunk_bias, const float* w_pi, const float* b_pi,
    const float* w_v, const float* b_v, const float* prob_old, const int64_t* action,
    const float* advantage, const float* return_, const float* valid, const int64_t* flat_idx,
    int T, int64_t B, int64_t M, int K, int A, float ratio_clip, const float* ratio_clip_dev,
    float value_loss_coeff, float entropy_loss_coeff, float* out_scalars, float* grad_h,
    float* grad_params, void* workspace, rlpyt_stream_t stream) {
  RL_CHECK_ARG(flat_idx == nullptr || (T > 0 && B > 0), RLPYT_EINVAL,
               "rlpyt_ppo_trunk_head_loss_fwd_bwd_f32: flat_idx needs T, B");
  RL_CHECK_ARG(flat_idx == nullptr || valid == nullptr, RLPYT_ESHAPE,
               "rlpyt_ppo_head_loss_fwd_bwd_f32: index mode takes valid == NULL (gather it first)");
  RL_CHECK_ARG(h && w_pi && b_pi && w_v && b_v && prob_old && action && advantage && return_ &&
                   out_scalars && grad_h && grad_params && workspace,
               RLPYT_EINVAL, "rlpyt_ppo_head_loss_fwd_bwd_f32: null pointer");
  RL_CHECK_ARG(M > 0 && A > 0 && A <= kHeadAMax && (K == 512 || K == 256), RLPYT_ESHAPE,
               "rlpyt_ppo_head_loss_fwd_bwd_f32: need M>0, 0<A<=8, K in {256,512} (M=%ld K=%d A=%d)",
               (long)M, K, A);
  hipStream_t s = (hipStream_t)stream;
  LossWs* ws = reinterpret_cast<LossWs*>(workspace);
  float* wpart = reinterpret_cast<float*>(static_cast<char*>(workspace) +
                                          (sizeof(LossWs) + 255) / 256 * 256);
  int n_valid_part = 0;
  if (valid != nullptr) {
    n_valid_part = (int)std::min<int64_t>(ceil_div(M, kLossBlock), kMaxLossGrid);
    RL_LAUNCH(valid_partial_kernel, dim3(n_valid_part), dim3(kLossBlock), 0, s, valid, M,
                       ws->valid_part);
    RL_LAUNCH_CHECK();
  }
  const int grid = (int)std::min<int64_t>(ceil_div(M, kHeadWavesPerBlock), kHeadGrid);
  const int n_waves = grid * kHeadWavesPerBlock;
  const int part = A * K + K + A + 1 + (trunk_bias != nullptr ? K : 0);
  const size_t lds = (size_t)3 * ((kHeadAMax + 2) * K + kHeadAMax + 1) * sizeof(float);
#define RL_HEAD_TB(KI_, AM_, TB_)                                                                 \
  RL_LAUNCH((ppo_head_loss_kernel<KI_, AM_, TB_>), dim3(grid), dim3(64 * kHeadWavesPerBlock), lds, \
            s, h, w_pi, b_pi, w_v, b_v, prob_old, action, advantage, return_, valid, M, A,          \
            ratio_clip, value_loss_coeff, entropy_loss_coeff, grad_h, wpart, ws, n_valid_part,      \
            flat_idx, T, B, trunk_bias, ratio_clip_dev)
#define RL_HEAD(KI_, AM_)                                                                         \
  do {                                                                                            \
    if (trunk_bias != nullptr) RL_HEAD_TB(KI_, AM_, true); else RL_HEAD_TB(KI_, AM_, false);        \
  } while (0)
  if (K == 512) {
    if (A <= 4) RL_HEAD(8, 4); else if (A <= 6) RL_HEAD(8, 6); else RL_HEAD(8, 8);
  } else {
    if (A <= 4) RL_HEAD(4, 4); else if (A <= 6) RL_HEAD(4, 6); else RL_HEAD(4, 8);
  }
#undef RL_HEAD
#undef RL_HEAD_TB
  RL_LAUNCH_CHECK();
  RL_LAUNCH(head_reduce_finalize_kernel, dim3((part + 63) / 64 + 1), dim3(1024), 0, s, wpart, grid,
            part, grad_params, ws, n_waves, M, valid != nullptr ? 1 : 0, value_loss_coeff,
            entropy_loss_coeff, out_scalars);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_ppo_head_loss_fwd_bwd_f32(
    const float* h, const float* w_pi, const float* b_pi, const float* w_v, const float* b_v,
    const float* prob_old, const int64_t* action, const float* advantage, const float* return_,
    const float* valid, const int64_t* flat_idx, int T, int64_t B, int64_t M, int K, int A,
    float ratio_clip, float value_loss_coeff, float entropy_loss_coeff, float* out_scalars,
    float* grad_h, float* grad_params, void* workspace, rlpyt_stream_t stream) {
  return rlpyt_ppo_trunk_head_loss_fwd_bwd_f32(h, nullptr, w_pi, b_pi, w_v, b_v, prob_old, action,
                                               advantage, return_, valid, flat_idx, T, B, M, K, A,
                                               ratio_clip, value_loss_coeff, entropy_loss_coeff,
                                               out_scalars, grad_h, grad_params, workspace, stream);
}
