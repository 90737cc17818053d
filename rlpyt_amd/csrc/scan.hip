// Return / advantage scans over [T, N] trajectories for gfx950 (MI355X).
//
// Replaces the Python time loops of rlpyt/algos/utils.py:8-40,67-112 (reference).
// Layout: row-major [T, N]; lanes own columns (coalesced along N = Batch), the T axis
// is walked sequentially backwards in register-staged chunks so that each lane keeps
// TT independent rows of loads in flight per wait.
//
// This translation unit is compiled with -ffp-contract=off: the EXACT variants must
// keep the reference's fp32 association with separate mul/add.
#include "common.h"

namespace rlpyt {
namespace {

template <int VEC>
__device__ __forceinline__ void load_f(const float* __restrict__ p, float (&o)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w;
  } else if constexpr (VEC == 2) {
    const float2 x = *reinterpret_cast<const float2*>(p);
    o[0] = x.x; o[1] = x.y;
  } else {
    o[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void store_f(float* __restrict__ p, const float (&o)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
  } else {
    *p = o[0];
  }
}
template <int VEC>
__device__ __forceinline__ void load_b(const uint8_t* __restrict__ p, uint8_t (&o)[VEC]) {
  if constexpr (VEC == 4) {
    const uint32_t x = *reinterpret_cast<const uint32_t*>(p);
    o[0] = x & 0xff; o[1] = (x >> 8) & 0xff; o[2] = (x >> 16) & 0xff; o[3] = x >> 24;
  } else if constexpr (VEC == 2) {
    const uint16_t x = *reinterpret_cast<const uint16_t*>(p);
    o[0] = x & 0xff; o[1] = x >> 8;
  } else {
    o[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void store_b(uint8_t* __restrict__ p, const uint8_t (&o)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<uint32_t*>(p) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) |
                                       ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<uint16_t*>(p) = (uint16_t)((uint16_t)o[0] | ((uint16_t)o[1] << 8));
  } else {
    *p = o[0];
  }
}

// First time index with done != 0 per owned column (T if none); a forward pass over the
// 1-byte mask only.  valid[t] = (t <= first_done)  <=>  1 - min(1, cumsum(done[:t])).
template <int VEC>
__device__ __forceinline__ void first_done_scan(const uint8_t* __restrict__ done, int T,
                                                int64_t N, int64_t col, int (&fd)[VEC]) {
#pragma unroll
  for (int k = 0; k < VEC; ++k) fd[k] = T;
  constexpr int U = 16;
  for (int t0 = 0; t0 < T; t0 += U) {
    uint8_t d[U][VEC];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      if (t0 + i < T) load_b<VEC>(done + (int64_t)(t0 + i) * N + col, d[i]);
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      if (t0 + i < T) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          if (d[i][k] && fd[k] == T) fd[k] = t0 + i;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// EXACT scans: MODE 0 = GAE, MODE 1 = discount-return (+ optional advantage = R - V).
// ---------------------------------------------------------------------------------------
template <int MODE, int VEC, int TT, bool VALID>
__global__ __launch_bounds__(256) void scan_exact_kernel(
    const float* __restrict__ reward, const float* __restrict__ value,
    const uint8_t* __restrict__ done, const float* __restrict__ bootstrap,
    float* __restrict__ advantage, float* __restrict__ return_, float* __restrict__ valid,
    int T, int64_t N, float g, float gl) {
  const int64_t col = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (col >= N) return;

  int fd[VEC];
  if constexpr (VALID) first_done_scan<VEC>(done, T, N, col, fd);

  float nextV[VEC];  // GAE: V[t+1];  discount: R[t+1]
  float nextA[VEC];  // GAE: A[t+1]
  load_f<VEC>(bootstrap + col, nextV);
#pragma unroll
  for (int k = 0; k < VEC; ++k) nextA[k] = 0.f;
  const bool want_v = (MODE == 0) || (value != nullptr && advantage != nullptr);

  for (int t0 = T; t0 > 0; t0 -= TT) {
    float r[TT][VEC], v[TT][VEC];
    uint8_t d[TT][VEC];
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int t = t0 - 1 - i;
      if (t >= 0) {
        const int64_t off = (int64_t)t * N + col;
        load_f<VEC>(reward + off, r[i]);
        if (want_v) load_f<VEC>(value + off, v[i]);
        load_b<VEC>(done + off, d[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int t = t0 - 1 - i;
      if (t >= 0) {
        const int64_t off = (int64_t)t * N + col;
        float a_out[VEC], r_out[VEC], m_out[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float nd = 1.0f - (float)d[i][k];
          if constexpr (MODE == 0) {
            // utils.py:36-38: delta = r + g*V[t+1]*nd - V ; A = delta + (g*l)*nd*A[t+1]
            const float delta = (r[i][k] + ((g * nextV[k]) * nd)) - v[i][k];
            const float a = (t == T - 1) ? delta : (delta + ((gl * nd) * nextA[k]));
            a_out[k] = a;
            r_out[k] = a + v[i][k];  // utils.py:39
            nextA[k] = a;
            nextV[k] = v[i][k];
          } else {
            // utils.py:18-20: R = r + R[t+1]*g*nd   (last row: r + g*bv*nd, same product)
            const float ret = r[i][k] + ((nextV[k] * g) * nd);
            r_out[k] = ret;
            nextV[k] = ret;
            if (want_v) a_out[k] = ret - v[i][k];  // pg/base.py:55
          }
          if constexpr (VALID) m_out[k] = (t <= fd[k]) ? 1.0f : 0.0f;
        }
        store_f<VEC>(return_ + off, r_out);
        if (MODE == 0 || want_v) store_f<VEC>(advantage + off, a_out);
        if constexpr (VALID) store_f<VEC>(valid + off, m_out);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// SEGMENTED scans (latency variant for small N): a workgroup owns 64 columns; wave w owns
// the time segment [w*L, (w+1)*L).  Phase 1: every wave reduces its segment to the affine
// map A_start = P + Q * A_in in registers (all loads of the segment in flight at once).
// Phase 2: the maps go through LDS; every wave folds the maps of the later segments to get
// its incoming carry (<= S-1 fused steps instead of T-L sequential ones).  Phase 3: the
// segment is replayed from registers with the carry and stored.  Re-associated => fp32
// tolerance, not bit-exact.
// ---------------------------------------------------------------------------------------
constexpr int kSegMaxL = 16;   // time steps per wave held in registers
constexpr int kSegMaxS = 16;   // waves (segments) per workgroup

template <int MODE, bool VALID>
__global__ __launch_bounds__(1024) void scan_segmented_kernel(
    const float* __restrict__ reward, const float* __restrict__ value,
    const uint8_t* __restrict__ done, const float* __restrict__ bootstrap,
    float* __restrict__ advantage, float* __restrict__ return_, float* __restrict__ valid,
    int T, int64_t N, int L, float g, float gl) {
  __shared__ float sP[kSegMaxS][kWave];
  __shared__ float sQ[kSegMaxS][kWave];
  __shared__ int sFd[kSegMaxS][kWave];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int S = blockDim.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * kWave + lane;
  const bool active = col < N;
  const int tb = w * L;                        // first step of my segment
  const int te = min(T, tb + L);               // one past the last
  const bool want_v = (MODE == 0) || (value != nullptr && advantage != nullptr);

  float r[kSegMaxL], v[kSegMaxL], nd[kSegMaxL];
  float vnext = 0.f;  // V just after my segment (GAE) -- bootstrap for the last one
  int myfd = T;
  if (active) {
#pragma unroll
    for (int i = 0; i < kSegMaxL; ++i) {
      const int t = tb + i;
      if (t < te) {
        const int64_t off = (int64_t)t * N + col;
        r[i] = reward[off];
        v[i] = want_v ? value[off] : 0.f;
        const uint8_t dd = done[off];
        nd[i] = 1.0f - (float)dd;
        if (VALID && dd && myfd == T) myfd = t;
      }
    }
    if (MODE == 0) vnext = (te < T) ? value[(int64_t)te * N + col] : bootstrap[col];
  }
  // Phase 1: affine map of the segment, walking backwards.
  float P = 0.f, Q = 1.f;
  if (active) {
#pragma unroll
    for (int i = kSegMaxL - 1; i >= 0; --i) {
      const int t = tb + i;
      if (t < te) {
        if (MODE == 0) {
          const float vn = (t + 1 < te) ? v[(i + 1) < kSegMaxL ? (i + 1) : i] : vnext;
          const float delta = (r[i] + ((g * vn) * nd[i])) - v[i];
          const float c = gl * nd[i];
          P = delta + c * P;
          Q = c * Q;
        } else {
          const float c = g * nd[i];
          P = r[i] + c * P;
          Q = c * Q;
        }
      }
    }
  }
  sP[w][lane] = P;
  sQ[w][lane] = Q;
  if (VALID) sFd[w][lane] = myfd;
  __syncthreads();
  // Phase 2: carry into my segment = value of the recurrence at the start of segment w+1.
  float carry = (MODE == 0) ? 0.f : (active ? bootstrap[col] : 0.f);
  for (int s = S - 1; s > w; --s) carry = sP[s][lane] + sQ[s][lane] * carry;
  int fd = T;
  if (VALID) {
    for (int s = 0; s < S; ++s) fd = min(fd, sFd[s][lane]);
  }
  if (!active) return;
  // Phase 3: replay with the carry.
  float next = carry;
#pragma unroll
  for (int i = kSegMaxL - 1; i >= 0; --i) {
    const int t = tb + i;
    if (t < te) {
      const int64_t off = (int64_t)t * N + col;
      if (MODE == 0) {
        const float vn = (t + 1 < te) ? v[(i + 1) < kSegMaxL ? (i + 1) : i] : vnext;
        const float delta = (r[i] + ((g * vn) * nd[i])) - v[i];
        const float a = delta + (gl * nd[i]) * next;
        advantage[off] = a;
        return_[off] = a + v[i];
        next = a;
      } else {
        const float ret = r[i] + (g * nd[i]) * next;
        return_[off] = ret;
        if (want_v) advantage[off] = ret - v[i];
        next = ret;
      }
      if (VALID) valid[off] = (t <= fd) ? 1.0f : 0.0f;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void valid_kernel(const uint8_t* __restrict__ done,
                                                    float* __restrict__ valid, int T,
                                                    int64_t N) {
  const int64_t col = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (col >= N) return;
  bool seen[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) seen[k] = false;
  constexpr int U = 8;
  for (int t0 = 0; t0 < T; t0 += U) {
    uint8_t d[U][VEC];
#pragma unroll
    for (int i = 0; i < U; ++i)
      if (t0 + i < T) load_b<VEC>(done + (int64_t)(t0 + i) * N + col, d[i]);
#pragma unroll
    for (int i = 0; i < U; ++i) {
      if (t0 + i < T) {
        float m[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          m[k] = seen[k] ? 0.0f : 1.0f;
          seen[k] = seen[k] || (d[i][k] != 0);
        }
        store_f<VEC>(valid + (int64_t)(t0 + i) * N + col, m);
      }
    }
  }
}

// n-step returns: every output element is independent (utils.py:84-99).
struct NStepCoef {
  float c[32];  // c[n] = (float)(discount ** n), folded in double on the host
};

template <int VEC>
__global__ __launch_bounds__(256) void nstep_kernel(const float* __restrict__ reward,
                                                    const uint8_t* __restrict__ done,
                                                    float* __restrict__ return_,
                                                    uint8_t* __restrict__ done_n, int T_in,
                                                    int T_out, int64_t N, int n_step,
                                                    NStepCoef coef) {
  const int64_t nvec = N / VEC;
  const int64_t total = (int64_t)T_out * nvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / nvec);
    const int64_t col = (i - (int64_t)t * nvec) * VEC;
    float ret[VEC];
    uint8_t dn[VEC];
    load_f<VEC>(reward + (int64_t)t * N + col, ret);
    load_b<VEC>(done + (int64_t)t * N + col, dn);
#pragma unroll
    for (int k = 0; k < VEC; ++k) dn[k] = dn[k] ? 1 : 0;
    for (int n = 1; n < n_step; ++n) {
      if (t + n >= T_in) break;  // do_truncated tail: rows without the n-th reward stop here
      float rn[VEC];
      uint8_t dd[VEC];
      load_f<VEC>(reward + (int64_t)(t + n) * N + col, rn);
      load_b<VEC>(done + (int64_t)(t + n) * N + col, dd);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        // return_ += (discount ** n) * reward[t+n] * (1 - done_n)
        ret[k] = ret[k] + ((coef.c[n] * rn[k]) * (1.0f - (float)dn[k]));
        dn[k] = (dn[k] | (dd[k] ? 1 : 0));
      }
    }
    store_f<VEC>(return_ + (int64_t)t * N + col, ret);
    store_b<VEC>(done_n + (int64_t)t * N + col, dn);
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }

template <int MODE, bool VALID>
int launch_scan(const float* reward, const float* value, const uint8_t* done,
                const float* bootstrap, float* advantage, float* return_, float* valid, int T,
                int64_t N, float g, float gl, int variant, hipStream_t s) {
  if (variant == RLPYT_SCAN_SEGMENTED && T <= kSegMaxL * kSegMaxS && T >= 2) {
    int S = kSegMaxS;
    int L = (int)ceil_div(T, S);
    S = (int)ceil_div(T, L);
    const int64_t grid = ceil_div(N, kWave);
    RL_LAUNCH((scan_segmented_kernel<MODE, VALID>), dim3((unsigned)grid),
                       dim3(S * kWave), 0, s, reward, value, done, bootstrap, advantage,
                       return_, valid, T, N, L, g, gl);
    RL_LAUNCH_CHECK();
    return RLPYT_OK;
  }
  // 4 columns per lane (16-byte loads) once that still fills the chip; 1 otherwise.
  const bool can4 = (N % 4 == 0) && aligned16(reward) && aligned16(return_) &&
                    aligned16(bootstrap) && aligned4(done) &&
                    (value == nullptr || aligned16(value)) &&
                    (advantage == nullptr || aligned16(advantage)) &&
                    (valid == nullptr || aligned16(valid));
  if (can4 && N >= (int64_t)4 * 256 * 1024) {
    const int64_t grid = ceil_div(N / 4, 256);
    RL_LAUNCH((scan_exact_kernel<MODE, 4, 8, VALID>), dim3((unsigned)grid), dim3(256),
                       0, s, reward, value, done, bootstrap, advantage, return_, valid, T, N, g,
                       gl);
  } else if (N >= 64 * 1024) {
    const int64_t grid = ceil_div(N, 256);
    RL_LAUNCH((scan_exact_kernel<MODE, 1, 8, VALID>), dim3((unsigned)grid), dim3(256),
                       0, s, reward, value, done, bootstrap, advantage, return_, valid, T, N, g,
                       gl);
  } else {
    // Few columns: one wave per workgroup to spread over CUs, deep chunks so a whole
    // [T<=32k] column segment of loads is in flight per wait.
    const int64_t grid = ceil_div(N, 64);
    RL_LAUNCH((scan_exact_kernel<MODE, 1, 32, VALID>), dim3((unsigned)grid), dim3(64),
                       0, s, reward, value, done, bootstrap, advantage, return_, valid, T, N, g,
                       gl);
  }
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

extern "C" int rlpyt_gae_f32(const float* reward, const float* value, const uint8_t* done,
                             const float* bootstrap, float* advantage, float* return_,
                             float* valid, int T, int64_t N, double discount,
                             double gae_lambda, int variant, rlpyt_stream_t stream) {
  RL_CHECK_ARG(T >= 0 && N >= 0, RLPYT_EINVAL, "rlpyt_gae_f32: negative size T=%d N=%ld", T,
               (long)N);
  if (T == 0 || N == 0) return RLPYT_OK;  // empty batch: nothing to do (pointers may be null)
  RL_CHECK_ARG(reward && value && done && bootstrap && advantage && return_, RLPYT_EINVAL,
               "rlpyt_gae_f32: null pointer");
  RL_CHECK_ARG(variant == RLPYT_SCAN_EXACT || variant == RLPYT_SCAN_SEGMENTED, RLPYT_EINVAL,
               "rlpyt_gae_f32: unknown variant %d", variant);
  if (T == 0 || N == 0) return RLPYT_OK;
  const float g = (float)discount;
  const float gl = (float)(discount * gae_lambda);  // folded in double first (App. B.1)
  hipStream_t s = (hipStream_t)stream;
  if (valid)
    return launch_scan<0, true>(reward, value, done, bootstrap, advantage, return_, valid, T, N,
                                g, gl, variant, s);
  return launch_scan<0, false>(reward, value, done, bootstrap, advantage, return_, nullptr, T,
                               N, g, gl, variant, s);
}

extern "C" int rlpyt_discount_return_f32(const float* reward, const uint8_t* done,
                                         const float* bootstrap, float* return_,
                                         const float* value, float* advantage, float* valid,
                                         int T, int64_t N, double discount, int variant,
                                         rlpyt_stream_t stream) {
  RL_CHECK_ARG(T >= 0 && N >= 0, RLPYT_EINVAL, "rlpyt_discount_return_f32: negative size");
  if (T == 0 || N == 0) return RLPYT_OK;
  RL_CHECK_ARG(reward && done && bootstrap && return_, RLPYT_EINVAL,
               "rlpyt_discount_return_f32: null pointer");
  RL_CHECK_ARG((value == nullptr) == (advantage == nullptr), RLPYT_EINVAL,
               "rlpyt_discount_return_f32: value and advantage must be given together");
  RL_CHECK_ARG(variant == RLPYT_SCAN_EXACT || variant == RLPYT_SCAN_SEGMENTED, RLPYT_EINVAL,
               "rlpyt_discount_return_f32: unknown variant %d", variant);
  if (T == 0 || N == 0) return RLPYT_OK;
  const float g = (float)discount;
  hipStream_t s = (hipStream_t)stream;
  if (valid)
    return launch_scan<1, true>(reward, value, done, bootstrap, advantage, return_, valid, T, N,
                                g, 0.f, variant, s);
  return launch_scan<1, false>(reward, value, done, bootstrap, advantage, return_, nullptr, T,
                               N, g, 0.f, variant, s);
}

extern "C" int rlpyt_valid_from_done(const uint8_t* done, float* valid, int T, int64_t N,
                                     rlpyt_stream_t stream) {
  RL_CHECK_ARG(T >= 0 && N >= 0, RLPYT_EINVAL, "rlpyt_valid_from_done: negative size");
  if (T == 0 || N == 0) return RLPYT_OK;
  RL_CHECK_ARG(done && valid, RLPYT_EINVAL, "rlpyt_valid_from_done: null pointer");
  if (T == 0 || N == 0) return RLPYT_OK;
  hipStream_t s = (hipStream_t)stream;
  if (N % 4 == 0 && aligned4(done) && aligned16(valid) && N >= 4 * 256 * 256) {
    RL_LAUNCH((valid_kernel<4>), dim3((unsigned)ceil_div(N / 4, 256)), dim3(256), 0, s,
                       done, valid, T, N);
  } else {
    const int bs = N >= 64 * 256 ? 256 : 64;
    RL_LAUNCH((valid_kernel<1>), dim3((unsigned)ceil_div(N, bs)), dim3(bs), 0, s, done,
                       valid, T, N);
  }
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_nstep_return_f32(const float* reward, const uint8_t* done, float* return_,
                                      uint8_t* done_n, int T_in, int64_t N, int n_step,
                                      double discount, int do_truncated,
                                      rlpyt_stream_t stream) {
  RL_CHECK_ARG(n_step >= 1 && n_step <= 32, RLPYT_EINVAL,
               "rlpyt_nstep_return_f32: n_step=%d outside [1,32]", n_step);
  RL_CHECK_ARG(T_in >= 0 && N >= 0, RLPYT_EINVAL, "rlpyt_nstep_return_f32: negative size");
  const int T_out = do_truncated ? T_in : T_in - (n_step - 1);
  if (T_out <= 0 || N == 0) return RLPYT_OK;
  RL_CHECK_ARG(reward && done && return_ && done_n, RLPYT_EINVAL,
               "rlpyt_nstep_return_f32: null pointer");
  NStepCoef coef;
  // Python's float ** int for small ints is repeated multiplication in double (pow());
  // use pow() to match libm exactly as CPython's float_pow does.
  for (int n = 0; n < 32; ++n) coef.c[n] = (float)pow(discount, (double)n);
  hipStream_t s = (hipStream_t)stream;
  const bool can4 = (N % 4 == 0) && aligned16(reward) && aligned16(return_) && aligned4(done) &&
                    aligned4(done_n);
  const int64_t total = (int64_t)T_out * (can4 ? N / 4 : N);
  const int64_t grid = std::min<int64_t>(ceil_div(total, 256), 256 * 8);
  if (can4)
    RL_LAUNCH((nstep_kernel<4>), dim3((unsigned)grid), dim3(256), 0, s, reward, done,
                       return_, done_n, T_in, T_out, N, n_step, coef);
  else
    RL_LAUNCH((nstep_kernel<1>), dim3((unsigned)grid), dim3(256), 0, s, reward, done,
                       return_, done_n, T_in, T_out, N, n_step, coef);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
