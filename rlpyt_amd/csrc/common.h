// Shared helpers for the gfx950 kernels behind include/rlpyt_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/rlpyt_hip.h"

#define RLPYT_ABI_VERSION RLPYT_HIP_ABI_VERSION

namespace rlpyt {

void set_error(const char* fmt, ...);

#define RL_CHECK_ARG(cond, code, ...)         \
  do {                                        \
    if (!(cond)) {                            \
      ::rlpyt::set_error(__VA_ARGS__);        \
      return (code);                          \
    }                                         \
  } while (0)

#define RL_HIP(expr)                                                               \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      ::rlpyt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                         __FILE__, __LINE__);                                      \
      return RLPYT_EHIP;                                                           \
    }                                                                              \
  } while (0)

#define RL_LAUNCH_CHECK()                                                          \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      ::rlpyt::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                      \
      return RLPYT_EHIP;                                                           \
    }                                                                              \
  } while (0)

// ---- kernel-variant registry (rlpyt_hip_variant_*) -----------------------------------------
// Every launch site goes through RL_LAUNCH, which counts the launch per kernel INSTANTIATION
// (host stub address; resolved to the device kernel's name on request), so that a parity test
// can assert which size / alignment-gated variant actually ran (tests/test_variants.py).
struct VariantSlot {
  const void* host_fn;
  const char* site;          // stringified kernel expression at the launch site
  unsigned long long count;  // launches (graph captures count once, at capture)
};
VariantSlot* variant_slot(const void* host_fn, const char* site);
void variant_hit(VariantSlot* slot);

#define RL_LAUNCH(kernel, grid, block, lds, stream, ...)                                    \
  do {                                                                                      \
    static ::rlpyt::VariantSlot* _rl_slot =                                                 \
        ::rlpyt::variant_slot(reinterpret_cast<const void*>(kernel), #kernel);             \
    ::rlpyt::variant_hit(_rl_slot);                                                         \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                      \
  } while (0)

#define RL_ALIGNED16(p) ((reinterpret_cast<uintptr_t>(p) & 15) == 0)

__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;  // CDNA wavefront width

// Wave-level sum over the 64 lanes (butterfly; every lane gets the total).
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    T o = __shfl_xor(v, off, kWave);
    v = o > v ? o : v;
  }
  return v;
}

// Block-level sum of K values per thread through LDS; result valid in thread 0.
// blockDim.x must be a multiple of 64 and <= 1024. `scratch` holds K*16 T's.
template <int K, typename T>
__device__ __forceinline__ void block_sum(T (&v)[K], T* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) scratch[wid * K + k] = v[k];
  }
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      T x = lane < nw ? scratch[lane * K + k] : T(0);
      v[k] = wave_sum(x);
    }
  }
  __syncthreads();
}

}  // namespace rlpyt
