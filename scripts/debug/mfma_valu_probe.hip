// Does f32 MFMA (v_mfma_f32_16x16x4_f32) share issue / execution bandwidth with plain f32 VALU?
// K independent v_fma_f32 per MFMA, 4 accumulators, 1..4 waves per SIMD; prints cycles per MFMA.
// hipcc --offload-arch=gfx950 -O2 mfma_valu_probe.hip -o /tmp/mvp && /tmp/mvp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = x + i;
  for (int it = 0; it < iters; ++it) {
#define ONE(acc)                                                                    \
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y)); \
  _Pragma("unroll") for (int k = 0; k < K; ++k)                                     \
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k & 7]) : "v"(y));
    ONE(a0) ONE(a1) ONE(a2) ONE(a3)
#undef ONE
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + s;
}

template <int K>
void run(float* d, int waves_per_simd) {
  const int iters = 4096, threads = 256 * waves_per_simd;   // one workgroup per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<K>, dim3(256), dim3(threads), 0, 0, d, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<K>, dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = 4.0 * iters * waves_per_simd;
  printf("K=%d valu/mfma, %d waves/SIMD: %.1f us, %.2f ns per MFMA per SIMD (32 clk @2.4GHz = 13.3 ns)\n",
         K, waves_per_simd, ms * 1e3, ms * 1e6 / mfma_per_simd);
}

int main() {
  float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
  for (int w : {1, 2, 4}) {
    run<0>(d, w); run<1>(d, w); run<2>(d, w); run<4>(d, w); run<6>(d, w); run<8>(d, w);
  }
  return 0;
}
