// What shader clock does the chip hold while n CUs run dense bf16 MFMA (v_mfma_f32_32x32x16_bf16,
// 2 waves per SIMD, back to back -- the matrix-pipe load of the bf16x6 GEMMs)?  Every workgroup reads
// clock64() (shader cycles) and wall_clock64() (constant 100 MHz) around its loop; MHz = cycles / time.
// Also prints the MFMA issue rate per SIMD.   hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o /tmp/cp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VALU>
__global__ __launch_bounds__(512) void burn(long long* out, int iters, float* sink) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < VALU; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k & 7]) : "v"(v[(k + 1) & 7]));
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int VALU>
void run(int nwg, int iters, long long* d, float* sink) {
  std::vector<long long> h(2 * nwg);
  hipLaunchKernelGGL(burn<VALU>, dim3(nwg), dim3(512), 0, 0, d, 64, sink);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(burn<VALU>, dim3(nwg), dim3(512), 0, 0, d, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), d, sizeof(long long) * 2 * nwg, hipMemcpyDeviceToHost);
  std::vector<double> mhz(nwg), cyc(nwg);
  for (int i = 0; i < nwg; ++i) { mhz[i] = (double)h[2 * i] / ((double)h[2 * i + 1] / 100.0); cyc[i] = (double)h[2 * i]; }
  std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
  // per SIMD: 2 waves x 4 MFMAs per iteration
  printf("valu/mfma %d, %4d workgroups: %8.1f us  shader clock min %.0f median %.0f max %.0f MHz  cycles per MFMA per SIMD %.1f\n",
         VALU, nwg, ms * 1e3, mhz.front(), mhz[nwg / 2], mhz.back(), cyc[nwg / 2] / (8.0 * iters));
}

int main() {
  long long* d; float* sink;
  hipMalloc(&d, sizeof(long long) * 2 * 2048); hipMalloc(&sink, 64);
  const int iters = 20000;      // ~ 160k MFMAs per SIMD ~ 2 ms
  for (int nwg : {1, 8, 32, 64, 128, 192, 256, 512}) run<0>(nwg, iters, d, sink);
  for (int nwg : {8, 128, 256}) run<4>(nwg, iters, d, sink);
  return 0;
}
