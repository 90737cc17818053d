// Where do the 8 waves of a 512-thread workgroup land?  Prints (CU, SIMD) per wave for a few
// workgroups: checks the assumption of conv2_bwd_kernel that waves w and w + 4 share a SIMD.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
  __shared__ float big[26000];   // ~104 KB: one workgroup per CU, like conv2_bwd
  big[threadIdx.x] = 0.f;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
  }
}
int main() {
  unsigned* d;
  const int nb = 512;
  hipMalloc(&d, nb * 8 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, d);
  static unsigned h[nb * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int same = 0, total = 0, hist[4] = {0, 0, 0, 0};
  for (int b = 0; b < nb; ++b) {
    for (int w = 0; w < 4; ++w) {
      const unsigned s0 = (h[b * 8 + w] >> 4) & 3, s1 = (h[b * 8 + w + 4] >> 4) & 3;
      same += s0 == s1;
      ++total;
    }
    if (b < 6) {
      printf("wg %d:", b);
      for (int w = 0; w < 8; ++w)
        printf(" w%d=(cu %u simd %u wave %u)", w, (h[b * 8 + w] >> 8) & 15, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15);
      printf("\n");
    }
    for (int w = 0; w < 8; ++w) hist[(h[b * 8 + w] >> 4) & 3]++;
  }
  printf("waves w and w+4 on the same SIMD: %d of %d; SIMD histogram %d %d %d %d\n", same, total,
         hist[0], hist[1], hist[2], hist[3]);
  return 0;
}
