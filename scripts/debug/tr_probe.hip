// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): which 16-bit element does lane l receive
// as element j, when every lane supplies the address of its own 4-element (8-byte) chunk?
// lds[i] = i and lane l points at chunk l (elements 4 l .. 4 l + 3), so the printed value v tells the
// source: chunk (= source lane) v / 4, element v % 4.   build: hipcc --offload-arch=gfx950 tr_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
  int h_addr[64];
  uint16_t h_out[256];
  int* d_addr;
  uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr));
  hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 2; ++pat) {
    for (int l = 0; l < 64; ++l) h_addr[l] = pat == 0 ? 4 * l : 4 * ((l * 7) % 64) + 256;  // scattered chunks
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d: lane: (source lane, element) x 4\n", pat);
    for (int l = 0; l < 64; ++l) {
      printf("%2d:", l);
      for (int j = 0; j < 4; ++j) {
        int v = h_out[l * 4 + j], src = -1;
        for (int s = 0; s < 64; ++s)
          if (v >= h_addr[s] && v < h_addr[s] + 4) src = s;
        printf(" (%2d,%d)", src, src >= 0 ? v - h_addr[src] : -1);
      }
      printf("%s", (l & 3) == 3 ? "\n" : "  ");
    }
  }
  return 0;
}
