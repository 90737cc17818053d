// Standalone phase-timing harness for sample_convs_kernel (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRLPYT_TIMING -Irlpyt_amd/csrc scripts/debug/sample_convs_main.hip -o scripts/debug/sample_convs_main
#include <stdarg.h>
#include <vector>
#include "conv.hip"
namespace rlpyt {
void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fprintf(stderr, "\n"); }
static VariantSlot g_slot;
VariantSlot* variant_slot(const void*, const char*) { return &g_slot; }
void variant_hit(VariantSlot*) {}
}
int main() {
  const int64_t B = 256, Bg = 64, T = 8;
  uint8_t *obs, *nf, *full; int64_t* t_dev; int32_t* slot; float *w1, *b1, *w2, *b2, *y2;
  hipMalloc(&obs, (size_t)T * B * 33280); hipMemset(obs, 7, (size_t)T * B * 33280);
  hipMalloc(&nf, Bg * 8320); hipMemset(nf, 9, Bg * 8320);
  hipMalloc(&full, Bg * 33280); hipMemset(full, 3, Bg * 33280);
  hipMalloc(&t_dev, 8); int64_t t = 3; hipMemcpy(t_dev, &t, 8, hipMemcpyHostToDevice);
  hipMalloc(&slot, Bg * 4); hipMemset(slot, 0xff, Bg * 4);      // -1: shift + newest frame
  hipMalloc(&w1, 4096 * 4); hipMalloc(&b1, 64); hipMalloc(&w2, 8192 * 4); hipMalloc(&b2, 128);
  hipMemset(w1, 0, 4096 * 4); hipMemset(b1, 0, 64); hipMemset(w2, 0, 8192 * 4); hipMemset(b2, 0, 128);
  hipMalloc(&y2, Bg * 3456 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < 20; ++r)
      rlpyt_atari_sample_convs_f32(obs, t_dev, B, 64, Bg, nf, full, slot, nullptr, nullptr, nullptr, nullptr, w1, b1,
                                   w2, b2, 1.f / 255, y2, nullptr);
    hipEventRecord(e1, nullptr);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("sync=%s %.2f us per launch (20 back to back)\n", hipGetErrorString(e), ms * 1e3 / 20);
  }
  std::vector<float> tt(512 * 16 * 8);
  rlpyt_debug_timing_read(tt.data(), (int)tt.size());
  const char* nm[8] = {"issue loads", "wait loads", "stage", "sync1", "conv1", "sync2", "conv2 chain", "reduce+store"};
  for (int k = 0; k < 8; ++k) {
    double s = 0, mx = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 16; ++w) { double v = tt[((size_t)b * 16 + w) * 8 + k]; s += v; mx = v > mx ? v : mx; }
    printf("%-14s mean %7.0f max %7.0f cycles\n", nm[k], s / 256 / 16, mx);
  }
  return 0;
}
