// Standalone phase-timing harness for convs_fwd_fused_kernel (no torch): per-wave cycle totals of the
// two windows and the barrier waits of both roles.  Build here, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRLPYT_TIMING -Irlpyt_amd/csrc scripts/debug/convs_fused_main.hip -o scripts/debug/convs_fused_main
#include <stdarg.h>
#include <vector>
#include "conv.hip"
namespace rlpyt {
void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fprintf(stderr, "\n"); }
static VariantSlot g_slot;
VariantSlot* variant_slot(const void*, const char*) { return &g_slot; }
void variant_hit(VariantSlot*) {}
}
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 8192;
  const int T = 128; const int64_t B = 256;
  uint8_t* obs; float *w1, *b1, *w2, *b2, *y1, *y2; uint32_t* mask;
  const size_t nobs = (size_t)T * B * 33280;
  hipMalloc(&obs, nobs);
  hipMalloc(&w1, 4096 * 4); hipMalloc(&b1, 64); hipMalloc(&w2, 8192 * 4); hipMalloc(&b2, 128);
  {  // random bytes / weights: zero-filled inputs clock the chip ~19 % higher than real data
    std::vector<uint32_t> h(nobs / 4);
    uint32_t sd = 12345u;
    for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = sd ^ (sd >> 13); }
    hipMemcpy(obs, h.data(), nobs, hipMemcpyHostToDevice);
    std::vector<float> w(8192);
    auto fill = [&](float* d, int n, float sc) {
      for (int i = 0; i < n; ++i) { sd = sd * 1664525u + 1013904223u; w[i] = (((sd >> 8) & 0xffff) / 65536.f - 0.5f) * sc; }
      hipMemcpy(d, w.data(), n * 4, hipMemcpyHostToDevice);
    };
    fill(w1, 4096, 0.12f); fill(b1, 16, 0.2f); fill(w2, 8192, 0.12f); fill(b2, 32, 0.2f);
  }
  hipMalloc(&y1, (size_t)M * 7600 * 4); hipMalloc(&y2, (size_t)M * 3456 * 4); hipMalloc(&mask, (size_t)M * 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, nullptr);
    int rc = rlpyt_atari_convs_fwd_f32(obs, nullptr, T, B, M, w1, b1, w2, b2, 1.f / 255, y1, y2, mask, nullptr);
    hipEventRecord(e1, nullptr);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("rc=%d sync=%s %.1f us\n", rc, hipGetErrorString(e), ms * 1e3);
  }
  std::vector<float> t(512 * 16 * 8);
  rlpyt_debug_timing_read(t.data(), (int)t.size());
  const char* names[2][4] = {{"C1 stage", "C1 wait1", "C1 prefetch+conv1", "C1 wait2"},
                             {"C2 taps", "C2 wait1", "C2 finish", "C2 wait2"}};
  const int64_t per = (M + 255) / 256;
  for (int role = 0; role < 2; ++role)
    for (int k = 0; k < 4; ++k) {
      double s = 0, mx = 0; int n = 0;
      for (int b = 0; b < 256; ++b)
        for (int w = 4 * role; w < 4 * role + 4; ++w) { const double v = t[((size_t)b * 16 + w) * 8 + k]; s += v; mx = v > mx ? v : mx; ++n; }
      printf("%-20s mean %8.0f cycles / image (max wave %8.0f)\n", names[role][k], s / n / per, mx / per);
    }
  return 0;
}
