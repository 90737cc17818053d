// Standalone phase-timing harness (no torch) for the persistent conv kernels that carry RL_T marks:
// per-wave cycle totals per image of conv1_wgrad_kernel / conv2_bwd_x6_kernel / conv2_fwd_x6_kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRLPYT_TIMING -Irlpyt_amd/csrc scripts/debug/conv_phase_main.hip -o scripts/debug/conv_phase_main
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "conv.hip"
namespace rlpyt {
void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fprintf(stderr, "\n"); }
static VariantSlot g_slot;
VariantSlot* variant_slot(const void*, const char*) { return &g_slot; }
void variant_hit(VariantSlot*) {}
}
static uint32_t sd = 12345u;
static void fill_bytes(void* d, size_t n) {
  std::vector<uint32_t> h(n / 4);
  for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = sd ^ (sd >> 13); }
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
}
static void fill_f32(float* d, size_t n, float sc) {
  std::vector<float> h(n);
  for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = (((sd >> 8) & 0xffff) / 65536.f - 0.5f) * sc; }
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
}
int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "wgrad1";
  const int64_t M = 8192; const int T = 128; const int64_t B = 256;
  uint8_t* obs; float *w1, *b1, *w2, *b2, *y1, *y2, *dy1, *g2, *ws, *dw, *db; uint32_t* mask;
  hipMalloc(&obs, (size_t)T * B * 33280); fill_bytes(obs, (size_t)T * B * 33280);
  hipMalloc(&w1, 4096 * 4); hipMalloc(&b1, 64); hipMalloc(&w2, 8192 * 4); hipMalloc(&b2, 128);
  fill_f32(w1, 4096, .12f); fill_f32(b1, 16, .2f); fill_f32(w2, 8192, .12f); fill_f32(b2, 32, .2f);
  hipMalloc(&y1, (size_t)M * 7600 * 4); hipMalloc(&dy1, (size_t)M * 7600 * 4);
  hipMalloc(&y2, (size_t)M * 3456 * 4); hipMalloc(&g2, (size_t)M * 3456 * 4); hipMalloc(&mask, (size_t)M * 512);
  fill_f32(y1, (size_t)M * 7600, 1.f); fill_f32(dy1, (size_t)M * 7600, 1.f); fill_f32(g2, (size_t)M * 3456, 1.f);
  fill_bytes(mask, (size_t)M * 512);
  hipMalloc(&ws, rlpyt_atari_conv_wgrad_workspace_bytes()); hipMalloc(&dw, 8192 * 4); hipMalloc(&db, 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, nullptr);
    int rc = 0;
    if (!strcmp(which, "wgrad1")) rc = rlpyt_atari_conv1_wgrad_f32(obs, nullptr, T, B, M, dy1, 1.f / 255, ws, dw, db, nullptr);
    else if (!strcmp(which, "bwd2")) rc = rlpyt_atari_conv2_bwd_x6_f32(g2, mask, y1, M, w2, dy1, ws, dw, db, nullptr);
    else rc = rlpyt_atari_conv2_fwd_f32(y1, M, w2, b2, y2, mask, nullptr);
    hipEventRecord(e1, nullptr);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%s rc=%d sync=%s %.1f us\n", which, rc, hipGetErrorString(e), ms * 1e3);
  }
  std::vector<float> t(512 * 16 * 8);
  rlpyt_debug_timing_read(t.data(), (int)t.size());
  const int64_t per = (M + 255) / 256;
  for (int w = 0; w < 8; ++w) {
    printf("wave %d:", w);
    for (int k = 0; k < 8; ++k) {
      double s = 0;
      for (int b = 0; b < 256; ++b) s += t[((size_t)b * 16 + w) * 8 + k];
      printf(" %7.0f", s / 256 / per);
    }
    printf("\n");
  }
  return 0;
}
