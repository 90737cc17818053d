// Standalone harness for rlpyt_lstm_seq_f32 (no torch): build here, run on the GPU box.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Irlpyt_amd/csrc scripts/debug/lstm_seq_main.hip -o scripts/debug/lstm_seq_main
#include <math.h>
#include <stdarg.h>
#include <vector>
#include "lstm_seq.hip"
namespace rlpyt {
void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fprintf(stderr, "\n"); }
static VariantSlot g_slot;
VariantSlot* variant_slot(const void*, const char*) { return &g_slot; }
void variant_hit(VariantSlot*) {}
}
int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 512, T = argc > 2 ? atoi(argv[2]) : 3, B = argc > 3 ? atoi(argv[3]) : 5;
  std::vector<float> xp((size_t)T * B * 4 * H), w((size_t)4 * H * H), h0((size_t)B * H), c0((size_t)B * H);
  unsigned s = 1;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : xp) v = rnd();
  for (auto& v : w) v = rnd() * 0.1f;
  for (auto& v : h0) v = rnd();
  for (auto& v : c0) v = rnd();
  float *dxp, *dw, *dh, *dc, *dout;
  hipMalloc(&dxp, xp.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&dh, h0.size() * 4);
  hipMalloc(&dc, c0.size() * 4); hipMalloc(&dout, (size_t)T * B * H * 4);
  hipMemcpy(dxp, xp.data(), xp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dh, h0.data(), h0.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dc, c0.data(), c0.size() * 4, hipMemcpyHostToDevice);
  printf("launch H=%d T=%d B=%d\n", H, T, B); fflush(stdout);
  int rc = rlpyt_lstm_seq_f32(dxp, dw, dh, dc, dout, T, B, H, nullptr);
  hipError_t e = hipDeviceSynchronize();
  printf("rc=%d sync=%s\n", rc, hipGetErrorString(e)); fflush(stdout);
  std::vector<float> out((size_t)T * B * H), cT((size_t)B * H);
  hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(cT.data(), dc, cT.size() * 4, hipMemcpyDeviceToHost);
  std::vector<double> h(h0.begin(), h0.end()), c(c0.begin(), c0.end()), hn(h.size());
  double err = 0;
  for (int t = 0; t < T; ++t) {
    for (int b = 0; b < B; ++b)
      for (int u = 0; u < H; ++u) {
        double g[4];
        for (int q = 0; q < 4; ++q) {
          double a = xp[((size_t)t * B + b) * 4 * H + q * H + u];
          for (int k = 0; k < H; ++k) a += (double)w[((size_t)q * H + u) * H + k] * h[(size_t)b * H + k];
          g[q] = a;
        }
        auto sg = [](double x) { return 1. / (1. + exp(-x)); };
        const double c1 = sg(g[1]) * c[(size_t)b * H + u] + sg(g[0]) * tanh(g[2]);
        c[(size_t)b * H + u] = c1;
        hn[(size_t)b * H + u] = sg(g[3]) * tanh(c1);
        err = fmax(err, fabs(hn[(size_t)b * H + u] - out[((size_t)t * B + b) * H + u]));
      }
    h = hn;
  }
  double errc = 0;
  for (size_t i = 0; i < cT.size(); ++i) errc = fmax(errc, fabs(c[i] - cT[i]));
  printf("max |h err| = %.3g, max |c err| = %.3g\n", err, errc);
  return 0;
}
