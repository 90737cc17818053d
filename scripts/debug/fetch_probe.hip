// FETCH_SIZE / WRITE_SIZE calibration probe (VERDICT r5 item 2, MI355X_MICROARCH.md "HBM": "other
// access widths ... are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel here touches each byte of its buffer EXACTLY ONCE in one of the access patterns the
// path's kernels use; the program prints the byte counts, scripts/fetch_calibration.sh runs it under
// rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one counter per pass) and divides.
//
//   read_stream16  : lane i of a wave reads 16 B at i            (1024 B contiguous per instruction:
//                    the guide's calibrated pattern, FETCH_SIZE KB = 1/2 bytes expected)
//   read_stream4   : lane i reads 4 B at i                       (256 B contiguous per instruction)
//   read_run64     : rollout_fc_kernel's weight read (csrc/step.hip): lane = (j = lane & 15, kq =
//                    lane >> 4) reads 16 B at row j, byte 16 kq + 64 g -- 16 rows x 64-byte runs per
//                    instruction, the 8 instructions of a thread walk 512 B along its row; rows
//                    13,824 B apart (K = 3456 floats)
//   read_run128    : 8 rows x 128-byte runs per instruction (lane = (j = lane & 7, kq = lane >> 3))
//   write_stream16 / write_stream4 / write_run64 : the same maps for stores (write_run64 = the
//                    split-K partial store of rollout_fc_kernel)
// Sizes: "w" = the trunk weight's own shape [512][3456] f32 (7.08 MB, L2 / Infinity-Cache resident
// between launches, as in the rollout) and "big" = [16384][3456] (226 MB) so that one launch cannot
// be served from the 4 MB L2s.  build: hipcc --offload-arch=gfx950 -O3 fetch_probe.hip -o fetch_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int K = 3456;    // row length (floats) of the run patterns

__device__ __forceinline__ void sink(float v, float* out) {
  if (v == 1.2345e-33f) out[0] = v;      // never true for the probe's data; keeps the loads alive
}

// ---- contiguous streams: grid-stride over 16 B / 4 B elements
template <int BIG>
__global__ __launch_bounds__(256) void read_stream16(const f4* __restrict__ p, long n16, float* out) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
    const f4 v = p[i];
    s += v[0] + v[1] + v[2] + v[3];
  }
  sink(s, out);
}
template <int BIG>
__global__ __launch_bounds__(256) void read_stream4(const float* __restrict__ p, long n4, float* out) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) s += p[i];
  sink(s, out);
}
template <int BIG>
__global__ __launch_bounds__(256) void write_stream16(f4* __restrict__ p, long n16) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256)
    p[i] = f4{1.f, 2.f, 3.f, 4.f};
}
template <int BIG>
__global__ __launch_bounds__(256) void write_stream4(float* __restrict__ p, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i] = 1.f;
}

// ---- run patterns over a [rows][K] f32 matrix: workgroup (bx, by) owns rows 64 bx .. + 63 (a wave
// 16 of them) and the 128-float K slice by (K = 27 x 128), exactly rollout_fc_kernel's W tiling
template <int RUN_LANES, int BIG>   // lanes per row run: 4 -> 64 B runs x 16 rows, 8 -> 128 B runs x 8 rows
__global__ __launch_bounds__(256) void read_run(const float* __restrict__ w, float* out) {
  constexpr int ROWS = 64 / RUN_LANES;              // rows per instruction
  constexpr int STEP = 4 * RUN_LANES;               // floats a run covers
  constexpr int NG = 128 / STEP;                    // instructions to walk the 128-float slice
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % ROWS, kq = lane / ROWS;
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16 / ROWS; ++r) {             // a wave still covers 16 rows in total
    const float* row = w + (long)(blockIdx.x * 64 + wave * 16 + r * ROWS + j) * K + blockIdx.y * 128 + 4 * kq;
    f4 v[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) v[g] = *reinterpret_cast<const f4*>(row + STEP * g);
#pragma unroll
    for (int g = 0; g < NG; ++g) s += v[g][0] + v[g][1] + v[g][2] + v[g][3];
  }
  sink(s, out);
}
// rollout_fc_kernel's partial store: D[row = 4 kq + r][col = j] -> 16 B at (row m = j, column 4 kq):
// 16 rows x 64-byte runs per instruction, rows N = 512 floats apart; here over a [rows][512] matrix,
// workgroup (bx, by): rows 64 by + 16 t + j, columns 64 bx + 16 wave + 4 kq
template <int BIG>
__global__ __launch_bounds__(256) void write_run64(float* __restrict__ p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
    *reinterpret_cast<f4*>(p + (long)(blockIdx.y * 64 + 16 * t + j) * 512 + blockIdx.x * 64 + wave * 16 + 4 * kq) =
        f4{1.f, 2.f, 3.f, 4.f};
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  const long rows_w = 512, rows_big = 16384;
  const long bytes_w = rows_w * K * 4, bytes_big = rows_big * K * 4;
  float *buf, *out;
  CK(hipMalloc(&buf, bytes_big));
  CK(hipMalloc(&out, 256));
  CK(hipMemset(buf, 0, bytes_big));
  CK(hipDeviceSynchronize());
  printf("{\"K\": %d, \"bytes_w\": %ld, \"bytes_big\": %ld, \"reps\": %d,\n \"launch_order\": [", K, bytes_w,
         bytes_big, reps);
  const char* sep = "";
  auto note = [&](const char* kernel, const char* size, long bytes, const char* kind) {
    printf("%s\n  {\"kernel\": \"%s\", \"size\": \"%s\", \"bytes\": %ld, \"kind\": \"%s\"}", sep, kernel, size,
           bytes, kind);
    sep = ",";
  };
  for (int rep = 0; rep < reps; ++rep) {
    for (int big = 0; big < 2; ++big) {
      const long rows = big ? rows_big : rows_w, bytes = big ? bytes_big : bytes_w;
      const char* sz = big ? "big" : "w";
      const int grid = (int)((bytes / 16 + 255) / 256 < 4096 ? (bytes / 16 + 255) / 256 : 4096);
#define LAUNCH_ALL(BIG_)                                                                            \
  read_stream16<BIG_><<<grid, 256>>>((const f4*)buf, bytes / 16, out);                              \
  read_stream4<BIG_><<<grid, 256>>>(buf, bytes / 4, out);                                           \
  read_run<4, BIG_><<<dim3((unsigned)(rows / 64), K / 128), 256>>>(buf, out);                       \
  read_run<8, BIG_><<<dim3((unsigned)(rows / 64), K / 128), 256>>>(buf, out);                       \
  write_stream16<BIG_><<<grid, 256>>>((f4*)buf, bytes / 16);                                        \
  write_stream4<BIG_><<<grid, 256>>>(buf, bytes / 4);                                               \
  write_run64<BIG_><<<dim3(8, (unsigned)(rows_p / 64)), 256>>>(buf);
      // write_run64: a [rows_p][512] f32 matrix with rows_p = bytes / 2048 (a multiple of 64 for both sizes)
      const long rows_p = bytes / 2048;
      if (big) { LAUNCH_ALL(1) } else { LAUNCH_ALL(0) }
      if (!rep) {
        const char* tag = big ? "1" : "0";
        char name[64];
        const char* kinds[7][2] = {{"read_stream16<%s>", "read"}, {"read_stream4<%s>", "read"},
                                   {"read_run<4, %s>", "read"}, {"read_run<8, %s>", "read"},
                                   {"write_stream16<%s>", "write"}, {"write_stream4<%s>", "write"},
                                   {"write_run64<%s>", "write"}};
        for (auto& kk : kinds) {
          snprintf(name, sizeof name, kk[0], tag);
          note(name, sz, bytes, kk[1]);
        }
      }
      CK(hipDeviceSynchronize());
    }
  }
  printf("\n ]}\n");
  CK(hipGetLastError());
  CK(hipFree(buf));
  CK(hipFree(out));
  return 0;
}
