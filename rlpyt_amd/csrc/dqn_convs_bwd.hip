// Backward pass of the DQN-family conv stack at update-batch sizes (round 6):
//   Conv2d(4, 32, 8, stride 4) -> ReLU -> Conv2d(32, 64, 4, stride 2, pad 1) -> ReLU ->
//   Conv2d(64, 64, 3, stride 1, pad 1) -> ReLU -> flatten
// = autograd through `self.conv` of rlpyt/models/dqn/atari_dqn_model.py:30-37 in the online network's
// pass of DQN.loss (rlpyt/algos/dqn/dqn.py:176-180, 226-265), for the activations the forward kernels of
// dqn_convs.hip left behind (y1 [N][475][32], y2 [N][108][64] channels-last, y3 [N][64][108]).
//
// Why own kernels: through the library one update (batch 128) spent ~335 us of device time here -- five
// implicit-GEMM kernels tuned for large batches (18-34 us each) inside ~45 launches of workspace fills,
// buffer copies, layout conversions, ReLU-mask products and bias-gradient reductions of 4-9 us
// (profiles/r6_dqn_region_own_fwd.txt).  Here: weight packing + five kernels + one fixed-order reduction.
//
// Arithmetic: v_mfma_f32_16x16x4_f32, f32 accumulate, the structure of dqn_convs.hip (the batch is
// latency-class: a few thousand MFMAs per image).  Lane l of a wave supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15] and receives D[row = 4 (l >> 4) + r][col = l & 15].
//   data gradients  (dgrad3, dgrad2): a convolution of the masked output gradient with the transposed,
//     flipped weights -- A = those weights (one 16-channel tile per wave, in registers, from a packed copy
//     made on the stream), B = the gradient plane with a zero border in LDS [pixel][channel]; conv2's
//     stride 2 makes four parity classes of input pixels with 2 x 2 taps each;
//   weight gradients (wgrad3, wgrad2, wgrad1): the contraction runs over the POSITIONS of an image
//     (K-slot = position), A = dz[position][16 output channels], B = the layer input under one tap
//     [position][16 input channels]; a workgroup owns one output-channel tile for a group of images,
//     accumulates in registers and writes ONE partial; dqn_bwd_reduce_kernel sums the partials in a
//     fixed order (deterministic) and scales conv1's weight gradient by 1/255.
// ReLU masks come from the kept activations (y > 0), bias gradients are the column sums of dz.
#include <algorithm>

#include "common.h"

namespace rlpyt {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int C0 = 4, H0 = 104, W0 = 80, HW0 = H0 * W0, IMG = C0 * HW0;      // 33,280 B
constexpr int C1 = 32, H1 = 25, W1 = 19, P1 = H1 * W1;                        // 475 positions
constexpr int C2 = 64, H2 = 12, W2 = 9, P2 = H2 * W2;                         // 108
constexpr int C3 = 64;
constexpr int K1 = C0 * 64, K2 = C1 * 16, K3 = C2 * 9;                        // 256, 512, 576
// gradient plane of conv2 / conv3 outputs: 12 x 9 pixels + border, 64 (+ 4 pad) channels
constexpr int GPW = W2 + 2, GPH = H2 + 2, GCS = C2 + 4;                       // 11, 14, 68
constexpr int R3T = K3 / 4;                          // 144 registers: dgrad3 weights of a channel tile
constexpr int R2T = (4 * C2) / 4;                    // 64 registers: dgrad2 weights of (class, channel tile)
constexpr int PKT3 = (C2 / 16) * R3T * 64;           // 36,864 floats
constexpr int PKT2 = 4 * (C1 / 16) * R2T * 64;       // 32,768 floats
constexpr int PACKED_BWD = PKT3 + PKT2;
constexpr int T2 = (P2 + 15) / 16;                   // 7 position tiles of 12 x 9
constexpr int DG3_THREADS = T2 * 64;                 // 448
constexpr int DG2_THREADS = 512;
constexpr int PART3 = C3 * K3 + C3, PART2 = C2 * K2 + C2, PART1 = C1 * K1 + C1;

// ---- transposed / flipped weights in register order ---------------------------------------------
// dgrad3: tile ct = 16 input channels ci; r = (tap' * 4 + hc) * 4 + sp: co = 16 hc + 4 kq + sp,
//         weight w3[co][ci][8 - tap']
// dgrad2: (class = 2 py + px, tile cit); r = ((2 a + b) * 4 + hc) * 4 + sp: co = 16 hc + 4 kq + sp,
//         weight w2[co][ci][py + 2 a][px + 2 b]
__global__ __launch_bounds__(256) void dqn_pack_bwd_weights_kernel(const float* __restrict__ w2,
                                                                   const float* __restrict__ w3,
                                                                   float* __restrict__ packed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= PACKED_BWD) return;
  const int lane = i & 63, j = lane & 15, kq = lane >> 4;
  if (i < PKT3) {
    const int r = (i >> 6) % R3T, ct = (i >> 6) / R3T;
    const int sp = r & 3, hc = (r >> 2) & 3, tap = r >> 4;
    packed[i] = w3[(16 * hc + 4 * kq + sp) * K3 + (ct * 16 + j) * 9 + (8 - tap)];
  } else {
    const int q = i - PKT3, r = (q >> 6) % R2T, u = (q >> 6) / R2T;      // u = class * 2 + cit
    const int cit = u & 1, cls = u >> 1, py = cls >> 1, px = cls & 1;
    const int sp = r & 3, hc = (r >> 2) & 3, ab = r >> 4, a = ab >> 1, b = ab & 1;
    packed[i] = w2[(16 * hc + 4 * kq + sp) * K2 + (cit * 16 + j) * 16 + (py + 2 * a) * 4 + px + 2 * b];
  }
}

// Staging of a 12 x 9 x 64 gradient into the zero-bordered plane [GPH * GPW][GCS], in two halves so that
// the loads of the NEXT image can be in flight behind the MFMAs of the current one:
//   NCHW: dz = g[c][pos] * (y[c][pos] > 0) from the flattened conv3 output / its gradient -- read in
//         memory order (lanes along the positions of a channel: coalesced; the first version read four
//         channels per lane, 432 bytes apart, and the staging was half of dgrad3's 33 us), scattered
//         into the plane (a wave's 64 stores: 4-way bank conflicts at GCS = 68), border zeroed once;
//   else:  dz[pos][c] as written by dgrad3 (already masked), one float4 per lane.
template <int NTHREADS>
struct NchwStage {
  static constexpr int NE = C2 * P2, NIT = (NE + NTHREADS - 1) / NTHREADS;
  float gv[NIT], yv[NIT];
  __device__ __forceinline__ void load(const float* __restrict__ g, const float* __restrict__ y, int tid) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = min(tid + k * NTHREADS, NE - 1);
      gv[k] = g[f];
      yv[k] = y[f];
    }
  }
  __device__ __forceinline__ void store(float* plane, int tid) const {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = tid + k * NTHREADS;
      const int c = f / P2, pos = f - c * P2, py = pos / W2, px = pos - py * W2;
      if (f < NE) plane[((py + 1) * GPW + px + 1) * GCS + c] = yv[k] > 0.f ? gv[k] : 0.f;
    }
  }
};
template <int NTHREADS>
struct NhwcStage {
  static constexpr int Q = C2 / 4, NV = GPH * GPW * Q, NIT = (NV + NTHREADS - 1) / NTHREADS;
  f32x4 v[NIT];
  __device__ __forceinline__ void load(const float* __restrict__ g, int tid) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = min(tid + k * NTHREADS, NV - 1);
      const int pix = f / Q, qd = f - pix * Q, py = pix / GPW, px = pix - py * GPW;
      const bool in = (py >= 1) && (py <= H2) && (px >= 1) && (px <= W2);
      const int pos = in ? (py - 1) * W2 + (px - 1) : 0;
      v[k] = *reinterpret_cast<const f32x4*>(g + pos * C2 + 4 * qd);
      if (!in) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __device__ __forceinline__ void store(float* plane, int tid) const {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = tid + k * NTHREADS;
      const int pix = f / Q, qd = f - pix * Q;
      if (f < NV) *reinterpret_cast<f32x4*>(plane + pix * GCS + 4 * qd) = v[k];
    }
  }
};

// Both data-gradient kernels are persistent over the images beyond kPersistImages (grid = min(N, 256) x 4:
// workgroup (slot, tile | class) walks the images slot, slot + slots, ... with ITS transposed weights in
// registers, the next image's gradient requested before the MFMAs of the current one), as the forward
// kernels of dqn_convs.hip.
constexpr int kPersistImages = 256;

// ---- dgrad3: dz2[n][pos][ci] = (y2 > 0) * sum_{tap', co} dz3pad[pos + tap' - 1][co] w3[co][ci][8 - tap'] ----
__global__ __launch_bounds__(DG3_THREADS) void dqn_dgrad3_kernel(const float* __restrict__ g3,
                                                                 const float* __restrict__ y3,
                                                                 const float* __restrict__ y2,
                                                                 const float* __restrict__ packed,
                                                                 float* __restrict__ dz2, int64_t N) {
  __shared__ __attribute__((aligned(16))) float plane[GPH * GPW * GCS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int64_t slots = gridDim.x >> 2;
  int64_t n = blockIdx.x >> 2;
  const int ct = (int)(blockIdx.x & 3);
  NchwStage<DG3_THREADS> st;
  st.load(g3 + n * (C3 * P2), y3 + n * (C3 * P2), tid);
  float wa[R3T];
  const float* __restrict__ wp = packed + (int64_t)ct * R3T * 64 + lane;
#pragma unroll
  for (int r = 0; r < R3T; ++r) wa[r] = wp[r * 64];
  for (int i = tid; i < GPH * GPW * GCS / 4; i += DG3_THREADS)      // the border stays zero
    reinterpret_cast<f32x4*>(plane)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int lpos = wave * 16 + j, q = min(lpos, P2 - 1);
  const int oy = q / W2, ox = q - oy * W2;
  const float* base = plane + (oy * GPW + ox) * GCS + 4 * kq;
  for (;;) {
    st.store(plane, tid);
    __syncthreads();
    const int64_t nn = n + slots;
    if (nn < N) st.load(g3 + nn * (C3 * P2), y3 + nn * (C3 * P2), tid);      // uniform
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int hc = 0; hc < 4; hc += 2) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(base + (ky * GPW + kx) * GCS + 16 * hc);
        const f32x4 bw = *reinterpret_cast<const f32x4*>(base + (ky * GPW + kx) * GCS + 16 * hc + 16);
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
          acc = mfma16(wa[(tap * 4 + hc) * 4 + sp], bv[sp], acc);
          acc_b = mfma16(wa[(tap * 4 + hc + 1) * 4 + sp], bw[sp], acc_b);
        }
      }
    }
    acc += acc_b;
    if (lpos < P2) {
      const int64_t o = (n * P2 + q) * C2 + ct * 16 + 4 * kq;
      const f32x4 m = *reinterpret_cast<const f32x4*>(y2 + o);
      f32x4 out;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[r] = m[r] > 0.f ? acc[r] : 0.f;
      *reinterpret_cast<f32x4*>(dz2 + o) = out;
    }
    if (nn >= N) break;
    n = nn;
    __syncthreads();
  }
}

// ---- dgrad2: the stride-2 transposed convolution, one parity class of input pixels per workgroup ----
// class (py, px): pixels iy = 2 u + 1 - py, ix = 2 v + 1 - px; taps ky = py + 2 a, kx = px + 2 b read
// dz2 at (oy, ox) = (u + 1 - py - a, v + 1 - px - b) -- out of range only by one, i.e. the zero border.
__global__ __launch_bounds__(DG2_THREADS) void dqn_dgrad2_kernel(const float* __restrict__ dz2,
                                                                 const float* __restrict__ y1,
                                                                 const float* __restrict__ packed,
                                                                 float* __restrict__ dz1, int64_t N) {
  __shared__ __attribute__((aligned(16))) float plane[GPH * GPW * GCS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int64_t slots = gridDim.x >> 2;
  int64_t n = blockIdx.x >> 2;
  const int cls = (int)(blockIdx.x & 3), py = cls >> 1, px = cls & 1;
  const int cit = wave & 1, slot = wave >> 1;
  const int nu = H1 / 2 + py, nv = W1 / 2 + px, npix = nu * nv;        // 12 | 13 rows, 9 | 10 columns
  NhwcStage<DG2_THREADS> st;
  st.load(dz2 + n * (P2 * C2), tid);
  float wa[R2T];
  const float* __restrict__ wp = packed + PKT3 + (int64_t)(cls * 2 + cit) * R2T * 64 + lane;
#pragma unroll
  for (int r = 0; r < R2T; ++r) wa[r] = wp[r * 64];
  for (;;) {
    st.store(plane, tid);
    __syncthreads();
    const int64_t nn = n + slots;
    if (nn < N) st.load(dz2 + nn * (P2 * C2), tid);                    // uniform
    for (int t = slot; t * 16 < npix; t += 4) {
      const int lp = t * 16 + j, p = min(lp, npix - 1);
      const int u = p / nv, v = p - u * nv;
      const float* base = plane + ((u + 2 - py) * GPW + (v + 2 - px)) * GCS + 4 * kq;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        const int off = -((ab >> 1) * GPW + (ab & 1)) * GCS;
#pragma unroll
        for (int hc = 0; hc < 4; hc += 2) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(base + off + 16 * hc);
          const f32x4 bw = *reinterpret_cast<const f32x4*>(base + off + 16 * hc + 16);
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) {
            acc = mfma16(wa[(ab * 4 + hc) * 4 + sp], bv[sp], acc);
            acc_b = mfma16(wa[(ab * 4 + hc + 1) * 4 + sp], bw[sp], acc_b);
          }
        }
      }
      acc += acc_b;
      if (lp < npix) {
        const int pos = (2 * u + 1 - py) * W1 + 2 * v + 1 - px;
        const int64_t o = (n * P1 + pos) * C1 + cit * 16 + 4 * kq;
        const f32x4 m = *reinterpret_cast<const f32x4*>(y1 + o);
        f32x4 out;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = m[r] > 0.f ? acc[r] : 0.f;
        *reinterpret_cast<f32x4*>(dz1 + o) = out;
      }
    }
    if (nn >= N) break;
    n = nn;
    __syncthreads();
  }
}

// ---- wgrad of conv2 / conv3: dW[co][ci][tap] = sum_{n, pos} dz[n][pos][co] x[n][pixel(pos, tap)][ci] ----
// CIN input channels, HI x WI input pixels, KH x KW taps with stride S (padding 1), 12 x 9 positions;
// NCHW_DZ: dz = g * (y > 0) from the flattened conv3 output (else dz as written by dgrad3).
// Workgroup = (image group, 16-channel tile ct of co); NW waves own TPW = (KH KW CIN / 16) / NW
// column tiles (tap, 16 input channels) each; one partial [co tile][K] + bias sums per workgroup.
template <int CIN, int HI, int WI, int KH, int KW, int S, int NW, bool NCHW_DZ>
__global__ __launch_bounds__(NW * 64) void dqn_wgrad23_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ y,
                                                             float* __restrict__ partial, int64_t N,
                                                             int G) {
  constexpr int PW = WI + 2, PH = HI + 2, CS = CIN + 4, Q = CIN / 4, NT = KH * KW * (CIN / 16);
  constexpr int TPW = NT / NW, K = CIN * KH * KW, NTH = NW * 64;
  static_assert(NT % NW == 0, "column tiles divide over the waves");
  constexpr int NV = PH * PW * Q, NIT = (NV + NTH - 1) / NTH;
  constexpr int COUT = 64, PART = COUT * K + COUT;
  __shared__ __attribute__((aligned(16))) float plane[PH * PW * CS];
  __shared__ __attribute__((aligned(16))) float dzs[P2 * 16];
  __shared__ int pb[P2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int grp = blockIdx.x >> 2, ct = blockIdx.x & 3;
  for (int i = tid; i < P2; i += NTH) pb[i] = ((S * (i / W2)) * PW + S * (i % W2)) * CS;
  for (int i = tid; i < PH * PW * CS; i += NTH) plane[i] = 0.f;        // border stays zero
  int toff[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wave * TPW + i, tap = t / (CIN / 16), cit = t - tap * (CIN / 16);
    toff[i] = ((tap / KW) * PW + tap % KW) * CS + cit * 16 + j;
  }
  f32x4 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;
  for (int64_t n = (int64_t)grp * G; n < min((int64_t)(grp + 1) * G, N); ++n) {
    __syncthreads();                       // zero fill / the previous image's readers are done
    // interior pixels of the input plane
    {
      const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(x + n * (HI * WI * CIN));
      f32x4 v[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int f = min(tid + k * NTH, NV - 1);
        const int pix = f / Q, py = pix / PW, px = pix - py * PW;
        const bool in = (py >= 1) && (py <= HI) && (px >= 1) && (px <= WI);
        v[k] = src[in ? ((py - 1) * WI + (px - 1)) * Q + (f - pix * Q) : 0];
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int f = tid + k * NTH;
        const int pix = f / Q, py = pix / PW, px = pix - py * PW;
        const bool in = (py >= 1) && (py <= HI) && (px >= 1) && (px <= WI);
        if (f < NV && in) *reinterpret_cast<f32x4*>(plane + pix * CS + 4 * (f - pix * Q)) = v[k];
      }
    }
    // dz of this output-channel tile: dzs[pos][16]
    if (NCHW_DZ) {
      const float* __restrict__ gs = g + n * (COUT * P2) + ct * 16 * P2;
      const float* __restrict__ ys = y + n * (COUT * P2) + ct * 16 * P2;
      for (int e = tid; e < 16 * P2; e += NTH) {
        const int c = e / P2, pos = e - c * P2;
        dzs[pos * 16 + c] = ys[e] > 0.f ? gs[e] : 0.f;
      }
    } else {
      for (int e = tid; e < 4 * P2; e += NTH) {
        const int pos = e >> 2, qd = e & 3;
        *reinterpret_cast<f32x4*>(dzs + pos * 16 + 4 * qd) =
            *reinterpret_cast<const f32x4*>(g + (n * P2 + pos) * COUT + ct * 16 + 4 * qd);
      }
    }
    __syncthreads();
    if (wave == 0) {                       // bias gradient: column sums of dzs
      float s = 0.f;
      for (int st = 0; st < P2 / 4; ++st) s += dzs[(4 * st + kq) * 16 + j];
      s += __shfl_xor(s, 16, kWave);
      s += __shfl_xor(s, 32, kWave);
      dbacc += s;
    }
#pragma unroll 3
    for (int st = 0; st < P2 / 4; ++st) {
      const int pos = 4 * st + kq;
      const float a = dzs[pos * 16 + j];
      const float* bp = plane + pb[pos];
#pragma unroll
      for (int i = 0; i < TPW; ++i) acc[i] = mfma16(a, bp[toff[i]], acc[i]);
    }
  }
  float* out = partial + (int64_t)grp * PART;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wave * TPW + i, tap = t / (CIN / 16), cit = t - tap * (CIN / 16);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      out[(ct * 16 + 4 * kq + r) * K + (cit * 16 + j) * (KH * KW) + tap] = acc[i][r];
  }
  if (wave == 0 && kq == 0) out[COUT * K + ct * 16 + j] = dbacc;
}

// ---- wgrad of conv1: dW1[co][c][ky][kx] = sum_{n, pos} dz1[n][pos][co] byte[n][c][4 oy + ky][4 ox + kx] ----
// workgroup = (image group, 16-channel tile of co), 8 waves x 2 column tiles of 16 = (c, 2 ky, 8 kx)
constexpr int W1G_THREADS = 512, P1P = (P1 + 3) & ~3;                 // 476 positions (one zero row)
__global__ __launch_bounds__(W1G_THREADS) void dqn_wgrad1_kernel(const uint8_t* __restrict__ obs,
                                                                 const float* __restrict__ dz1,
                                                                 float* __restrict__ partial, int64_t N,
                                                                 int G) {
  __shared__ __attribute__((aligned(16))) uint8_t img[IMG];
  __shared__ __attribute__((aligned(16))) float dzs[P1P * 16];
  __shared__ int pb[P1P];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int grp = blockIdx.x >> 1, ct = blockIdx.x & 1;
  for (int i = tid; i < P1P; i += W1G_THREADS) {
    const int p = min(i, P1 - 1);
    pb[i] = (p / W1) * 4 * W0 + (p % W1) * 4;
  }
  if (tid < 16) dzs[P1 * 16 + tid] = 0.f;                              // the padding position
  int toff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = wave * 2 + i;                                        // (c = t >> 2, ky = 2 (t & 3) + (j >> 3))
    toff[i] = (t >> 2) * HW0 + (2 * (t & 3) + (j >> 3)) * W0 + (j & 7);
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float dbacc = 0.f;
  for (int64_t n = (int64_t)grp * G; n < min((int64_t)(grp + 1) * G, N); ++n) {
    __syncthreads();
    {
      const uint4* __restrict__ src = reinterpret_cast<const uint4*>(obs + n * IMG);
      for (int i = tid; i < IMG / 16; i += W1G_THREADS) reinterpret_cast<uint4*>(img)[i] = src[i];
      for (int e = tid; e < 4 * P1; e += W1G_THREADS) {
        const int pos = e >> 2, qd = e & 3;
        *reinterpret_cast<f32x4*>(dzs + pos * 16 + 4 * qd) =
            *reinterpret_cast<const f32x4*>(dz1 + (n * P1 + pos) * C1 + ct * 16 + 4 * qd);
      }
    }
    __syncthreads();
    if (wave == 0) {
      float s = 0.f;
      for (int st = 0; st < P1P / 4; ++st) s += dzs[(4 * st + kq) * 16 + j];
      s += __shfl_xor(s, 16, kWave);
      s += __shfl_xor(s, 32, kWave);
      dbacc += s;
    }
#pragma unroll 7
    for (int st = 0; st < P1P / 4; ++st) {
      const int pos = 4 * st + kq;
      const float a = dzs[pos * 16 + j];
      const uint8_t* bp = img + pb[pos];
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = mfma16(a, (float)bp[toff[i]], acc[i]);
    }
  }
  float* out = partial + (int64_t)grp * PART1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(ct * 16 + 4 * kq + r) * K1 + (wave * 2 + i) * 16 + j] = acc[i][r];
  if (wave == 0 && kq == 0) out[C1 * K1 + ct * 16 + j] = dbacc;
}

// ---- the partials of the three layers, summed over the image groups in group order ------------
struct BwdReduce {
  const float* part[3];
  float* dw[3];
  float* db[3];
  int groups[3], nw[3], nb[3];
  float scale[3];
};

__global__ __launch_bounds__(256) void dqn_bwd_reduce_kernel(BwdReduce a) {
  int e = blockIdx.x * 256 + threadIdx.x;
  int l = 0;
  while (l < 3 && e >= a.nw[l] + a.nb[l]) {
    e -= a.nw[l] + a.nb[l];
    ++l;
  }
  if (l == 3) return;
  const int stride = a.nw[l] + a.nb[l];
  const float* __restrict__ p = a.part[l] + e;
  float s = 0.f;
  int g = 0;
  for (; g + 8 <= a.groups[l]; g += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(g + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; g < a.groups[l]; ++g) s += p[(int64_t)g * stride];
  if (e < a.nw[l]) a.dw[l][e] = s * a.scale[l];        // conv1's input is byte * scale
  else a.db[l][e - a.nw[l]] = s;
}

// images per workgroup so that a layer's grid stays near one wave of workgroups
int group_size(int64_t N, int tiles) {
  int g = (int)((N * tiles + 255) / 256);
  return g < 1 ? 1 : g;
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

// workspace: packed transposed weights | dz2 [N][108][64] | dz1 [N][475][32] | partials of the three layers
static void bwd_layout(int64_t N, int64_t off[6], int grp[3], int G[3]) {
  G[0] = group_size(N, 2); G[1] = group_size(N, 4); G[2] = group_size(N, 4);
  for (int l = 0; l < 3; ++l) grp[l] = (int)((N + G[l] - 1) / G[l]);
  off[0] = 0;                                       // packed
  off[1] = off[0] + PACKED_BWD;                     // dz2
  off[2] = off[1] + N * (int64_t)(P2 * C2);         // dz1
  off[3] = off[2] + N * (int64_t)(P1 * C1);         // partial 1
  off[4] = off[3] + (int64_t)grp[0] * PART1;        // partial 2
  off[5] = off[4] + (int64_t)grp[1] * PART2;        // partial 3
}

extern "C" int64_t rlpyt_dqn_convs_bwd_workspace_floats(int64_t N) {
  if (N <= 0) return 0;
  int64_t off[6];
  int grp[3], G[3];
  bwd_layout(N, off, grp, G);
  return off[5] + (int64_t)grp[2] * PART3;
}

extern "C" int rlpyt_dqn_convs_bwd_f32(const uint8_t* obs, int64_t N, const float* w2, const float* w3,
                                       const float* y1, const float* y2, const float* y3,
                                       const float* g3, float scale, float* workspace, float* dw1,
                                       float* db1, float* dw2, float* db2, float* dw3, float* db3,
                                       rlpyt_stream_t stream) {
  RL_CHECK_ARG(N >= 0 && N <= (1 << 20), RLPYT_EINVAL, "rlpyt_dqn_convs_bwd_f32: bad sizes");
  if (N == 0) return RLPYT_OK;
  RL_CHECK_ARG(obs && w2 && w3 && y1 && y2 && y3 && g3 && workspace && dw1 && db1 && dw2 && db2 && dw3 &&
                   db3,
               RLPYT_EINVAL, "rlpyt_dqn_convs_bwd_f32: null pointer");
  RL_CHECK_ARG(RL_ALIGNED16(obs) && RL_ALIGNED16(y1) && RL_ALIGNED16(y2) && RL_ALIGNED16(workspace),
               RLPYT_ESHAPE, "rlpyt_dqn_convs_bwd_f32: obs / y1 / y2 / workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int64_t off[6];
  int grp[3], G[3];
  bwd_layout(N, off, grp, G);
  float* packed = workspace + off[0];
  float* dz2 = workspace + off[1];
  float* dz1 = workspace + off[2];
  float* part1 = workspace + off[3];
  float* part2 = workspace + off[4];
  float* part3 = workspace + off[5];
  RL_LAUNCH(dqn_pack_bwd_weights_kernel, dim3((PACKED_BWD + 255) / 256), dim3(256), 0, s, w2, w3, packed);
  RL_LAUNCH_CHECK();
  const int64_t slots = std::min<int64_t>(N, kPersistImages);
  RL_LAUNCH(dqn_dgrad3_kernel, dim3((unsigned)(slots * 4)), dim3(DG3_THREADS), 0, s, g3, y3, y2, packed, dz2, N);
  RL_LAUNCH_CHECK();
  RL_LAUNCH((dqn_wgrad23_kernel<C2, H2, W2, 3, 3, 1, 12, true>), dim3((unsigned)(grp[2] * 4)), dim3(12 * 64),
            0, s, y2, g3, y3, part3, N, G[2]);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(dqn_dgrad2_kernel, dim3((unsigned)(slots * 4)), dim3(DG2_THREADS), 0, s, dz2, y1, packed, dz1, N);
  RL_LAUNCH_CHECK();
  RL_LAUNCH((dqn_wgrad23_kernel<C1, H1, W1, 4, 4, 2, 8, false>), dim3((unsigned)(grp[1] * 4)), dim3(8 * 64),
            0, s, y1, dz2, nullptr, part2, N, G[1]);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(dqn_wgrad1_kernel, dim3((unsigned)(grp[0] * 2)), dim3(W1G_THREADS), 0, s, obs, dz1, part1, N, G[0]);
  RL_LAUNCH_CHECK();
  BwdReduce a;
  a.part[0] = part1; a.part[1] = part2; a.part[2] = part3;
  a.dw[0] = dw1; a.dw[1] = dw2; a.dw[2] = dw3;
  a.db[0] = db1; a.db[1] = db2; a.db[2] = db3;
  a.groups[0] = grp[0]; a.groups[1] = grp[1]; a.groups[2] = grp[2];
  a.nw[0] = C1 * K1; a.nw[1] = C2 * K2; a.nw[2] = C3 * K3;
  a.nb[0] = C1; a.nb[1] = C2; a.nb[2] = C3;
  a.scale[0] = scale; a.scale[1] = 1.f; a.scale[2] = 1.f;
  const int total = PART1 + PART2 + PART3;
  RL_LAUNCH(dqn_bwd_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
