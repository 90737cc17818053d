// conv2 / conv3 of the DQN-family stack on the bf16 matrix pipe ("bf16x6", round 6):
//   Conv2d(32, 64, 4, stride 2, pad 1) on y1 [N][475][32]  ->  y2 [N][108][64]      (channels-last)
//   Conv2d(64, 64, 3, stride 1, pad 1) on y2               ->  out [N][64][108]     (flatten order)
// (rlpyt/models/dqn/atari_dqn_model.py:30-37, atari_r2d1_model.py:33-41; the forward of every pass).
// csrc/dqn_convs.hip runs them on v_mfma_f32_16x16x4_f32 -- 32 cycles for K = 4 -- and is bound by that
// pipe from a few hundred images on (0.6-0.7 of its issue bound at R2D1's 2 560-5 440 images per pass).
// Here both operands are split into three bf16 pieces by truncation (hi + mid + lo == x exactly: 3 x 8
// significand bits) and the six products of order <= 2 are accumulated in f32, smallest first, on
// v_mfma_f32_16x16x32_bf16 (16 cycles for K = 32): 6 x 16 cycles per 32 K against 8 x 32 -- 2.7 x less
// matrix-pipe time; the three dropped products are together <= 2^-24 |ab| (one f32 rounding), the scheme
// of conv2_fwd_x6_kernel / gemm_nt_x6_kernel (tests hold it to torch-f32's own error level).
//
// Structure.  Workgroup (256 threads) = (image slot, 16-channel tile ct, half of the 12 output rows); it is
// persistent over the images (slot, slot + slots, ...).  Its four waves are (position tile, K half): a
// wave holds the weight pieces of ITS half of the contraction for ct in 96 / 108 VGPRs (8 / 9 K-32 steps x 3
// pieces x 16 bytes, from a packed copy made on the stream), so a workgroup needs two waves per position
// tile, whose partial tiles meet in LDS.  The half image (6 output rows = 54 positions = two passes of 2 x
// 16) needs 14 x 21 (conv2) / 8 x 11 (conv3) input pixels incl. the zero border: staged from the f32
// activations as three bf16 planes [piece][pixel][CIN + 8] (70.6 / 38.0 KB -> two or more workgroups per
// CU: one stages while another contracts; no register prefetch needed).  A lane's B operand of a step = the
// 8 consecutive channels 8 kb .. of its position's pixel under the step's tap: one ds_read_b128 per piece.
#include <algorithm>

#include "common.h"

namespace rlpyt {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_bf16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the high halves (= truncated bf16) of two f32, earlier element in the low half
__device__ __forceinline__ uint32_t pack_hi16(float lo_elem, float hi_elem) {
  return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
// x - bf16_trunc(x): exact
__device__ __forceinline__ float bf16_rem(float x) {
  return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
}

constexpr int H2 = 12, W2 = 9, P2 = H2 * W2;        // output positions of both layers
constexpr int HROWS = 6, HPOS = HROWS * W2;          // a workgroup's half: 6 output rows, 54 positions
constexpr int X6_THREADS = 256;
constexpr int kX6Slots = 256;                        // image slots of the persistent grid

template <int CIN, int KH, int KW>
struct X6Geom {
  static constexpr int CH = CIN / 32;                // K-32 steps per tap
  static constexpr int NS = KH * KW * CH;            // steps of the whole contraction: 16 | 18
  static constexpr int NSH = NS / 2;                 // per K half: 8 | 9
  static constexpr int K = CIN * KH * KW;
  // packed weights: [ct 4][khalf 2][step NSH][piece 3][lane 64] x uint4
  static constexpr int PACK_U4 = 4 * 2 * NSH * 3 * 64;
};

// ---- weights -> bf16 pieces in operand order (both layers in ONE launch, on the stream) ----------
// element e of lane (j, kb) at step s of (ct, khalf): co = 16 ct + j, tap = s' / CH, ci = 32 (s' % CH) +
// 8 kb + e with s' = khalf NSH + s;  w[co][ci][ky][kx] (torch layout)
template <int CIN, int KH, int KW>
__device__ __forceinline__ void x6_pack_unit(const float* __restrict__ w, uint4* __restrict__ packed, int u) {
  typedef X6Geom<CIN, KH, KW> G;
  const int lane = u & 63, j = lane & 15, kb = lane >> 4;
  const int s = (u >> 6) % G::NSH, kh = ((u >> 6) / G::NSH) & 1, ct = (u >> 6) / (2 * G::NSH);
  const int sp = kh * G::NSH + s, tap = sp / G::CH, c0 = 32 * (sp % G::CH) + 8 * kb;
  const float* __restrict__ src = w + (int64_t)(16 * ct + j) * G::K + (int64_t)c0 * (KH * KW) + tap;
  float x[8], r1[8], r2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    x[e] = src[e * (KH * KW)];
    r1[e] = bf16_rem(x[e]);
    r2[e] = bf16_rem(r1[e]);
  }
  uint4* dst = packed + ((int64_t)((ct * 2 + kh) * G::NSH + s) * 3) * 64 + lane;
  dst[0] = uint4{pack_hi16(x[0], x[1]), pack_hi16(x[2], x[3]), pack_hi16(x[4], x[5]), pack_hi16(x[6], x[7])};
  dst[64] = uint4{pack_hi16(r1[0], r1[1]), pack_hi16(r1[2], r1[3]), pack_hi16(r1[4], r1[5]), pack_hi16(r1[6], r1[7])};
  dst[128] = uint4{pack_hi16(r2[0], r2[1]), pack_hi16(r2[2], r2[3]), pack_hi16(r2[4], r2[5]), pack_hi16(r2[6], r2[7])};
}
constexpr int kPackUnits2 = 4 * 2 * X6Geom<32, 4, 4>::NSH * 64, kPackUnits3 = 4 * 2 * X6Geom<64, 3, 3>::NSH * 64;

__global__ __launch_bounds__(256) void dqn_x6_pack_kernel(const float* __restrict__ w2, const float* __restrict__ w3,
                                                          uint4* __restrict__ p2, uint4* __restrict__ p3) {
  const int u = blockIdx.x * 256 + threadIdx.x;        // (layer, ct, khalf, step, lane)
  if (u < kPackUnits2) x6_pack_unit<32, 4, 4>(w2, p2, u);
  else if (u < kPackUnits2 + kPackUnits3) x6_pack_unit<64, 3, 3>(w3, p3, u - kPackUnits2);
}

// ---- the layer ---------------------------------------------------------------------------------
// CIN channels, HI x WI input pixels (channels-last f32), KH x KW taps, stride S, padding 1, 12 x 9 outputs,
// 64 output channels; NCHW_OUT: out[n][channel][position] instead of [n][position][channel]
template <int CIN, int HI, int WI, int KH, int KW, int S, bool NCHW_OUT>
__global__ __launch_bounds__(X6_THREADS) void dqn_conv23_x6_kernel(const float* __restrict__ x,
                                                                  const uint4* __restrict__ packed,
                                                                  const float* __restrict__ bvec,
                                                                  float* __restrict__ out, int64_t N) {
  typedef X6Geom<CIN, KH, KW> G;
  constexpr int PW = WI + 2;                           // padded width
  constexpr int NR = S * (HROWS - 1) + KH;             // padded rows a half needs: 14 | 8
  constexpr int NPX = NR * PW;                         // 294 | 88 pixels
  constexpr int CINB = CIN * 2 + 16;                   // bytes per (piece, pixel): channels + pad (80 | 144)
  constexpr int PLB = NPX * CINB;                      // bytes per piece plane
  constexpr int Q = CIN / 4, NV = NPX * Q;             // float4 units of a half plane
  constexpr int NIT = (NV + X6_THREADS - 1) / X6_THREADS;
  constexpr int COUT = 64;
  __shared__ __attribute__((aligned(16))) uint8_t plane[3 * PLB];
  __shared__ __attribute__((aligned(16))) f32x4 red[2][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kb = lane >> 4;
  const int tl = wave & 1, kh = wave >> 1;             // position tile of a pass, K half
  const int64_t slots = gridDim.x >> 3;
  int64_t n = blockIdx.x >> 3;
  const int ct = (int)(blockIdx.x & 3), half = (int)((blockIdx.x >> 2) & 1);
  // this wave's weight pieces: NSH steps x 3 pieces
  uint4 wa[G::NSH][3];
  {
    const uint4* __restrict__ wp = packed + ((int64_t)(ct * 2 + kh) * G::NSH * 3) * 64 + lane;
#pragma unroll
    for (int s = 0; s < G::NSH; ++s)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) wa[s][pc] = wp[(s * 3 + pc) * 64];
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>(bvec + ct * 16 + 4 * kb);
  // staging map of this thread: unit f = (pixel, channel quad) of the half plane
  int ssrc[NIT], sdst[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int f = min(tid + k * X6_THREADS, NV - 1);
    const int px = f / Q, qd = f - px * Q, r = px / PW, c = px - r * PW;
    const int iy = S * HROWS * half + r - 1, ix = c - 1;
    const bool in = (iy >= 0) && (iy < HI) && (ix >= 0) && (ix < WI);
    ssrc[k] = in ? (iy * WI + ix) * CIN + 4 * qd : -1;
    sdst[k] = px * CINB + 8 * qd;
  }
  // per pass p: this lane's position and the byte offset of its pixel under tap (0, 0)
  int q_of[2], b_of[2];
  bool live[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int lp = (2 * p + tl) * 16 + j;
    const int lpc = min(lp, HPOS - 1), oyl = lpc / W2, ox = lpc - oyl * W2;
    live[p] = lp < HPOS;
    q_of[p] = (HROWS * half + oyl) * W2 + ox;
    b_of[p] = ((S * oyl) * PW + S * ox) * CINB + 16 * kb;
  }
  for (;;) {
    // ---- stage the half plane: f32 -> three bf16 pieces -------------------------------------
    {
      const float* __restrict__ src = x + n * (int64_t)(HI * WI * CIN);
      f32x4 v[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        v[k] = *reinterpret_cast<const f32x4*>(src + max(ssrc[k], 0));
        if (ssrc[k] < 0) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        if (tid + k * X6_THREADS < NV) {
          float r1[4], r2[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            r1[e] = bf16_rem(v[k][e]);
            r2[e] = bf16_rem(r1[e]);
          }
          uint8_t* d = plane + sdst[k];
          *reinterpret_cast<uint2*>(d) = uint2{pack_hi16(v[k][0], v[k][1]), pack_hi16(v[k][2], v[k][3])};
          *reinterpret_cast<uint2*>(d + PLB) = uint2{pack_hi16(r1[0], r1[1]), pack_hi16(r1[2], r1[3])};
          *reinterpret_cast<uint2*>(d + 2 * PLB) = uint2{pack_hi16(r2[0], r2[1]), pack_hi16(r2[2], r2[3])};
        }
      }
    }
    __syncthreads();
    // ---- two passes of 2 x 16 positions ------------------------------------------------------
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
      const uint8_t* base = plane + b_of[p];
#pragma unroll
      for (int s = 0; s < G::NSH; ++s) {
        const int sp = kh * G::NSH + s;                  // (kh is wave-uniform; both values unrolled below)
        const int tap = sp / G::CH, chalf = sp - tap * G::CH;
        const int ky = tap / KW, kx = tap - ky * KW;
        const uint8_t* bp = base + (ky * PW + kx) * CINB + chalf * 64;
        const uint4 b0 = *reinterpret_cast<const uint4*>(bp);
        const uint4 b1 = *reinterpret_cast<const uint4*>(bp + PLB);
        const uint4 b2 = *reinterpret_cast<const uint4*>(bp + 2 * PLB);
        // six products, smallest first, on two accumulator chains
        acc = mfma_bf16(wa[s][2], b0, acc);
        acc_b = mfma_bf16(wa[s][0], b2, acc_b);
        acc = mfma_bf16(wa[s][1], b1, acc);
        acc_b = mfma_bf16(wa[s][1], b0, acc_b);
        acc = mfma_bf16(wa[s][0], b1, acc);
        acc_b = mfma_bf16(wa[s][0], b0, acc_b);
      }
      acc += acc_b;
      if (kh == 1) red[tl][lane] = acc;                  // uniform per wave
      __syncthreads();
      if (kh == 0) {
        acc += red[tl][lane];
        if (live[p]) {
          const int q = q_of[p];
          if (NCHW_OUT) {
            float* o = out + n * (int64_t)(COUT * P2) + (ct * 16 + 4 * kb) * P2 + q;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r * P2] = fmaxf(acc[r] + bias[r], 0.f);
          } else {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[r] + bias[r], 0.f);
            *reinterpret_cast<f32x4*>(out + (n * P2 + q) * COUT + ct * 16 + 4 * kb) = o;
          }
        }
      }
      __syncthreads();                                   // red / (last pass) the plane are free again
    }
    n += slots;
    if (n >= N) break;
  }
}


// ---- the same layers on 32 x 32 tiles ------------------------------------------------------------
// dqn_conv23_x6_kernel needs three 16-byte LDS reads per lane for the six v_mfma_f32_16x16x32_bf16 of a
// K-32 step: with its four waves that is 128 bytes per clock, the CU's whole LDS bandwidth -- the kernel is
// LDS-read-bound from a few hundred images on.  v_mfma_f32_32x32x16_bf16 takes the same 16 bytes per lane
// and operand for twice the products (32 output channels x 32 positions x 16 K in 32 cycles): half the LDS
// bytes per MAC.  Workgroup (256 threads) = (image slot, half of the 64 output channels, PART of the 12
// output rows: 3 rows for conv2, 6 for conv3), persistent over the images; its four waves are the four
// QUARTERS of the contraction (a wave keeps the weight pieces of its quarter for its 32 channels in 96 /
// 108 VGPRs), whose partial tiles meet in LDS in quarter order.  The part's input pixels incl. zero border
// (8 x 21 for conv2, 8 x 11 for conv3) are staged as three bf16 planes exactly as above: 40.3 / 38.0 KB +
// 12 KB of partial tiles -> three workgroups per CU, and a plane is staged for 32 output channels instead of
// 16 (conv2: 10.8 K float4 units per image instead of 18.8 K, conv3: 5.6 K instead of 11.3 K).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32_bf16(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int CIN, int KH, int KW>
struct T32Geom {
  static constexpr int CH16 = CIN / 16;              // K-16 steps per tap
  static constexpr int NS = KH * KW * CH16;          // 32 | 36
  static constexpr int NSQ = NS / 4;                 // per K quarter: 8 | 9
  static constexpr int K = CIN * KH * KW;
  // packed weights: [co half 2][K quarter 4][step NSQ][piece 3][lane 64] x uint4
  static constexpr int PACK_U4 = 2 * 4 * NSQ * 3 * 64;
  static_assert(NS % 4 == 0, "the contraction splits into four quarters of whole K-16 steps");
};

// element e of lane (i = lane & 31, kb = lane >> 5) at step s of (ch, kq): co = 32 ch + i, s' = kq NSQ + s,
// tap = s' / CH16, ci = 16 (s' % CH16) + 8 kb + e
template <int CIN, int KH, int KW>
__device__ __forceinline__ void t32_pack_unit(const float* __restrict__ w, uint4* __restrict__ packed, int u) {
  typedef T32Geom<CIN, KH, KW> G;
  const int lane = u & 63, i = lane & 31, kb = lane >> 5;
  const int s = (u >> 6) % G::NSQ, kq = ((u >> 6) / G::NSQ) & 3, ch = (u >> 6) / (4 * G::NSQ);
  const int sp = kq * G::NSQ + s, tap = sp / G::CH16, c0 = 16 * (sp % G::CH16) + 8 * kb;
  const float* __restrict__ src = w + (int64_t)(32 * ch + i) * G::K + (int64_t)c0 * (KH * KW) + tap;
  float x[8], r1[8], r2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    x[e] = src[e * (KH * KW)];
    r1[e] = bf16_rem(x[e]);
    r2[e] = bf16_rem(r1[e]);
  }
  uint4* dst = packed + ((int64_t)((ch * 4 + kq) * G::NSQ + s) * 3) * 64 + lane;
  dst[0] = uint4{pack_hi16(x[0], x[1]), pack_hi16(x[2], x[3]), pack_hi16(x[4], x[5]), pack_hi16(x[6], x[7])};
  dst[64] = uint4{pack_hi16(r1[0], r1[1]), pack_hi16(r1[2], r1[3]), pack_hi16(r1[4], r1[5]), pack_hi16(r1[6], r1[7])};
  dst[128] = uint4{pack_hi16(r2[0], r2[1]), pack_hi16(r2[2], r2[3]), pack_hi16(r2[4], r2[5]), pack_hi16(r2[6], r2[7])};
}
constexpr int kT32Units2 = 2 * 4 * T32Geom<32, 4, 4>::NSQ * 64, kT32Units3 = 2 * 4 * T32Geom<64, 3, 3>::NSQ * 64;

__global__ __launch_bounds__(256) void dqn_t32_pack_kernel(const float* __restrict__ w2, const float* __restrict__ w3,
                                                           uint4* __restrict__ p2, uint4* __restrict__ p3) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u < kT32Units2) t32_pack_unit<32, 4, 4>(w2, p2, u);
  else if (u < kT32Units2 + kT32Units3) t32_pack_unit<64, 3, 3>(w3, p3, u - kT32Units2);
}

// PROWS output rows per workgroup (12 / PROWS parts per image)
template <int CIN, int HI, int WI, int KH, int KW, int S, int PROWS, bool NCHW_OUT>
__global__ __launch_bounds__(X6_THREADS) void dqn_conv23_t32_kernel(const float* __restrict__ x,
                                                                   const uint4* __restrict__ packed,
                                                                   const float* __restrict__ bvec,
                                                                   float* __restrict__ out, int64_t N) {
  typedef T32Geom<CIN, KH, KW> G;
  constexpr int PARTS = H2 / PROWS, PPOS = PROWS * W2;   // 4 | 2 parts of 27 | 54 positions
  constexpr int NPASS = (PPOS + 31) / 32;              // 1 | 2 passes of 32 positions
  constexpr int PW = WI + 2;
  constexpr int NR = S * (PROWS - 1) + KH;             // padded rows a part needs: 8 | 8
  constexpr int NPX = NR * PW;                         // 168 | 88 pixels
  constexpr int CINB = CIN * 2 + 16;                   // bytes per (piece, pixel)
  constexpr int PLB = NPX * CINB;
  constexpr int Q = CIN / 4, NV = NPX * Q;
  constexpr int NIT = (NV + X6_THREADS - 1) / X6_THREADS;
  constexpr int COUT = 64;
  __shared__ __attribute__((aligned(16))) uint8_t plane[3 * PLB];
  __shared__ float red[3][16][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int kq = __builtin_amdgcn_readfirstlane(tid >> 6);      // this wave's quarter of the contraction
  const int j = lane & 31, kb = lane >> 5;
  const int64_t slots = gridDim.x / (2 * PARTS);
  int64_t n = blockIdx.x / (2 * PARTS);
  const int ch = (int)(blockIdx.x & 1), part = (int)((blockIdx.x >> 1) % PARTS);
  uint4 wa[G::NSQ][3];
  {
    const uint4* __restrict__ wp = packed + ((int64_t)(ch * 4 + kq) * G::NSQ * 3) * 64 + lane;
#pragma unroll
    for (int s = 0; s < G::NSQ; ++s)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) wa[s][pc] = wp[(s * 3 + pc) * 64];
  }
  // D of the 32 x 32 MFMA: column = lane & 31 (position), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  f32x4 bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(bvec + 32 * ch + 8 * g + 4 * kb);
  int ssrc[NIT], sdst[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int f = min(tid + k * X6_THREADS, NV - 1);
    const int px = f / Q, qd = f - px * Q, r = px / PW, c = px - r * PW;
    const int iy = S * PROWS * part + r - 1, ix = c - 1;
    const bool in = (iy >= 0) && (iy < HI) && (ix >= 0) && (ix < WI);
    ssrc[k] = in ? (iy * WI + ix) * CIN + 4 * qd : -1;
    sdst[k] = px * CINB + 8 * qd;
  }
  int q_of[NPASS], b_of[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int lp = 32 * p + j;
    const int lpc = min(lp, PPOS - 1), oyl = lpc / W2, ox = lpc - oyl * W2;
    q_of[p] = (PROWS * part + oyl) * W2 + ox;
    b_of[p] = ((S * oyl) * PW + S * ox) * CINB + 16 * kb;
  }
  // the part's pixels of image n as float4 units, requested one image ahead: the loads of image n + slots are
  // in flight behind the MFMAs of image n (a workgroup's round trip to HBM / L2 per image was what bounded
  // the kernel at hundreds of images: staging 5-6 loads per thread, then 48-108 MFMAs per wave)
  f32x4 v[NIT];
#define RLPYT_T32_LOAD(ni)                                                                     \
  {                                                                                            \
    const float* __restrict__ src_ = x + (ni) * (int64_t)(HI * WI * CIN);                      \
    _Pragma("unroll") for (int k_ = 0; k_ < NIT; ++k_)                                         \
      v[k_] = *reinterpret_cast<const f32x4*>(src_ + max(ssrc[k_], 0));                        \
  }
  RLPYT_T32_LOAD(n)
  for (;;) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (tid + k * X6_THREADS < NV) {
        const f32x4 u = ssrc[k] < 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : v[k];
        float r1[4], r2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          r1[e] = bf16_rem(u[e]);
          r2[e] = bf16_rem(r1[e]);
        }
        uint8_t* d = plane + sdst[k];
        *reinterpret_cast<uint2*>(d) = uint2{pack_hi16(u[0], u[1]), pack_hi16(u[2], u[3])};
        *reinterpret_cast<uint2*>(d + PLB) = uint2{pack_hi16(r1[0], r1[1]), pack_hi16(r1[2], r1[3])};
        *reinterpret_cast<uint2*>(d + 2 * PLB) = uint2{pack_hi16(r2[0], r2[1]), pack_hi16(r2[2], r2[3])};
      }
    }
    __syncthreads();
    const int64_t nn = n + slots;
    if (nn < N) RLPYT_T32_LOAD(nn)                       // (uniform)
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const uint8_t* base = plane + b_of[p];
#pragma unroll
      for (int s = 0; s < G::NSQ; ++s) {
        const int sp = kq * G::NSQ + s;                  // (kq is wave-uniform: scalar arithmetic)
        const int tap = sp / G::CH16, c16 = sp - tap * G::CH16;
        const int ky = tap / KW, kx = tap - ky * KW;
        const uint8_t* bp = base + (ky * PW + kx) * CINB + c16 * 32;
        const uint4 b0 = *reinterpret_cast<const uint4*>(bp);
        const uint4 b1 = *reinterpret_cast<const uint4*>(bp + PLB);
        const uint4 b2 = *reinterpret_cast<const uint4*>(bp + 2 * PLB);
        // six products, smallest first
        acc = mfma32_bf16(wa[s][2], b0, acc);
        acc = mfma32_bf16(wa[s][0], b2, acc);
        acc = mfma32_bf16(wa[s][1], b1, acc);
        acc = mfma32_bf16(wa[s][1], b0, acc);
        acc = mfma32_bf16(wa[s][0], b1, acc);
        acc = mfma32_bf16(wa[s][0], b0, acc);
      }
      if (kq > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[kq - 1][r][lane] = acc[r];
      }
      __syncthreads();
      if (kq == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
        // lanes past the part's last position hold that position's values (clamped pixel): they store the
        // same numbers to the same addresses -- no branch around the stores
        const int q = q_of[p];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = 32 * ch + 8 * g + 4 * kb;
          if (NCHW_OUT) {
            float* o = out + n * (int64_t)(COUT * P2) + co * P2 + q;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r * P2] = fmaxf(acc[4 * g + r] + bias[g][r], 0.f);
          } else {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[4 * g + r] + bias[g][r], 0.f);
            *reinterpret_cast<f32x4*>(out + (n * P2 + q) * COUT + co) = o;
          }
        }
      }
      __syncthreads();                                   // red / (last pass) the plane are free again
    }
    if (nn >= N) break;
    n = nn;
  }
#undef RLPYT_T32_LOAD
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

// RLPYT_DQN_CONV23_T32=0: the 16 x 16 tile kernels (A/B runs); read per call
static bool t32_on() {
  const char* e = getenv("RLPYT_DQN_CONV23_T32");
  return !(e && e[0] == '0');
}
constexpr int kX6PackU4 = X6Geom<32, 4, 4>::PACK_U4 + X6Geom<64, 3, 3>::PACK_U4;
constexpr int kT32PackU4 = T32Geom<32, 4, 4>::PACK_U4 + T32Geom<64, 3, 3>::PACK_U4;

// packed bf16 pieces of w2 and w3 in operand order: [16 x 16 tile order | 32 x 32 tile order] (the same
// 3 x 2 bytes per weight each; a packing launch fills the one the switch selects)
extern "C" int64_t rlpyt_dqn_convs_x6_packed_bytes(void) { return (int64_t)(kX6PackU4 + kT32PackU4) * 16; }

// w2 [64,32,4,4], w3 [64,64,3,3] (torch layouts) -> `packed` (rlpyt_dqn_convs_x6_packed_bytes() bytes)
extern "C" int rlpyt_dqn_convs_x6_pack(const float* w2, const float* w3, void* packed, rlpyt_stream_t stream) {
  RL_CHECK_ARG(w2 && w3 && packed && RL_ALIGNED16(packed), RLPYT_EINVAL,
               "rlpyt_dqn_convs_x6_pack: null / unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  if (t32_on()) {
    uint4* p2 = static_cast<uint4*>(packed) + kX6PackU4;
    uint4* p3 = p2 + T32Geom<32, 4, 4>::PACK_U4;
    RL_LAUNCH(dqn_t32_pack_kernel, dim3((kT32Units2 + kT32Units3 + 255) / 256), dim3(256), 0, s, w2, w3, p2, p3);
    RL_LAUNCH_CHECK();
    return RLPYT_OK;
  }
  uint4* p2 = static_cast<uint4*>(packed);
  uint4* p3 = p2 + X6Geom<32, 4, 4>::PACK_U4;
  RL_LAUNCH(dqn_x6_pack_kernel, dim3((kPackUnits2 + kPackUnits3 + 255) / 256), dim3(256), 0, s, w2, w3, p2, p3);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

// conv2 + conv3 on the bf16x6 scheme: y1 [N][475][32] -> y2 [N][108][64] -> out [N][64*108], weights as
// packed by rlpyt_dqn_convs_x6_pack
extern "C" int rlpyt_dqn_conv23_x6_f32(const float* y1, int64_t N, const void* packed, const float* b2,
                                       const float* b3, float* y2, float* out, rlpyt_stream_t stream) {
  RL_CHECK_ARG(N >= 0 && N <= (1 << 20), RLPYT_EINVAL, "rlpyt_dqn_conv23_x6_f32: bad sizes");
  if (N == 0) return RLPYT_OK;
  RL_CHECK_ARG(y1 && b2 && b3 && packed && y2 && out, RLPYT_EINVAL, "rlpyt_dqn_conv23_x6_f32: null pointer");
  RL_CHECK_ARG(RL_ALIGNED16(y1) && RL_ALIGNED16(y2) && RL_ALIGNED16(out) && RL_ALIGNED16(packed) &&
                   RL_ALIGNED16(b2) && RL_ALIGNED16(b3),
               RLPYT_ESHAPE, "rlpyt_dqn_conv23_x6_f32: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int64_t slots = std::min<int64_t>(N, kX6Slots);
  if (t32_on()) {
    const uint4* p2 = static_cast<const uint4*>(packed) + kX6PackU4;
    const uint4* p3 = p2 + T32Geom<32, 4, 4>::PACK_U4;
    RL_LAUNCH((dqn_conv23_t32_kernel<32, 25, 19, 4, 4, 2, 3, false>), dim3((unsigned)(slots * 8)), dim3(X6_THREADS), 0,
              s, y1, p2, b2, y2, N);
    RL_LAUNCH_CHECK();
    RL_LAUNCH((dqn_conv23_t32_kernel<64, 12, 9, 3, 3, 1, 6, true>), dim3((unsigned)(slots * 4)), dim3(X6_THREADS), 0,
              s, y2, p3, b3, out, N);
    RL_LAUNCH_CHECK();
    return RLPYT_OK;
  }
  const uint4* p2 = static_cast<const uint4*>(packed);
  const uint4* p3 = p2 + X6Geom<32, 4, 4>::PACK_U4;
  RL_LAUNCH((dqn_conv23_x6_kernel<32, 25, 19, 4, 4, 2, false>), dim3((unsigned)(slots * 8)), dim3(X6_THREADS), 0, s,
            y1, p2, b2, y2, N);
  RL_LAUNCH_CHECK();
  RL_LAUNCH((dqn_conv23_x6_kernel<64, 12, 9, 3, 3, 1, true>), dim3((unsigned)(slots * 8)), dim3(X6_THREADS), 0, s,
            y2, p3, b3, out, N);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
