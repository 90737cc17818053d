// f32 GEMM  C[M,N] = A[M,K] * B[N,K]^T  on the bf16 matrix pipe ("bf16x6"), for the trunk of the
// PPO update (rlpyt/models/mlp.py:24-31: Linear(3456, 512); forward x W^T and, with B = W^T, the
// input gradient g W).  hipBLASLt's f32 GEMM runs these at 0.72-0.87 of the f32 MFMA peak -- there
// is no faster f32 instruction on gfx950.  Here both operands are split into three bf16 pieces
// (round to nearest: x == x0 + x1 + x2 exactly) while they pass from HBM to LDS and the six
// products of order <= 2 are accumulated in f32, smallest first; the three dropped products are
// together <= 2^-24 |ab| (one f32 rounding; 2^-27 rms, unbiased) (same scheme and the same accuracy test as
// conv2_fwd_x6_kernel, csrc/conv.hip).  6 x v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace
// 8 x v_mfma_f32_32x32x2_f32 (64 cycles, K = 2): 2.7x less matrix-pipe time per K.
//
// Tiling: workgroup = 128 x 128 of C, 8 waves (2 x 4), wave = 64 x 32 = 2 MFMA tiles (32
// accumulator VGPRs); K-step 16.  LDS stage = {A, B} x 3 pieces x 128 rows x 48 B (32 B of data +
// 16 B pad: the 8-lane groups of a ds_read_b128 then cover all 32 banks), two stages = 73,728 B
// -> two workgroups per CU when there are enough tiles (the forward shape has 256 tiles: one
// workgroup = 2 waves per SIMD; with 4 waves per workgroup it ran at one wave per SIMD and every
// split / LDS phase idled the matrix pipe: 226 us).  Per K-step a thread moves 1 + 1 float4
// HBM -> registers (one step ahead), splits them and writes 6 x 8 B to the other LDS stage; one
// barrier per K-step.  blockIdx -> tile map keeps the column tiles of one row block on one XCD.
// (Round 5 built a variant whose B operand -- the trunk weight, the same for all 64 row blocks --
// arrives pre-split as three bf16 planes made once per optimizer step: bit-identical results, half
// the split VALU, and SLOWER -- forward 148.7 -> 192.7 us in the bench, 175 -> 188 us isolated,
// profiles/r5_ab_presplit_weight.jsonl, r5_gemm_bench_presplit_weight.json: three 8-byte loads per
// thread and step instead of one 16-byte load.  The kernel is bound by its vector-memory requests,
// not by the split VALU beside the MFMAs; removed.)
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace rlpyt {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma32_bf16(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo_elem, float hi_elem) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {lo_elem, hi_elem};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// (x0, x1) -> three packed bf16 pieces, hi + mid + lo == x exactly (finite, no overflow)
__device__ __forceinline__ void split3_rn(float x0, float x1, uint32_t& hi, uint32_t& mid,
                                          uint32_t& lo) {
  hi = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(r0, r1);
  lo = cvt_pk_bf16(r0 - __uint_as_float(mid << 16), r1 - __uint_as_float(mid & 0xffff0000u));
}

// Debug build (-DRLPYT_TIMING): per-wave cycle totals of the phases of the K loop
// (scripts/debug/phase_timing.py gemm_fwd | gemm_dgrad); compiled out of the product.
#ifdef RLPYT_TIMING
__device__ float g_timing_gemm[512 * 16 * 8];
#define RL_T0() long long t_prev_ = clock64(), t_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RL_T(k)                                  \
  {                                              \
    const long long t_now_ = clock64();          \
    t_acc_[k] += t_now_ - t_prev_;               \
    t_prev_ = t_now_;                            \
  }
#define RL_TOUT()                                                                        \
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) {                                     \
    float* dbg_ = g_timing_gemm + ((int64_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8;   \
    for (int k = 0; k < 8; ++k) dbg_[k] = (float)t_acc_[k];                              \
  }
#else
#define RL_T0()
#define RL_T(k)
#define RL_TOUT()
#endif

constexpr int GT = 128;                  // column-tile edge (and row-tile edge of the small variant)
constexpr int G_THREADS = 512;
constexpr int G_BK = 16;                 // K per barrier step
constexpr int G_ROWB = 2 * G_BK + 16;    // bytes per (piece, row) of a K-step: data + 16 B pad

// TM = rows of C per workgroup: 128 (wave = 64 x 32) or 256 (wave = 64 x 64).  The kernel is bound
// by the CU's load path (~9 B/clk/CU measured: 16 KB per K-step of a 128 x 128 tile = 1780 cycles,
// 2.3x the matrix-pipe time); a 256 x 128 tile loads 24 KB for twice the MFMAs.  Used where the
// shape has enough 256-row tiles to fill the chip (the input gradient: 32 x 27); the forward shape
// (4 M outputs) is one 128 x 128 tile per CU.  For TM = 256 three padded LDS stages would need
// 166 KB, so the B operand is stored unpadded (32 B per row) with its two 16-byte K halves swapped
// in every other group of 4 rows -- conflict-free for the 8-lane groups of ds_read_b128 as well.
template <int TM>
__global__ __launch_bounds__(G_THREADS) void gemm_nt_x6_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N,
    int K, int tiles_m, int tiles_n) {
  constexpr bool BSW = TM == 256;                    // swizzled unpadded B
  constexpr int ROWB_B = BSW ? 2 * G_BK : G_ROWB;
  constexpr int PB_A = TM * G_ROWB, PB_B = GT * ROWB_B;      // bytes per piece
  constexpr int OB_A = 3 * PB_A, SB = OB_A + 3 * PB_B;       // B operand offset, stage bytes
  constexpr int NA = TM * 4 / G_THREADS;             // float4 of A per thread and K-step: 1 / 2
  constexpr int WM = TM / 64, WN = 8 / WM;           // wave grid: 2 x 4 / 4 x 2
  constexpr int TNW = GT / WN, NJ = TNW / 32;        // wave tile columns 32 / 64: 1 / 2 MFMA tiles
  __shared__ __attribute__((aligned(16))) uint8_t lds[3 * SB];
  static_assert(3 * SB <= 160 * 1024, "three stages fit the LDS");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;     // wave tile: rows 64 wm .. + 63, columns TNW wn ..
  // XCD-aware tile map.  Workgroup ids go round-robin over the 8 XCDs (4 MB of L2 each, 32 CUs =
  // 32 concurrent workgroups).  Every XCD owns tiles_m / 8 row blocks and walks them in panels
  // of (its row blocks) x (4 column tiles): the workgroups running together then share
  // few A row-block tiles and 4 B column tiles instead of one A tile and 27 different B tiles
  // (all of W^T, 7 MB: re-streamed past the L2 for every row block).
  const int n_tiles = tiles_m * tiles_n;
  int tm, tn;
  if ((tiles_m & 7) == 0) {
    const int R = tiles_m >> 3, xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int cb = li / (4 * R), rem = li - cb * 4 * R;
    const int cw = min(4, tiles_n - 4 * cb);
    tm = xcd * R + rem / cw;
    tn = 4 * cb + rem % cw;
  } else {
    // any other row-block count (the row ranges of a split launch, below): tiles numbered panel-major
    // (4 column tiles x all row blocks per panel), XCD c takes the numbers [c n/8, (c+1) n/8) -- the 32
    // workgroups an XCD runs together are 8 row blocks x 4 column tiles
    const int per_xcd = (n_tiles + 7) >> 3;
    const int li = blockIdx.x >> 3, tile = (blockIdx.x & 7) * per_xcd + li;
    if (li >= per_xcd || tile >= n_tiles) return;
    const int cb = tile / (4 * tiles_m), rem = tile - cb * 4 * tiles_m;
    const int cw = min(4, tiles_n - 4 * cb);
    tm = rem / cw;
    tn = 4 * cb + rem % cw;
  }

  // staging map: float4 f = tid + 512 i = (row = f >> 2, kq = f & 3); rows clamped at the edge
  const float* ga[NA];
  int sdst_a[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = tid + G_THREADS * i, row = f >> 2, kq = f & 3;
    ga[i] = A + (int64_t)min(tm * TM + row, M - 1) * K + 4 * kq;
    sdst_a[i] = row * G_ROWB + kq * 8;
  }
  const int brow = tid >> 2, bkq = tid & 3;
  const float* gb = B + (int64_t)min(tn * GT + brow, N - 1) * K + 4 * bkq;
  // swizzle: 16-byte half (bkq >> 1) of row r goes to half ^ ((r >> 2) & 1)
  const int sdst_b = OB_A + brow * ROWB_B +
                     (BSW ? ((((bkq >> 1) ^ ((brow >> 2) & 1)) << 4) | ((bkq & 1) << 3)) : bkq * 8);
  // two register sets: the rows of step s are requested two steps before they are split
  f32x4 ra0[NA], ra1[NA], rb0, rb1;
#define RLPYT_G_FETCH(ra, rb, k0_)                                                             \
  {                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NA; ++i)                                             \
      ra[i] = *reinterpret_cast<const f32x4*>(ga[i] + (k0_));                                  \
    rb = *reinterpret_cast<const f32x4*>(gb + (k0_));                                          \
  }
#define RLPYT_G_STAGE(ra, rb, st_)                                                             \
  {                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                           \
      uint32_t p_[3][2];                                                                       \
      split3_rn(ra[i][0], ra[i][1], p_[0][0], p_[1][0], p_[2][0]);                             \
      split3_rn(ra[i][2], ra[i][3], p_[0][1], p_[1][1], p_[2][1]);                             \
      uint8_t* d_ = lds + (st_) * SB + sdst_a[i];                                              \
      _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_)                                         \
        *reinterpret_cast<uint2*>(d_ + s_ * PB_A) = uint2{p_[s_][0], p_[s_][1]};               \
    }                                                                                          \
    uint32_t q_[3][2];                                                                         \
    split3_rn(rb[0], rb[1], q_[0][0], q_[1][0], q_[2][0]);                                     \
    split3_rn(rb[2], rb[3], q_[0][1], q_[1][1], q_[2][1]);                                     \
    uint8_t* e_ = lds + (st_) * SB + sdst_b;                                                   \
    _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_)                                           \
      *reinterpret_cast<uint2*>(e_ + s_ * PB_B) = uint2{q_[s_][0], q_[s_][1]};                 \
  }
  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // operand addresses of this lane: row (column) l & 31 of the MFMA tile, K half l >> 5
  const int a_off = (wm * 64 + (lane & 31)) * G_ROWB + (lane >> 5) * 16;
  int b_off[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = wn * TNW + 32 * j + (lane & 31);
    b_off[j] = OB_A + col * ROWB_B + (BSW ? (((lane >> 5) ^ ((col >> 2) & 1)) << 4) : (lane >> 5) * 16);
  }
  // fragments of one K-step: 3 pieces x (2 row tiles of A + NJ column tiles of B)
#define RLPYT_G_FRAGS(af_, bf_, st_)                                                           \
  _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                             \
      bf_[j][s] = *reinterpret_cast<const uint4*>(lds + (st_) * SB + b_off[j] + s * PB_B);     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                              \
      af_[i][s] = *reinterpret_cast<const uint4*>(lds + (st_) * SB + a_off +                   \
                                                  i * 32 * G_ROWB + s * PB_A);                 \
  }
  // six products per tile, smallest first
#define RLPYT_G_TERM(af_, bf_, sa_, sb_)                                                       \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                               \
    acc[i][j] = mfma32_bf16(af_[i][sa_], bf_[j][sb_], acc[i][j]);
#define RLPYT_G_MMA(af_, bf_)                                                                  \
  RLPYT_G_TERM(af_, bf_, 2, 0)                                                                 \
  RLPYT_G_TERM(af_, bf_, 0, 2)                                                                 \
  RLPYT_G_TERM(af_, bf_, 1, 1)                                                                 \
  RLPYT_G_TERM(af_, bf_, 1, 0)                                                                 \
  RLPYT_G_TERM(af_, bf_, 0, 1)                                                                 \
  RLPYT_G_TERM(af_, bf_, 0, 0)

  // THREE LDS stages: during step s the fragments of step s + 1 are read (its stage was completed
  // at this step's barrier) while the MFMAs run on the fragments read during step s - 1, and the
  // registers of step s + 2 are split into the third stage.  With two stages the fragment reads of
  // all 8 waves started together right after every barrier and the matrix pipe waited for the LDS
  // queue to drain (~40 % of every step, measured 190-200 us for the trunk shapes).
  const int nk = K / G_BK;
  uint4 af0[2][3], bf0[NJ][3], af1[2][3], bf1[NJ][3];
  RLPYT_G_FETCH(ra0, rb0, 0)
  if (nk > 1) RLPYT_G_FETCH(ra1, rb1, G_BK)
  RLPYT_G_STAGE(ra0, rb0, 0)
  if (nk > 2) RLPYT_G_FETCH(ra0, rb0, 2 * G_BK)
  RLPYT_G_STAGE(ra1, rb1, 1)           // (nk == 1: stale registers into a stage nobody reads)
  if (nk > 3) RLPYT_G_FETCH(ra1, rb1, 3 * G_BK)
  __syncthreads();
  RLPYT_G_FRAGS(af0, bf0, 0)
  int st_next = 1, st_write = 2;     // stage of step ks + 1 / of step ks + 2
  // one step: prefetch the next fragments, MFMAs on the current ones with the split of step
  // ks + 2 in their gaps, then request step ks + 4 into the registers just consumed
  constexpr int NMMA = 12 * NJ;                          // MFMAs per step
  constexpr int NV = (22 * (NA + 1) + NMMA - 1) / NMMA;  // split VALU per MFMA gap: 4 / 3
#define RLPYT_G_STEP(afc_, bfc_, afn_, bfn_, ra, rb, ks_)                                      \
  {                                                                                            \
    RL_T(3)                                                                                    \
    __syncthreads();   /* stage of step ks + 1 complete; stage of step ks + 2 free */          \
    RL_T(0)                                                                                    \
    RLPYT_G_FRAGS(afn_, bfn_, st_next)   /* (past the last step: stale, unused) */            \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    RL_T(1)                                                                                    \
    RLPYT_G_MMA(afc_, bfc_)                                                                    \
    RLPYT_G_STAGE(ra, rb, st_write)      /* (past the last step: into a stage nobody reads) */ \
    /* the split VALU goes BETWEEN the MFMAs, NV per gap: issued as a block it runs while   */ \
    /* the matrix pipe idles (both waves of a SIMD are in the same phase)                     */ \
    _Pragma("unroll") for (int g_ = 0; g_ < NMMA; ++g_) {                                      \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                      \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    RL_T(2)                                                                                    \
    if ((ks_) + 4 < nk) RLPYT_G_FETCH(ra, rb, ((ks_) + 4) * G_BK)                              \
    st_next = st_next == 2 ? 0 : st_next + 1;                                                  \
    st_write = st_write == 2 ? 0 : st_write + 1;                                               \
  }
  int ks = 0;
  RL_T0()
#pragma unroll 1
  for (; ks + 1 < nk; ks += 2) {
    RLPYT_G_STEP(af0, bf0, af1, bf1, ra0, rb0, ks)
    RLPYT_G_STEP(af1, bf1, af0, bf0, ra1, rb1, ks + 1)
  }
  if (ks < nk) RLPYT_G_STEP(af0, bf0, af1, bf1, ra0, rb0, ks)
  RL_TOUT()
#undef RLPYT_G_STEP
#undef RLPYT_G_MMA
#undef RLPYT_G_TERM
#undef RLPYT_G_FRAGS
#undef RLPYT_G_STAGE
#undef RLPYT_G_FETCH
  // D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31] of tile (i, j)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = tn * GT + wn * TNW + 32 * j + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tm * TM + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) C[(int64_t)row * N + col] = acc[i][j][r];
      }
    }
}

// Row blocks of 256 for the first launch of rlpyt_gemm_nt_f32's tall shapes (0: all rows as 128-row
// tiles; ceil(M / 256): one launch, as before round 6).  Cost in rounds of one tile per CU; a round of
// 128-row tiles is priced at 0.55 of a 256-row round (measured: scripts/gemm_bench.py --nt-split).
// RLPYT_GEMM_NT_R256 overrides (A/B runs).
int gemm_nt_plan(int64_t M, int tiles_n) {
  static int n_cu = 0;
  if (n_cu == 0) {
    hipDeviceProp_t p;
    int dev = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess &&
            p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  const int all = (int)ceil_div(M, 256);
  if (const char* e = getenv("RLPYT_GEMM_NT_R256")) {
    const int r = atoi(e);
    return r < 0 ? all : std::min(r, all);
  }
  int best = all;
  double best_cost = 1e30;
  for (int r = all; r >= 0; --r) {
    const int64_t rest = std::max<int64_t>(0, M - (int64_t)r * 256);
    const double cost = (double)ceil_div((int64_t)r * tiles_n, n_cu) +
                        0.55 * (double)ceil_div(ceil_div(rest, 128) * tiles_n, n_cu);
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = r;
    }
  }
  return best;
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

#ifdef RLPYT_TIMING
extern "C" int rlpyt_debug_timing_read_gemm(float* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timing_gemm), (size_t)n * sizeof(float));
}
#endif

extern "C" int rlpyt_gemm_nt_f32(const float* a, const float* b, float* c, int64_t M, int64_t N,
                                 int64_t K, rlpyt_stream_t stream) {
  RL_CHECK_ARG(a && b && c, RLPYT_EINVAL, "rlpyt_gemm_nt_f32: null pointer");
  RL_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 32 == 0 && M < (1 << 30) && N < (1 << 30) &&
                   K < (1 << 30),
               RLPYT_ESHAPE, "rlpyt_gemm_nt_f32: need M, N > 0 and K a positive multiple of 32 "
                             "(M=%ld N=%ld K=%ld)", (long)M, (long)N, (long)K);
  RL_CHECK_ARG((reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0,
               RLPYT_ESHAPE, "rlpyt_gemm_nt_f32: a / b must be 16-byte aligned");
  const int tiles_n = (int)ceil_div(N, GT);
  hipStream_t s = (hipStream_t)stream;
  // 256-row tiles when they still fill the chip twice over (the trunk's input gradient: 32 x 27 tiles).
  // One workgroup per CU (LDS), so the launch runs in whole rounds of n_cu tiles: 864 tiles on 256 CUs
  // are 4 rounds, the last one 3/8 full.  The rows are therefore cut in two launches where that is
  // shorter by the round count: the first `r256` row blocks as 256-row tiles, the rows behind them as
  // 128-row tiles (a round of those costs ~0.55 of a 256-row round: gemm_nt_plan).
  int r256 = 0;
  if (ceil_div(M, 256) * tiles_n >= 512) r256 = gemm_nt_plan(M, tiles_n);
  if (r256 > 0) {
    const int64_t m256 = std::min<int64_t>(M, (int64_t)r256 * 256);
    const int grid = 8 * ((r256 * tiles_n + 7) / 8);     // whole rounds over the 8 XCDs
    RL_LAUNCH((gemm_nt_x6_kernel<256>), dim3(grid), dim3(G_THREADS), 0, s, a, b, c, (int)m256, (int)N,
              (int)K, r256, tiles_n);
    RL_LAUNCH_CHECK();
    a += m256 * K;
    c += m256 * N;
    M -= m256;
  }
  if (M > 0) {
    const int tiles_m = (int)ceil_div(M, GT);
    const int grid = 8 * ((tiles_m * tiles_n + 7) / 8);
    RL_LAUNCH((gemm_nt_x6_kernel<128>), dim3(grid), dim3(G_THREADS), 0, s, a, b, c, (int)M, (int)N,
              (int)K, tiles_m, tiles_n);
  }
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
