// Weight gradient of the PPO update's trunk Linear(3456, 512) (rlpyt/models/mlp.py:24-31 under
// autograd) on the bf16 matrix pipe ("bf16x6": both operands split into three bf16 pieces on their
// way HBM -> LDS, six products of order <= 2 accumulated in f32; dropped terms <= 2^-24 |ab|, see
// split_bf16.h / gemm.hip):
//   TN  C[M,N] = A[K,M]^T B[K,N]    weight gradient    g^T x         (contraction over the batch)
// TN contracts over the batch (K = 8192) and has only 4 x 27 output tiles: K is cut into 8 chunks,
// chunk c <-> XCD c (each XCD streams ITS rows of x and g once through its own L2; the 4 row tiles
// that share an x panel are adjacent work units), every unit writes a partial tile and
// gemm_reduce_slots_kernel sums the slots in a fixed order (deterministic, no atomics).  The
// tiles left over after whole rounds of 32 CUs are cut into `sub` K-parts so that the last round
// is short instead of sparse.
// (Rounds 3-4 also carried a producer / consumer-wave body for the NT / NN layouts here; it measured
// slower than the lock-step kernels -- 196-233 vs 156-163 us forward, 228-240 vs 195-212 us input
// gradient, profiles/r3_gemm_* -- and was removed in round 5.)
#include <stdlib.h>

#include <algorithm>

#include "split_bf16.h"

namespace rlpyt {
namespace {

constexpr int PT = 128;                    // tile edge of C
constexpr int P_THREADS = 512;
constexpr int P_BK = 32;                   // K granule of the plan (PpShape::nk counts K / 32)

struct PpShape {
  int M, N;                  // C is [M, N]
  int lda, ldb;              // leading dimensions of A and B (elements)
  int tiles_m, tiles_n;
  int nk;                    // K / 32
  int S;                     // K chunks: 1, or 8 (chunk <-> XCD, partial tiles)
  int full, sub;             // S == 8: units [0, full) of an XCD are whole chunks, the tiles behind
                             // them are cut into `sub` K-parts
};

// ---------------------------------------------------------------------------------------------
// TN in lock step (the structure of gemm_nt_x6_kernel<128>, gemm.hip: all 8 waves stage AND multiply,
// wave tile 64 x 32, K-step 16, three LDS stages, one barrier per step) with the transposition moved
// from the staging writes to the fragment READS: both operands of g^T x are K-major in memory
// (A[k][m], B[k][n]), so a thread's float4 -- 4 consecutive m (n) of one k -- is split and written
// as it comes, 8 bytes per piece, into a K-major LDS tile [16 k][128 columns], and the 8 consecutive
// K a lane needs of its column come from two ds_read_b64_tr_b16 per piece (split_bf16.h).  (The
// producer / consumer kernel of round 3 paid the transposition in its producers -- 4 x 4 register
// blocks, four LDS rows per thread: 212 us + 11 us for the trunk's weight gradient against 145 us for
// the same MACs in the forward layout.)  Rows are 320 B apart (256 + a pad that was swept: unpadded rows put
// the four K rows a 16-lane group of a transpose read touches on the same banks, +36 us).
// Work units, K chunks <-> XCDs and the partial-tile slots: tn_plan below.
constexpr int T_BK = 16;
#ifndef TX_PAD
#define TX_PAD 64     // (swept 0 .. 160 B: 0 / 16 -> 226-230 us, 32 -> 194, 64 -> 190, 48 / 80 .. 160 -> 205-210;
#endif                //  two workgroups per CU at 128 VGPRs: 260 -- profiles/r3_tn*_sweep.log)
#ifndef TX_OCC
#define TX_OCC 2
#endif
constexpr int T_ROWB = 2 * PT + TX_PAD;    // bytes per (piece, k) row: 128 bf16 + pad
constexpr int T_PB = T_BK * T_ROWB;        // 4,608
constexpr int T_OB = 3 * T_PB;             // 13,824 per operand
constexpr int T_SB = 2 * T_OB;             // 27,648 per stage

__global__ __launch_bounds__(P_THREADS, TX_OCC) void gemm_tn_x6_kernel(const float* __restrict__ A,
                                                               const float* __restrict__ B,
                                                               float* __restrict__ C, const PpShape sh) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[3 * T_SB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;          // wave tile: rows 64 wm .. + 63, columns 32 wn ..
  int tm, tn, ks0, ks1, slot = 0;
  {
    const int n_tiles = sh.tiles_m * sh.tiles_n;
    if (sh.S == 1) {
      const int per_xcd = (n_tiles + 7) >> 3;
      const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
      if (tile >= n_tiles || (int)(blockIdx.x >> 3) >= per_xcd) return;
      tn = tile / sh.tiles_m;
      tm = tile - tn * sh.tiles_m;
      ks0 = 0;
      ks1 = sh.nk;
    } else {
      const int c = blockIdx.x & 7, li = blockIdx.x >> 3;
      int t, part, nparts;
      if (li < sh.full) {
        t = li; part = 0; nparts = 1;
      } else {
        const int v = li - sh.full;
        t = sh.full + v / sh.sub; part = v - (v / sh.sub) * sh.sub; nparts = sh.sub;
      }
      if (t >= n_tiles) return;
      tn = t / sh.tiles_m;                    // the row tiles that share a B panel are adjacent
      tm = t - tn * sh.tiles_m;
      const int c0 = (int)((int64_t)c * sh.nk / 8), c1 = (int)((int64_t)(c + 1) * sh.nk / 8);
      ks0 = c0 + (int)((int64_t)part * (c1 - c0) / nparts);
      ks1 = c0 + (int)((int64_t)(part + 1) * (c1 - c0) / nparts);
      slot = c * sh.sub + part;
    }
  }
  const int nk = 2 * (ks1 - ks0);                   // K-16 steps (PpShape counts K-32 steps)
  float* Cout = C + (int64_t)slot * sh.M * sh.N;

  // staging: float4 tid = (k = tid >> 5, columns 4 (tid & 31) .. + 3) of each operand's
  // [16 k][128 columns] step tile (dim % 4 == 0: a float4 is all in or all out; clamped at the edge)
  const int sk = tid >> 5, sq = tid & 31;
  const float* ga = A + ((int64_t)ks0 * P_BK + sk) * sh.lda + min(tm * PT + 4 * sq, sh.M - 4);
  const float* gb = B + ((int64_t)ks0 * P_BK + sk) * sh.ldb + min(tn * PT + 4 * sq, sh.N - 4);
  const int64_t stepa = (int64_t)T_BK * sh.lda, stepb = (int64_t)T_BK * sh.ldb;
  const int sdst = sk * T_ROWB + sq * 8;
  f32x4 ra0, rb0, ra1, rb1;
#define TX_FETCH(ra, rb, step_)                                                     \
  {                                                                                 \
    ra = *reinterpret_cast<const f32x4*>(ga + (step_) * stepa);                     \
    rb = *reinterpret_cast<const f32x4*>(gb + (step_) * stepb);                     \
  }
#define TX_STAGE(ra, rb, st_)                                                       \
  {                                                                                 \
    uint32_t p_[3][2], q_[3][2];                                                    \
    split3_rn(ra[0], ra[1], p_[0][0], p_[1][0], p_[2][0]);                          \
    split3_rn(ra[2], ra[3], p_[0][1], p_[1][1], p_[2][1]);                          \
    split3_rn(rb[0], rb[1], q_[0][0], q_[1][0], q_[2][0]);                          \
    split3_rn(rb[2], rb[3], q_[0][1], q_[1][1], q_[2][1]);                          \
    uint8_t* d_ = lds + (st_) * T_SB + sdst;                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                              \
      *reinterpret_cast<uint2*>(d_ + s_ * T_PB) = uint2{p_[s_][0], p_[s_][1]};      \
      *reinterpret_cast<uint2*>(d_ + T_OB + s_ * T_PB) = uint2{q_[s_][0], q_[s_][1]}; \
    }                                                                               \
  }
  // fragment addresses: lane = (s = lane & 15, column half (lane >> 4) & 1, K half h = lane >> 5);
  // transpose read r (0 | 1) of a 16-lane group covers K 8 h + 4 r .. + 3 of 16 columns
  const int fs = lane & 15, fc = 16 * ((lane >> 4) & 1) + 4 * (fs & 3), fk = 8 * (lane >> 5) + (fs >> 2);
  int a_off[2][2], b_off[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) a_off[i][r] = (fk + 4 * r) * T_ROWB + (wm * 64 + 32 * i + fc) * 2;
    b_off[r] = T_OB + (fk + 4 * r) * T_ROWB + (wn * 32 + fc) * 2;
  }
#define TX_FRAGS(af_, bf_, st_)                                                     \
  _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                   \
    const uint8_t* b_ = lds + (st_) * T_SB + s * T_PB;                              \
    const uint2 x0_ = lds_tr16(b_ + b_off[0]), x1_ = lds_tr16(b_ + b_off[1]);       \
    bf_[s] = uint4{x0_.x, x0_.y, x1_.x, x1_.y};                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                 \
      const uint2 y0_ = lds_tr16(b_ + a_off[i][0]), y1_ = lds_tr16(b_ + a_off[i][1]); \
      af_[i][s] = uint4{y0_.x, y0_.y, y1_.x, y1_.y};                                \
    }                                                                               \
  }
#define TX_TERM(af_, bf_, sa_, sb_)                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[i] = mfma32_bf16(af_[i][sa_], bf_[sb_], acc[i]);
#define TX_MMA(af_, bf_)                                                            \
  TX_TERM(af_, bf_, 2, 0) TX_TERM(af_, bf_, 0, 2) TX_TERM(af_, bf_, 1, 1)           \
  TX_TERM(af_, bf_, 1, 0) TX_TERM(af_, bf_, 0, 1) TX_TERM(af_, bf_, 0, 0)
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  uint4 af0[2][3], bf0[3], af1[2][3], bf1[3];
  TX_FETCH(ra0, rb0, 0)
  if (nk > 1) TX_FETCH(ra1, rb1, 1)
  TX_STAGE(ra0, rb0, 0)
  if (nk > 2) TX_FETCH(ra0, rb0, 2)
  TX_STAGE(ra1, rb1, 1)               // (nk == 1: stale registers into a stage nobody reads)
  if (nk > 3) TX_FETCH(ra1, rb1, 3)
  __syncthreads();
  TX_FRAGS(af0, bf0, 0)
  int st_next = 1, st_write = 2;      // stage of step ks + 1 / of step ks + 2
  // one step (as gemm_nt_x6_kernel): fragments of the next step, MFMAs on the current ones with
  // the split of step ks + 2 in their gaps, then the request of step ks + 4
#define TX_STEP(afc_, bfc_, afn_, bfn_, ra, rb, ks_)                                \
  {                                                                                 \
    __syncthreads();                                                                \
    TX_FRAGS(afn_, bfn_, st_next)                                                   \
    __builtin_amdgcn_sched_barrier(0);                                              \
    TX_MMA(afc_, bfc_)                                                              \
    TX_STAGE(ra, rb, st_write)                                                      \
    _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                            \
    }                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                              \
    if ((ks_) + 4 < nk) TX_FETCH(ra, rb, (ks_) + 4)                                 \
    st_next = st_next == 2 ? 0 : st_next + 1;                                       \
    st_write = st_write == 2 ? 0 : st_write + 1;                                    \
  }
  int ks = 0;
#pragma unroll 1
  for (; ks + 1 < nk; ks += 2) {
    TX_STEP(af0, bf0, af1, bf1, ra0, rb0, ks)
    TX_STEP(af1, bf1, af0, bf0, ra1, rb1, ks + 1)
  }
  if (ks < nk) TX_STEP(af0, bf0, af1, bf1, ra0, rb0, ks)
#undef TX_STEP
#undef TX_MMA
#undef TX_TERM
#undef TX_FRAGS
#undef TX_STAGE
#undef TX_FETCH
  // D[row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col = lane & 31] of tile i
  const int col = tn * PT + wn * 32 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tm * PT + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < sh.M && col < sh.N) Cout[(int64_t)row * sh.N + col] = acc[i][r];
    }
}

// out[e] = sum over the K chunks c = 0..7 (and the K parts of the left-over tiles) of the partial
// tiles, in a fixed order; float4 per thread (N % 4 == 0: a float4 stays inside one tile)
__global__ __launch_bounds__(256) void gemm_reduce_slots_kernel(const float* __restrict__ partial,
                                                                float* __restrict__ out, PpShape sh) {
  const int64_t MN = (int64_t)sh.M * sh.N;
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= MN) return;
  const int row = (int)(e / sh.N), col = (int)(e - (int64_t)row * sh.N);
  const int t = (col / PT) * sh.tiles_m + row / PT;
  const int nparts = t >= sh.full ? sh.sub : 1;
  f32x4 v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
    v[c] = *reinterpret_cast<const f32x4*>(partial + (int64_t)(c * sh.sub) * MN + e);
  for (int p = 1; p < nparts; ++p)
#pragma unroll
    for (int c = 0; c < 8; ++c)
      v[c] += *reinterpret_cast<const f32x4*>(partial + (int64_t)(c * sh.sub + p) * MN + e);
  const f32x4 r = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  *reinterpret_cast<f32x4*>(out + e) = r;
}

int cus_per_xcd() {
  static int n = 0;
  if (n == 0) {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess &&
        p.multiProcessorCount >= 8)
      n = p.multiProcessorCount / 8;
    else
      n = 32;
  }
  return n;
}

// split plan of the TN GEMM: whole rounds of one unit per CU first, the left-over tiles in K parts
void tn_plan(PpShape& sh) {
  const int n_tiles = sh.tiles_m * sh.tiles_n, cu = cus_per_xcd();
  sh.S = 8;
  sh.full = (n_tiles / cu) * cu;
  const int r = n_tiles - sh.full, chunk = sh.nk / 8;
  sh.sub = 1;
  if (r > 0) sh.sub = std::max(1, std::min(std::min(cu / r, 4), chunk / 4));
  if (sh.sub == 1) sh.full = n_tiles;
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;


static int pp_check(const char* fn, const float* a, const float* b, const float* c, int64_t M,
                    int64_t N, int64_t K) {
  RL_CHECK_ARG(a && b && c, RLPYT_EINVAL, "%s: null pointer", fn);
  RL_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 32 == 0 && M < (1 << 30) && N < (1 << 30) &&
                   K < (1 << 30),
               RLPYT_ESHAPE, "%s: need M, N > 0 and K a positive multiple of 32 (M=%ld N=%ld K=%ld)",
               fn, (long)M, (long)N, (long)K);
  RL_CHECK_ARG(RL_ALIGNED16(a) && RL_ALIGNED16(b) && RL_ALIGNED16(c), RLPYT_ESHAPE,
               "%s: a / b / c must be 16-byte aligned", fn);
  return RLPYT_OK;
}

static PpShape pp_shape(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb) {
  PpShape sh;
  sh.M = (int)M; sh.N = (int)N; sh.lda = (int)lda; sh.ldb = (int)ldb;
  sh.tiles_m = (int)ceil_div(M, PT); sh.tiles_n = (int)ceil_div(N, PT);
  sh.nk = (int)(K / P_BK);
  sh.S = 1; sh.full = sh.tiles_m * sh.tiles_n; sh.sub = 1;
  return sh;
}

static int pp_grid(const PpShape& sh) { return 8 * ((sh.tiles_m * sh.tiles_n + 7) / 8); }



extern "C" int64_t rlpyt_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32) return 0;
  PpShape sh = pp_shape(M, N, K, M, N);
  if (sh.nk < 64) return 0;                  // no K split below 2048 rows
  tn_plan(sh);
  return (int64_t)8 * sh.sub * M * N * (int64_t)sizeof(float);
}

extern "C" int rlpyt_gemm_tn_f32(const float* a, const float* b, float* c, int64_t M, int64_t N,
                                 int64_t K, void* workspace, rlpyt_stream_t stream) {
  if (int e = pp_check("rlpyt_gemm_tn_f32", a, b, c, M, N, K)) return e;
  RL_CHECK_ARG(M % 4 == 0 && N % 4 == 0, RLPYT_ESHAPE,
               "rlpyt_gemm_tn_f32: M and N must be multiples of 4 (M=%ld N=%ld)", (long)M, (long)N);
  PpShape sh = pp_shape(M, N, K, M, N);
  hipStream_t s = (hipStream_t)stream;
  if (sh.nk < 64) {            // short contraction: one unit per tile, straight into c
    RL_LAUNCH(gemm_tn_x6_kernel, dim3(pp_grid(sh)), dim3(P_THREADS), 0, s, a, b, c, sh);
    RL_LAUNCH_CHECK();
    return RLPYT_OK;
  }
  RL_CHECK_ARG(workspace && RL_ALIGNED16(workspace), RLPYT_EINVAL,
               "rlpyt_gemm_tn_f32: K >= 2048 needs the workspace of rlpyt_gemm_tn_workspace_bytes");
  tn_plan(sh);
  const int n_tiles = sh.tiles_m * sh.tiles_n;
  const int units = sh.full + (n_tiles - sh.full) * sh.sub;
  float* ws = static_cast<float*>(workspace);
  RL_LAUNCH(gemm_tn_x6_kernel, dim3(8 * units), dim3(P_THREADS), 0, s, a, b, ws, sh);
  RL_LAUNCH_CHECK();
  const int64_t n4 = M * N / 4;
  RL_LAUNCH(gemm_reduce_slots_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, s, ws, c, sh);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
