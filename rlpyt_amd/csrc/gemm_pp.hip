// f32 GEMMs of the PPO update's trunk Linear(3456, 512) (rlpyt/models/mlp.py:24-31 forward and
// backward) on the bf16 matrix pipe ("bf16x6": both operands split into three bf16 pieces on their
// way HBM -> LDS, six products of order <= 2 accumulated in f32; dropped terms <= 2^-24 |ab|, see
// split_bf16.h / gemm.hip) for the operand layouts
//   NT  C[M,N] = A[M,K]  B[N,K]^T   forward            x W^T
//   NN  C[M,N] = A[M,K]  B[K,N]     input gradient     g W           (no transposed copy of W)
//   TN  C[M,N] = A[K,M]^T B[K,N]    weight gradient    g^T x         (contraction over the batch)
// TN -- the product's weight-gradient kernel -- is gemm_tn_x6_kernel further down (lock step,
// transposing LDS reads).  NT and NN share ONE body (opt-in A/B variants of the lock-step NT kernel
// of gemm.hip, ops.GEMM_NT_PINGPONG / ops.GEMM_DGRAD_NN; the TN layout ran through it too until the
// lock-step kernel replaced it: 212 -> 184 us) with SPECIALISED waves: of the 8 waves of a workgroup (tile 128 x 128 of C), waves 4-7 are
// PRODUCERS -- fetch both operands (8 float4 per thread and K-32 step, requested two steps ahead),
// split them into bf16 pieces, write an LDS stage -- and waves 0-3 are CONSUMERS, one per SIMD, wave
// tile 64 x 64: 24 fragment reads and 48 MFMAs per step.  One barrier per K-32 step.
//
// What shaped it (per-phase cycle counters and phase-elimination builds, profiles/r3_gemm_*.log):
//  * v_mfma issue is asynchronous on gfx950: 24 dependent MFMAs issue in ~80 cycles and drain from a
//    queue at 32 cycles each, so ONE wave per SIMD keeps the matrix pipe busy as long as its next
//    fragments are in registers before the queue runs dry.  A ping-pong version (wave halves
//    alternating between an MFMA segment and a load segment, two barriers per step) overlapped
//    every PAIR of its phases and still took 3300-3800 cycles per step against 1540 of matrix-pipe
//    time: whatever the loading half waited for held the computing half at the barrier.
//  * the consumer therefore never reads fragments right behind a barrier: the LDS ring has THREE
//    stages, and the fragments of step v + 1 are read slice by slice BEHIND the MFMAs of step v
//    (slice 0 of v + 1 into the registers slice 0 of v just left, while slice 1 of v runs).  With
//    two stages the reads sat between the barrier and the first MFMA: 2090 cycles per step.
//  * three stages fit the 160 KB of LDS only unpadded: rows are 64 B (32 bf16 along K), and the
//    16-byte chunk c of row r lives at chunk c ^ ((r >> 2) & 3): the 16-lane groups of a
//    ds_read_b128 (rows r, K half h) then cover all 64 banks, the 8-byte staging writes of two
//    adjacent rows all 32.  3 x {A, B} x 3 pieces x 128 rows x 64 B = 147,456 B.
//  * 64 x 64 wave tiles halve the fragment traffic per MFMA (0.5 KB instead of 0.75 KB).
//
// Staging of an operand X whose K axis is contiguous ("KC": X[r][k]): a thread moves 4 float4 =
// 4 k of rows r, r+32, r+64, r+96 (8 lanes cover 128 contiguous bytes of a row) and writes
// 3 x 4 ds_write_b64.  K strided ("KS": X[k][r], the transposing layouts of NN / TN): a thread
// owns a 4 (k) x 4 (r) block -- four float4 loads along r from four consecutive k rows -- packs
// (k, k+1) pairs per r and writes, per piece, one ds_write_b64 (4 k of one r) to each of its four
// LDS rows (lanes: k group = lane & 7, r group = lane >> 3).
//
// TN contracts over the batch (K = 8192) and has only 4 x 27 output tiles: K is cut into 8 chunks,
// chunk c <-> XCD c (each XCD streams ITS rows of x and g once through its own L2; the 4 row tiles
// that share an x panel are adjacent work units), every unit writes a partial tile and
// gemm_reduce_slots_kernel sums the slots in a fixed order (deterministic, no atomics).  The
// tiles left over after whole rounds of 32 CUs are cut into `sub` K-parts so that the last round
// is short instead of sparse.
#include <stdlib.h>

#include <algorithm>

#include "split_bf16.h"

namespace rlpyt {
namespace {

constexpr int PT = 128;                    // tile edge of C
constexpr int P_THREADS = 512;
constexpr int P_BK = 32;                   // K per step (two MFMA slices of 16)
constexpr int P_ROWB = 2 * P_BK;           // 64 B per (piece, row), chunks swizzled (no pad)
constexpr int P_PB = PT * P_ROWB;          // bytes per piece of one operand
constexpr int P_OB = 3 * P_PB;             // bytes per operand of one stage
constexpr int P_SB = 2 * P_OB;             // bytes per stage: 49,152
constexpr int P_NST = 3;                   // stages of the LDS ring
constexpr int P_LDS = P_NST * P_SB;        // 147,456
constexpr int KC = 0, KS = 1;              // operand layouts: K contiguous / K strided

// byte offset of (row, 8-byte slot q = k / 4) inside a piece: chunk q >> 1 swizzled by the row
__device__ __forceinline__ int pp_slot(int row, int q) {
  return row * P_ROWB + ((((q >> 1) ^ (row >> 2)) & 3) << 4) + ((q & 1) << 3);
}

// PP_SKIP (debug builds only; results are then wrong): bit 1 no split / LDS writes, 2 no global loads
// in the loop, 3 no MFMAs -- what a phase costs is what leaving it out saves
#ifndef PP_SKIP
#define PP_SKIP 0
#endif
// Debug build (-DRLPYT_TIMING): per-wave cycle totals of the loop phases (scripts/debug/
// gemm_pp_timing.py); compiled out of the product.  slots: 0 producer stage + fetch, 1 its barrier
// wait, 2 consumer MFMA issue + fragment reads, 3 its barrier wait
#ifdef RLPYT_TIMING
__device__ float g_timing_gemm_pp[1024 * 8 * 4];
#define PP_T0() long long t_prev_ = clock64(), t_acc_[4] = {0, 0, 0, 0};
#define PP_T(k)                                  \
  {                                              \
    const long long t_now_ = clock64();          \
    t_acc_[k] += t_now_ - t_prev_;               \
    t_prev_ = t_now_;                            \
  }
#define PP_TOUT()                                                                         \
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) {                                     \
    float* dbg_ = g_timing_gemm_pp + ((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 4;  \
    for (int k = 0; k < 4; ++k) dbg_[k] = (float)t_acc_[k];                               \
  }
#else
#define PP_T0()
#define PP_T(k)
#define PP_TOUT()
#endif

typedef int i32x4_ __attribute__((ext_vector_type(4)));

struct PpShape {
  int M, N;                  // C is [M, N]
  int lda, ldb;              // leading dimensions of A and B (elements)
  int tiles_m, tiles_n;
  int nk;                    // K / 32
  int S;                     // K chunks: 1, or 8 (chunk <-> XCD, partial tiles)
  int full, sub;             // S == 8: units [0, full) of an XCD are whole chunks, the tiles behind
                             // them are cut into `sub` K-parts
  int rot;                   // > 0: unit li starts its K loop at step (li * rot) % nk and wraps
                             // (operands whose rows are a power of two apart -- g: 2 KB -- put the
                             // same K offset of every row on the same 2 of 16 L2 channels; units
                             // that walk K in step all camp on them)
};

// split the fetched rows into bf16 pieces and write them into LDS (operand base `dst`)
template <int L>
__device__ __forceinline__ void pp_stage(const f32x4 (&r)[4], uint8_t* dst, const int (&soff)[4]) {
  if constexpr (L == KC) {
    // r[i] = 4 consecutive k of row (row + 32 i): one 8-byte write per piece and row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t p[3][2];
      split3_rn(r[i][0], r[i][1], p[0][0], p[1][0], p[2][0]);
      split3_rn(r[i][2], r[i][3], p[0][1], p[1][1], p[2][1]);
#pragma unroll
      for (int s = 0; s < 3; ++s)
        *reinterpret_cast<uint2*>(dst + soff[i] + s * P_PB) = uint2{p[s][0], p[s][1]};
    }
  } else {
    // r[i][j] = element (k = 4 mg + i, row = 4 ng + j): transpose the 4 x 4 block while packing
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t p[3][2];
      split3_rn(r[0][j], r[1][j], p[0][0], p[1][0], p[2][0]);
      split3_rn(r[2][j], r[3][j], p[0][1], p[1][1], p[2][1]);
#pragma unroll
      for (int s = 0; s < 3; ++s)
        *reinterpret_cast<uint2*>(dst + soff[j] + s * P_PB) = uint2{p[s][0], p[s][1]};
    }
  }
}

// this producer thread's share of an operand: 4 global float4 per K-32 step, 4 LDS slots
template <int L>
__device__ __forceinline__ void pp_operand_map(const float* __restrict__ X, int ld, int dim, int r0,
                                               int ks0, int ht, const float* (&gp)[4], int (&soff)[4],
                                               int64_t& kstride) {
  if constexpr (L == KC) {
    const int row = ht >> 3, kq = ht & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gp[i] = X + (int64_t)min(r0 + row + 32 * i, dim - 1) * ld + (int64_t)ks0 * P_BK + 4 * kq;
      soff[i] = pp_slot(row + 32 * i, kq);
    }
    kstride = 1;
  } else {
    const int mg = ht & 7, ng = 8 * (ht >> 6) + ((ht >> 3) & 7);
    const int col = min(r0 + 4 * ng, dim - 4);   // dim % 4 == 0: blocks are all in or all out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gp[i] = X + ((int64_t)ks0 * P_BK + 4 * mg + i) * ld + col;
      soff[i] = pp_slot(4 * ng + i, mg);
    }
    kstride = ld;
  }
}

#define WS_BAR()                            \
  {                                         \
    __builtin_amdgcn_sched_barrier(0);      \
    __syncthreads();                        \
    __builtin_amdgcn_sched_barrier(0);      \
  }

// Producer waves (4-7).  Ring protocol: at the barrier that ends step v, stage (v + 2) % 3 holds
// step v + 2; during step v the producers write it while the consumers read stage (v + 1) % 3.
template <int LA, int LB>
__device__ __forceinline__ void ws_producer(const float* __restrict__ A, const float* __restrict__ B,
                                            const PpShape& sh, int tm, int tn, int ks0, int nk,
                                            int rot, uint8_t* lds) {
  const int ht = threadIdx.x & 255;
  const float *gpa[4], *gpb[4];
  int soa[4], sob[4];
  int64_t ksa, ksb;
  pp_operand_map<LA>(A, sh.lda, sh.M, tm * PT, ks0, ht, gpa, soa, ksa);
  pp_operand_map<LB>(B, sh.ldb, sh.N, tn * PT, ks0, ht, gpb, sob, ksb);
  f32x4 Ra[2][4], Rb[2][4];            // two fetch sets
  const int last = nk - 1;
#define WS_FETCH(set_, step_)                                                     \
  {                                                                               \
    int kk_ = min((step_), last) + rot;                                           \
    kk_ = kk_ >= nk ? kk_ - nk : kk_;                                             \
    const int64_t ko_ = (int64_t)kk_ * P_BK;                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                               \
      Ra[set_][i] = *reinterpret_cast<const f32x4*>(gpa[i] + ko_ * ksa);          \
      Rb[set_][i] = *reinterpret_cast<const f32x4*>(gpb[i] + ko_ * ksb);          \
    }                                                                             \
  }
#define WS_STAGE(set_, st_)                                                       \
  {                                                                               \
    pp_stage<LA>(Ra[set_], lds + (st_) * P_SB, soa);                              \
    pp_stage<LB>(Rb[set_], lds + (st_) * P_SB + P_OB, sob);                       \
  }
  // prologue: stages 0 and 1 <- steps 0 and 1; the sets then hold steps 2 and 3
  WS_FETCH(0, 0)
  WS_FETCH(1, 1)
  WS_STAGE(0, 0)
  WS_FETCH(0, 2)
  WS_STAGE(1, 1)
  WS_FETCH(1, 3)
  WS_BAR()
  // step v: set v & 1 (step v + 2) -> stage (v + 2) % 3, refilled with step v + 4
#define WS_PSTEP(p_, v_)                                                          \
  {                                                                               \
    if (!(PP_SKIP & 2)) WS_STAGE(p_, st)                                          \
    else { _Pragma("unroll") for (int i = 0; i < 4; ++i)                          \
             asm volatile("" ::"v"(Ra[p_][i]), "v"(Rb[p_][i])); }                 \
    if (!(PP_SKIP & 4)) WS_FETCH(p_, (v_) + 4)                                    \
    st = st == P_NST - 1 ? 0 : st + 1;                                            \
    PP_T(0)                                                                       \
    WS_BAR()                                                                      \
    PP_T(1)                                                                       \
  }
  int st = 2, v = 0;
  PP_T0()
#pragma unroll 1
  for (; v + 1 < nk; v += 2) {
    WS_PSTEP(0, v)
    WS_PSTEP(1, v + 1)
  }
  PP_TOUT()
  if (v < nk) WS_PSTEP(0, v)
#undef WS_PSTEP
#undef WS_STAGE
#undef WS_FETCH
}

// Consumer waves (0-3): wave tile 64 x 64 = 2 x 2 MFMA tiles.  a_off / b_off: this lane's row /
// column (lane & 31) of tile 0 and its swizzled 16-byte chunk for K slice 0 / 1 (K half lane >> 5).
__device__ __forceinline__ void ws_consumer(const uint8_t* lds, int nk, const int (&a_off)[2],
                                            const int (&b_off)[2], f32x16 (&acc)[2][2]) {
  uint4 fa[2][2][3], fb[2][2][3];      // [slice][tile][piece]
#define WS_FRAGS(sl_, base_)                                                      \
  _Pragma("unroll") for (int p = 0; p < 3; ++p)                                   \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                 \
    const uint8_t* b_ = (base_) + p * P_PB + t * 32 * P_ROWB;                     \
    fa[sl_][t][p] = *reinterpret_cast<const uint4*>(b_ + a_off[sl_]);             \
    fb[sl_][t][p] = *reinterpret_cast<const uint4*>(b_ + b_off[sl_]);             \
  }
#define WS_TERM(sl_, sa_, sb_)                                                    \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                   \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                   \
    acc[i][j] = mfma32_bf16(fa[sl_][i][sa_], fb[sl_][j][sb_], acc[i][j]);
  // six products per tile, smallest first
#define WS_MMA(sl_)                                                               \
  if (!(PP_SKIP & 8)) {                                                           \
    WS_TERM(sl_, 2, 0) WS_TERM(sl_, 0, 2) WS_TERM(sl_, 1, 1)                      \
    WS_TERM(sl_, 1, 0) WS_TERM(sl_, 0, 1) WS_TERM(sl_, 0, 0)                      \
  } else {                                                                        \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                 \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                 \
      asm volatile("" ::"v"(__builtin_bit_cast(i32x4_, fa[sl_][t][p])),           \
                   "v"(__builtin_bit_cast(i32x4_, fb[sl_][t][p])));               \
  }
  WS_BAR()                              // stages 0 and 1 ready
  WS_FRAGS(0, lds)
  WS_FRAGS(1, lds)
  int st = 1;                           // stage of step v + 1
  PP_T0()
#pragma unroll 1
  for (int v = 0; v < nk; ++v) {
    const uint8_t* nxt = lds + st * P_SB;
    WS_MMA(0)
    __builtin_amdgcn_sched_barrier(0);  // the reads of slice 0 go BEHIND its MFMAs, not before
    WS_FRAGS(0, nxt)                    // (past the last step: a stage nobody needs)
    __builtin_amdgcn_sched_barrier(0);
    WS_MMA(1)
    __builtin_amdgcn_sched_barrier(0);
    WS_FRAGS(1, nxt)
    st = st == P_NST - 1 ? 0 : st + 1;
    PP_T(2)
    WS_BAR()
    PP_T(3)
  }
  PP_TOUT()
#undef WS_MMA
#undef WS_TERM
#undef WS_FRAGS
}
#undef WS_BAR

template <int LA, int LB>
__device__ __forceinline__ void gemm_pp_body(const float* __restrict__ A, const float* __restrict__ B,
                                             float* __restrict__ C, const PpShape sh) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[P_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- work unit -> (tile, K range, output slot) -------------------------------------------
  int tm, tn, ks0, ks1, slot = 0;
  if (sh.S == 1) {
    const int n_tiles = sh.tiles_m * sh.tiles_n;
    if ((sh.tiles_m & 7) == 0) {
      // every XCD owns tiles_m / 8 row blocks and walks them in panels of 4 column tiles: the 32
      // workgroups running together on an XCD share few A row blocks and 4 B column tiles
      const int R = sh.tiles_m >> 3, xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
      const int cb = li / (4 * R), rem = li - cb * 4 * R;
      const int cw = min(4, sh.tiles_n - 4 * cb);
      tm = xcd * R + rem / cw;
      tn = 4 * cb + rem % cw;
    } else {
      const int per_xcd = (n_tiles + 7) >> 3;
      const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
      if (tile >= n_tiles || (int)(blockIdx.x >> 3) >= per_xcd) return;
      tm = tile / sh.tiles_n;
      tn = tile - tm * sh.tiles_n;
    }
    ks0 = 0;
    ks1 = sh.nk;
  } else {
    const int c = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int n_tiles = sh.tiles_m * sh.tiles_n;
    int t, part, nparts;
    if (li < sh.full) {
      t = li; part = 0; nparts = 1;
    } else {
      const int v = li - sh.full;
      t = sh.full + v / sh.sub; part = v - (v / sh.sub) * sh.sub; nparts = sh.sub;
    }
    if (t >= n_tiles) return;
    tn = t / sh.tiles_m;                      // the row tiles that share a B panel are adjacent
    tm = t - tn * sh.tiles_m;
    const int c0 = (int)((int64_t)c * sh.nk / 8), c1 = (int)((int64_t)(c + 1) * sh.nk / 8);
    ks0 = c0 + (int)((int64_t)part * (c1 - c0) / nparts);
    ks1 = c0 + (int)((int64_t)(part + 1) * (c1 - c0) / nparts);
    slot = c * sh.sub + part;
  }
  const int nk = ks1 - ks0;
  float* Cout = C + (int64_t)slot * sh.M * sh.N;
  const int rot = sh.rot > 0 ? (int)(((int64_t)(blockIdx.x >> 3) * sh.rot) % nk) : 0;

  if (wave >= 4) {       // (the branch is wave-uniform; both roles execute 1 + nk barriers)
    ws_producer<LA, LB>(A, B, sh, tm, tn, ks0, nk, rot, lds);
    return;
  }
  // consumers: 2 x 2 waves of 64 x 64
  const int cm = wave & 1, cn = wave >> 1;
  const int fr = lane & 31, fh = lane >> 5;
  int a_off[2], b_off[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    a_off[sl] = pp_slot(cm * 64 + fr, 2 * (2 * sl + fh));
    b_off[sl] = P_OB + pp_slot(cn * 64 + fr, 2 * (2 * sl + fh));
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  ws_consumer(lds, nk, a_off, b_off, acc);
  // D[row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col = lane & 31] of tile (i, j)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = tn * PT + cn * 64 + 32 * j + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tm * PT + cm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row < sh.M && col < sh.N) Cout[(int64_t)row * sh.N + col] = acc[i][j][r];
      }
    }
}

__global__ __launch_bounds__(P_THREADS) void gemm_nt_pp_kernel(const float* A, const float* B,
                                                               float* C, PpShape sh) {
  gemm_pp_body<KC, KC>(A, B, C, sh);
}
__global__ __launch_bounds__(P_THREADS) void gemm_nn_pp_kernel(const float* A, const float* B,
                                                               float* C, PpShape sh) {
  gemm_pp_body<KC, KS>(A, B, C, sh);
}

// ---------------------------------------------------------------------------------------------
// TN in lock step (the structure of gemm_nt_x6_kernel<128>, gemm.hip: all 8 waves stage AND multiply,
// wave tile 64 x 32, K-step 16, three LDS stages, one barrier per step) with the transposition moved
// from the staging writes to the fragment READS: both operands of g^T x are K-major in memory
// (A[k][m], B[k][n]), so a thread's float4 -- 4 consecutive m (n) of one k -- is split and written
// as it comes, 8 bytes per piece, into a K-major LDS tile [16 k][128 columns], and the 8 consecutive
// K a lane needs of its column come from two ds_read_b64_tr_b16 per piece (split_bf16.h).  The
// producer / consumer kernel above pays the transposition in its producers (4 x 4 register blocks,
// four LDS rows per thread): 212 us + 11 us for the trunk's weight gradient against 145 us for the
// same MACs in the forward layout.  Rows are 320 B apart (256 + a pad that was swept: unpadded rows put
// the four K rows a 16-lane group of a transpose read touches on the same banks, +36 us).
// Work units, K chunks <-> XCDs and the partial-tile slots are those of gemm_pp_body (tn_plan).
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

#ifdef RLPYT_TIMING
extern "C" int rlpyt_debug_timing_read_gemm_pp(float* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timing_gemm_pp), (size_t)n * sizeof(float));
}
#endif

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
  sh.S = 1; sh.full = sh.tiles_m * sh.tiles_n; sh.sub = 1; sh.rot = 0;
  return sh;
}

static int pp_grid(const PpShape& sh) { return 8 * ((sh.tiles_m * sh.tiles_n + 7) / 8); }

extern "C" int rlpyt_gemm_nt_pp_f32(const float* a, const float* b, float* c, int64_t M, int64_t N,
                                    int64_t K, rlpyt_stream_t stream) {
  if (int e = pp_check("rlpyt_gemm_nt_f32", a, b, c, M, N, K)) return e;
  const PpShape sh = pp_shape(M, N, K, K, K);
  RL_LAUNCH(gemm_nt_pp_kernel, dim3(pp_grid(sh)), dim3(P_THREADS), 0, (hipStream_t)stream, a, b,
            c, sh);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_gemm_nn_f32(const float* a, const float* b, float* c, int64_t M, int64_t N,
                                 int64_t K, rlpyt_stream_t stream) {
  if (int e = pp_check("rlpyt_gemm_nn_f32", a, b, c, M, N, K)) return e;
  RL_CHECK_ARG(N % 4 == 0, RLPYT_ESHAPE, "rlpyt_gemm_nn_f32: N must be a multiple of 4 (N=%ld)",
               (long)N);
  PpShape sh = pp_shape(M, N, K, K, N);
  if (const char* e = getenv("RLPYT_GEMM_ROT")) sh.rot = atoi(e);
  RL_LAUNCH(gemm_nn_pp_kernel, dim3(pp_grid(sh)), dim3(P_THREADS), 0, (hipStream_t)stream, a, b,
            c, sh);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

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
