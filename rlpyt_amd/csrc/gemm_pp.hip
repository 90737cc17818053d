// f32 GEMMs of the PPO update's trunk Linear(3456, 512) (rlpyt/models/mlp.py:24-31 forward and
// backward) on the bf16 matrix pipe ("bf16x6": both operands split into three bf16 pieces on their
// way HBM -> LDS, six products of order <= 2 accumulated in f32; dropped terms <= 2^-24 |ab|, see
// split_bf16.h / gemm.hip), as ONE kernel body for the three operand layouts
//   NT  C[M,N] = A[M,K]  B[N,K]^T   forward            x W^T
//   NN  C[M,N] = A[M,K]  B[K,N]     input gradient     g W           (no transposed copy of W)
//   TN  C[M,N] = A[K,M]^T B[K,N]    weight gradient    g^T x         (contraction over the batch)
// and a "ping-pong" schedule: the 8 waves of a workgroup are two halves (waves 0-3 / 4-7: one wave
// of each half per SIMD).  While one half issues the 24 MFMAs of a K-32 step back to back (pure
// matrix-pipe segment, 768 cycles), the other half runs its LOAD segment: ds_read the fragments
// of its next step, split the f32 rows it fetched two steps ago into bf16 pieces, ds_write them
// into the other LDS stage, issue the global loads of three steps ahead; then the halves swap
// (two barriers per step).  A SIMD's matrix pipe always has one wave feeding it, and neither LDS
// traffic nor split VALU sits between MFMAs (MI355X_MICROARCH.md "Two waves per SIMD").  Half 0
// stages the A operand, half 1 the B operand.
//
// Tile 128 x 128 of C per workgroup, wave tile 64 x 32 (2 MFMA 32x32x16 tiles), K-step 32 = two
// MFMA K-slices.  LDS: 2 stages x {A, B} x 3 pieces x 128 rows x 80 B (64 B = 32 bf16 along K +
// 16 B pad: ds_read_b128 of 32 rows x 2 K-halves is conflict-free at a pitch of 20 dwords) =
// 122,880 B, one workgroup per CU.
//
// Staging of an operand X whose K axis is contiguous ("KC": X[r][k]): a thread moves 4 float4 =
// 4 k of rows r, r+32, r+64, r+96 (8 lanes cover 128 contiguous bytes of a row) and writes
// 3 x 4 ds_write_b64.  K strided ("KS": X[k][r], the transposing layouts of NN / TN): a thread
// owns a 4 (k) x 4 (r) block -- four float4 loads along r from four consecutive k rows -- packs
// (k, k+1) pairs per r and writes, per piece, one ds_write_b64 (4 k of one r) to each of its four
// LDS rows; lanes are mapped (k group = lane & 7, r group = lane >> 3) so that the 16-lane groups
// of a ds_write_b64 cover all 32 banks.
//
// TN contracts over the batch (K = 8192) and has only 4 x 27 output tiles: K is cut into 8 chunks,
// chunk c <-> XCD c (each XCD streams ITS rows of x and g once through its own L2; the 4 row tiles
// that share an x panel are adjacent work units), every unit writes a partial tile and
// gemm_reduce_slots_kernel sums the slots in a fixed order (deterministic, no atomics).  The
// tiles left over after whole rounds of 32 CUs are cut into `sub` K-parts so that the last round
// is short instead of sparse.
#include <algorithm>
#include "split_bf16.h"

namespace rlpyt {
namespace {

constexpr int PT = 128;                    // tile edge of C
constexpr int P_THREADS = 512;
constexpr int P_BK = 32;                   // K per step (two MFMA slices of 16)
constexpr int P_ROWB = 2 * P_BK + 16;      // 80 B per (piece, row): data + pad
constexpr int P_PB = PT * P_ROWB;          // bytes per piece of one operand
constexpr int P_OB = 3 * P_PB;             // bytes per operand of one stage
constexpr int P_SB = 2 * P_OB;             // bytes per stage
constexpr int P_LDS = 2 * P_SB;            // 122,880
constexpr int KC = 0, KS = 1;              // operand layouts: K contiguous / K strided

// Debug build (-DRLPYT_TIMING): per-wave cycle totals of the loop phases (scripts/debug/
// gemm_pp_timing.py); compiled out of the product.  slots: 0 load segment, 1 barrier after load,
// 2 compute segment, 3 barrier after compute
// PP_SKIP (debug builds only; results are then wrong): bit 0 no fragment reads, 1 no split / LDS
// writes, 2 no global loads in the loop, 3 no MFMAs -- what a phase costs is what leaving it out saves
#ifndef PP_SKIP
#define PP_SKIP 0
#endif
// Fetch register sets (debug builds may pass -DPP_SETS=4): 2 = the rows of step u + 1 are split in
// load segment u and the set is refilled at once with step u + 3; 4 = four sets, refilled at the
// START of a segment, four segments ahead (measured: 4-7 % SLOWER than 2 -- more loads in flight
// lower the CU's vector-memory throughput; profiles/r3_gemm_pp_sweep2.log)
#ifndef PP_SETS
#define PP_SETS 2
#endif
// 1: warp-specialised producer / consumer waves (below); 0: the ping-pong halves
#ifndef PP_WS
#define PP_WS 1
#endif
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

// One half of the workgroup (HALF 0: waves 0-3, stages operand A, computes first; HALF 1: waves
// 4-7, stages B).  X: this half's operand (layout L, leading dimension ld, `dim` rows / columns of
// C on its output axis, tile origin r0), K steps [ks0, ks0 + nk).
template <int L, int HALF>
__device__ __forceinline__ void pp_half(const float* __restrict__ X, int ld, int dim, int r0, int ks0,
                                        int nk, uint8_t* lds, int a_off, int b_off,
                                        f32x16 (&acc)[2]) {
  const int ht = threadIdx.x & 255;
  // ---- this thread's share of the operand: 4 global float4 per step, 4 LDS rows ----------------
  const float* gp[4];
  int soff[4];
  int64_t kstride;
  uint8_t* const lds_op = lds + HALF * P_OB;      // operand base inside a stage
  if constexpr (L == KC) {
    const int row = ht >> 3, kq = ht & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gp[i] = X + (int64_t)min(r0 + row + 32 * i, dim - 1) * ld + (int64_t)ks0 * P_BK + 4 * kq;
      soff[i] = (row + 32 * i) * P_ROWB + kq * 8;
    }
    kstride = 1;
  } else {
    const int mg = ht & 7, ng = 8 * (ht >> 6) + ((ht >> 3) & 7);
    const int col = min(r0 + 4 * ng, dim - 4);   // dim % 4 == 0: blocks are all in or all out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gp[i] = X + ((int64_t)ks0 * P_BK + 4 * mg + i) * ld + col;
      soff[i] = (4 * ng + i) * P_ROWB + mg * 8;
    }
    kstride = ld;
  }
  uint4 fa[2][2][3], fb[2][3];        // [slice][row tile][piece], [slice][piece]
  f32x4 R[PP_SETS][4];
  const int last = nk - 1;
#define PP_FETCH(set_, step_)                                                     \
  {                                                                               \
    const int64_t ko_ = (int64_t)min((step_), last) * P_BK * kstride;             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                 \
      R[set_][i] = *reinterpret_cast<const f32x4*>(gp[i] + ko_);                  \
  }
#define PP_STAGE(set_, stage_) pp_stage<L>(R[set_], lds_op + (stage_) * P_SB, soff);
#define PP_FRAGS(stage_)                                                          \
  _Pragma("unroll") for (int sl = 0; sl < 2; ++sl)                                \
  _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                 \
    const uint8_t* b_ = lds + (stage_) * P_SB + sl * 32 + s * P_PB;               \
    fb[sl][s] = *reinterpret_cast<const uint4*>(b_ + b_off);                      \
    fa[sl][0][s] = *reinterpret_cast<const uint4*>(b_ + a_off);                   \
    fa[sl][1][s] = *reinterpret_cast<const uint4*>(b_ + a_off + 32 * P_ROWB);     \
  }
  // LOAD segment for step u (parity p = u & 1): fragments of step u from stage p; the rows of step
  // u + 1 (register set (u + 1) % PP_SETS) -> pieces -> stage p ^ 1; that set is then refilled with
  // step u + 3 (PP_SETS == 4: instead, step u + 4 is requested first, into the set u % 4 that the
  // previous segment emptied)
  // (debug) keep values alive / opaque when a phase is left out, so that the others survive DCE
#define PP_SINK4(x_) asm volatile("" ::"v"(__builtin_bit_cast(i32x4_, x_)));
#define PP_OPAQUE4(x_)                            \
  {                                               \
    i32x4_ t_;                                    \
    asm volatile("" : "=v"(t_));                  \
    x_ = __builtin_bit_cast(uint4, t_);           \
  }
#define PP_ALL_FRAGS(OP_)                                                         \
  _Pragma("unroll") for (int sl = 0; sl < 2; ++sl)                                \
  _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                 \
    OP_(fb[sl][s]) OP_(fa[sl][0][s]) OP_(fa[sl][1][s])                            \
  }
#define PP_LOAD(p_, free_, use_, u_)                                              \
  {                                                                               \
    if (PP_SETS == 4 && !(PP_SKIP & 4)) PP_FETCH((free_) % PP_SETS, (u_) + 4)                 \
    if (!(PP_SKIP & 1)) PP_FRAGS(p_)                                              \
    else fb[0][0] = *reinterpret_cast<const uint4*>(lds + (p_) * P_SB + b_off);   \
    __builtin_amdgcn_sched_barrier(0);   /* requests and LDS reads before any vmcnt wait */ \
    if (!(PP_SKIP & 2)) PP_STAGE((use_) % PP_SETS, (p_) ^ 1)                      \
    else { _Pragma("unroll") for (int i = 0; i < 4; ++i) PP_SINK4(R[(use_) % PP_SETS][i]) } \
    if (PP_SETS == 2 && !(PP_SKIP & 4)) PP_FETCH((use_) % PP_SETS, (u_) + 3)      \
  }
  // COMPUTE segment: 2 slices x six products (smallest first) x 2 row tiles, nothing else
#define PP_TERM(sl_, sa_, sb_)                                                    \
  acc[0] = mfma32_bf16(fa[sl_][0][sa_], fb[sl_][sb_], acc[0]);                    \
  acc[1] = mfma32_bf16(fa[sl_][1][sa_], fb[sl_][sb_], acc[1]);
#define PP_COMPUTE()                                                              \
  if (PP_SKIP & 8) { PP_ALL_FRAGS(PP_SINK4) } else                                \
  _Pragma("unroll") for (int sl = 0; sl < 2; ++sl) {                              \
    PP_TERM(sl, 2, 0) PP_TERM(sl, 0, 2) PP_TERM(sl, 1, 1)                         \
    PP_TERM(sl, 1, 0) PP_TERM(sl, 0, 1) PP_TERM(sl, 0, 0)                         \
  }

  // the machine scheduler must not move MFMAs (register-only) across the phase boundaries
#define PP_BAR()                            \
  {                                         \
    __builtin_amdgcn_sched_barrier(0);      \
    __syncthreads();                        \
    __builtin_amdgcn_sched_barrier(0);      \
  }
  if (PP_SKIP & 1) { PP_ALL_FRAGS(PP_OPAQUE4) }
  // ---- prologue: stage 0 <- step 0; the sets hold the next steps -----------------------------------
  static_assert(PP_SETS == 2 || PP_SETS == 4, "the loops below index the sets modulo 2 or 4");
  PP_FETCH(0, 0)
  PP_FETCH(1, 1)
  if (PP_SETS == 4) {
    PP_FETCH(2 % PP_SETS, 2)
    PP_FETCH(3 % PP_SETS, 3)
  }
  PP_STAGE(0, 0)
  if (PP_SETS == 2) PP_FETCH(0, 2)
  PP_BAR()
  // one K-32 step of each half, i = step index mod 4 (static): half 0 computes step s + i while
  // half 1 runs the load segment of step s + i, then half 0 runs the load segment of s + i + 1
  // while half 1 computes s + i
#define PP_H0(i_)                                                                 \
  {                                                                               \
    PP_COMPUTE()                                                                  \
    PP_T(2)                                                                       \
    PP_BAR()                                                                      \
    PP_T(3)                                                                       \
    PP_LOAD(((i_) + 1) & 1, ((i_) + 1) & 3, ((i_) + 2) & 3, s + (i_) + 1)         \
    PP_T(0)                                                                       \
    PP_BAR()                                                                      \
    PP_T(1)                                                                       \
  }
#define PP_H1(i_)                                                                 \
  {                                                                               \
    PP_LOAD((i_) & 1, (i_) & 3, ((i_) + 1) & 3, s + (i_))                         \
    PP_T(0)                                                                       \
    PP_BAR()                                                                      \
    PP_T(1)                                                                       \
    PP_COMPUTE()                                                                  \
    PP_T(2)                                                                       \
    PP_BAR()                                                                      \
    PP_T(3)                                                                       \
  }
  int s = 0;
  if constexpr (HALF == 0) {
    PP_LOAD(0, 0, 1, 0)               // enters the loop with the fragments of step 0
    PP_BAR()
    PP_T0()
#pragma unroll 1
    for (; s + 3 < nk; s += 4) {
      PP_H0(0) PP_H0(1) PP_H0(2) PP_H0(3)
    }
    PP_TOUT()
    if (s < nk) PP_H0(0)
    if (s + 1 < nk) PP_H0(1)
    if (s + 2 < nk) PP_H0(2)
  } else {
    PP_BAR()
    PP_T0()
#pragma unroll 1
    for (; s + 3 < nk; s += 4) {
      PP_H1(0) PP_H1(1) PP_H1(2) PP_H1(3)
    }
    PP_TOUT()
    if (s < nk) PP_H1(0)
    if (s + 1 < nk) PP_H1(1)
    if (s + 2 < nk) PP_H1(2)
  }
#undef PP_H1
#undef PP_H0
#undef PP_BAR
#undef PP_ALL_FRAGS
#undef PP_OPAQUE4
#undef PP_SINK4
#undef PP_COMPUTE
#undef PP_TERM
#undef PP_LOAD
#undef PP_FRAGS
#undef PP_STAGE
#undef PP_FETCH
}

// ---------------------------------------------------------------------------------------------
// Warp-specialised variant (PP_WS, the default): waves 4-7 are PRODUCERS (fetch both operands, split,
// write the LDS stage: 8 float4 per thread and K-32 step), waves 0-3 are CONSUMERS (one per SIMD,
// wave tile 64 x 64: 24 fragment reads and 48 MFMAs per step), ONE barrier per step.  What the
// ping-pong halves above measured (profiles/r3_gemm_pp_sweep*.log): MFMA issue is asynchronous on
// gfx950 (24 MFMAs issue in ~80 cycles and drain from a queue), so a single wave per SIMD keeps the
// matrix pipe busy while it waits at the barrier and reads the next fragments -- the two phases of a
// ping-pong step only added barrier waits (3300-3800 cycles per step against 1540 of matrix-pipe
// time, although every pair of its phases overlapped fine).  64 x 64 wave tiles also halve the LDS
// fragment traffic per MFMA (0.5 KB instead of 0.75 KB).
// ---------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ void pp_operand_map(const float* __restrict__ X, int ld, int dim, int r0,
                                               int ks0, int ht, const float* (&gp)[4], int (&soff)[4],
                                               int64_t& kstride) {
  if constexpr (L == KC) {
    const int row = ht >> 3, kq = ht & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gp[i] = X + (int64_t)min(r0 + row + 32 * i, dim - 1) * ld + (int64_t)ks0 * P_BK + 4 * kq;
      soff[i] = (row + 32 * i) * P_ROWB + kq * 8;
    }
    kstride = 1;
  } else {
    const int mg = ht & 7, ng = 8 * (ht >> 6) + ((ht >> 3) & 7);
    const int col = min(r0 + 4 * ng, dim - 4);   // dim % 4 == 0: blocks are all in or all out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gp[i] = X + ((int64_t)ks0 * P_BK + 4 * mg + i) * ld + col;
      soff[i] = (4 * ng + i) * P_ROWB + mg * 8;
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

template <int LA, int LB>
__device__ __forceinline__ void ws_producer(const float* __restrict__ A, const float* __restrict__ B,
                                            const PpShape& sh, int tm, int tn, int ks0, int nk,
                                            uint8_t* lds) {
  const int ht = threadIdx.x & 255;
  const float *gpa[4], *gpb[4];
  int soa[4], sob[4];
  int64_t ksa, ksb;
  pp_operand_map<LA>(A, sh.lda, sh.M, tm * PT, ks0, ht, gpa, soa, ksa);
  pp_operand_map<LB>(B, sh.ldb, sh.N, tn * PT, ks0, ht, gpb, sob, ksb);
  f32x4 Ra[2][4], Rb[2][4];            // two fetch sets: steps u + 1 and u + 2
  const int last = nk - 1;
#define WS_FETCH(set_, step_)                                                     \
  if (!(PP_SKIP & 4)) {                                                           \
    const int64_t ko_ = (int64_t)min((step_), last) * P_BK;                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                               \
      Ra[set_][i] = *reinterpret_cast<const f32x4*>(gpa[i] + ko_ * ksa);          \
      Rb[set_][i] = *reinterpret_cast<const f32x4*>(gpb[i] + ko_ * ksb);          \
    }                                                                             \
  }
#define WS_STAGE(set_, stage_)                                                    \
  if (!(PP_SKIP & 2)) {                                                           \
    pp_stage<LA>(Ra[set_], lds + (stage_) * P_SB, soa);                           \
    pp_stage<LB>(Rb[set_], lds + (stage_) * P_SB + P_OB, sob);                    \
  } else {                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                               \
      asm volatile("" ::"v"(Ra[set_][i]), "v"(Rb[set_][i]));                      \
    }                                                                             \
  }
  // step u of the producer: the rows of step u + 1 -> pieces -> stage (u + 1) & 1, refill the set
#define WS_PSTEP(p_, u_)                                                          \
  {                                                                               \
    WS_STAGE((p_) ^ 1, (p_) ^ 1)                                                  \
    WS_FETCH((p_) ^ 1, (u_) + 3)                                                  \
    PP_T(0)                                                                       \
    WS_BAR()                                                                      \
    PP_T(1)                                                                       \
  }
  {
    const int64_t k0 = 0, k1 = (int64_t)min(1, last) * P_BK, k2 = (int64_t)min(2, last) * P_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Ra[0][i] = *reinterpret_cast<const f32x4*>(gpa[i] + k0 * ksa);
      Rb[0][i] = *reinterpret_cast<const f32x4*>(gpb[i] + k0 * ksb);
      Ra[1][i] = *reinterpret_cast<const f32x4*>(gpa[i] + k1 * ksa);
      Rb[1][i] = *reinterpret_cast<const f32x4*>(gpb[i] + k1 * ksb);
    }
    pp_stage<LA>(Ra[0], lds, soa);
    pp_stage<LB>(Rb[0], lds + P_OB, sob);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Ra[0][i] = *reinterpret_cast<const f32x4*>(gpa[i] + k2 * ksa);
      Rb[0][i] = *reinterpret_cast<const f32x4*>(gpb[i] + k2 * ksb);
    }
  }
  WS_BAR()                              // stage 0 ready
  PP_T0()
  int s = 0;
#pragma unroll 1
  for (; s + 1 < nk; s += 2) {
    WS_PSTEP(0, s)
    WS_PSTEP(1, s + 1)
  }
  PP_TOUT()
  if (s < nk) WS_PSTEP(0, s)
#undef WS_PSTEP
#undef WS_STAGE
#undef WS_FETCH
}

__device__ __forceinline__ void ws_consumer(const uint8_t* lds, int nk, int a_off, int b_off,
                                            f32x16 (&acc)[2][2]) {
  uint4 fa[2][2][3], fb[2][2][3];      // [slice][tile][piece]
#define WS_FRAGS(stage_)                                                          \
  _Pragma("unroll") for (int sl = 0; sl < 2; ++sl)                                \
  _Pragma("unroll") for (int p = 0; p < 3; ++p)                                   \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                 \
    const uint8_t* b_ = lds + (stage_) * P_SB + sl * 32 + p * P_PB + t * 32 * P_ROWB; \
    fa[sl][t][p] = *reinterpret_cast<const uint4*>(b_ + a_off);                   \
    fb[sl][t][p] = *reinterpret_cast<const uint4*>(b_ + b_off);                   \
  }
#define WS_TERM(sl_, sa_, sb_)                                                    \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                   \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                   \
    acc[i][j] = mfma32_bf16(fa[sl_][i][sa_], fb[sl_][j][sb_], acc[i][j]);
  // six products per tile, smallest first; the fragment reads of slice 1 land under slice 0's MFMAs
#define WS_CSTEP(p_)                                                              \
  {                                                                               \
    WS_FRAGS(p_)                                                                  \
    if (!(PP_SKIP & 8)) {                                                         \
      _Pragma("unroll") for (int sl = 0; sl < 2; ++sl) {                          \
        WS_TERM(sl, 2, 0) WS_TERM(sl, 0, 2) WS_TERM(sl, 1, 1)                     \
        WS_TERM(sl, 1, 0) WS_TERM(sl, 0, 1) WS_TERM(sl, 0, 0)                     \
      }                                                                           \
    } else {                                                                      \
      _Pragma("unroll") for (int sl = 0; sl < 2; ++sl)                            \
      _Pragma("unroll") for (int p = 0; p < 3; ++p)                               \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                             \
        asm volatile("" ::"v"(__builtin_bit_cast(i32x4_, fa[sl][t][p])),          \
                     "v"(__builtin_bit_cast(i32x4_, fb[sl][t][p])));              \
      }                                                                           \
    }                                                                             \
    PP_T(2)                                                                       \
    WS_BAR()                                                                      \
    PP_T(3)                                                                       \
  }
  WS_BAR()                              // stage 0 ready
  PP_T0()
  int s = 0;
#pragma unroll 1
  for (; s + 1 < nk; s += 2) {
    WS_CSTEP(0)
    WS_CSTEP(1)
  }
  PP_TOUT()
  if (s < nk) WS_CSTEP(0)
#undef WS_CSTEP
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
  const int half = wave >> 2;                 // 0: stages A, computes first; 1: stages B
  const int wm = wave & 1, wn = (wave >> 1) & 3;   // wave tile: rows 64 wm.., columns 32 wn..
  // (waves w and w + 4 share a SIMD and a row half; their column quarters differ)

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

#if PP_WS
  if (half != 0) {
    ws_producer<LA, LB>(A, B, sh, tm, tn, ks0, nk, lds);
    return;
  }
  {
    // consumers: 2 x 2 waves of 64 x 64; fragment addresses: row (column) lane & 31, K half lane >> 5
    const int cm = wave & 1, cn = wave >> 1;
    const int a_off = (cm * 64 + (lane & 31)) * P_ROWB + (lane >> 5) * 16;
    const int b_off = P_OB + (cn * 64 + (lane & 31)) * P_ROWB + (lane >> 5) * 16;
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
        const int col = tn * PT + cn * 64 + 32 * j + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = tm * PT + cm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < sh.M && col < sh.N) Cout[(int64_t)row * sh.N + col] = acc[i][j][r];
        }
      }
    return;
  }
#endif
  // fragment addresses of this lane: row (column) lane & 31 of an MFMA tile, K half lane >> 5
  const int a_off = (wm * 64 + (lane & 31)) * P_ROWB + (lane >> 5) * 16;
  const int b_off = P_OB + (wn * 32 + (lane & 31)) * P_ROWB + (lane >> 5) * 16;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // The two halves run DIFFERENT instruction streams with the same number of barriers (the branch
  // is wave-uniform): each gets its own register allocation and a straight-line loop body.
  if (half == 0)
    pp_half<LA, 0>(A, sh.lda, sh.M, tm * PT, ks0, nk, lds, a_off, b_off, acc);
  else
    pp_half<LB, 1>(B, sh.ldb, sh.N, tn * PT, ks0, nk, lds, a_off, b_off, acc);
  // D[row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col = lane & 31] of row tile i
  const int col = tn * PT + wn * 32 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tm * PT + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < sh.M && col < sh.N) Cout[(int64_t)row * sh.N + col] = acc[i][r];
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
__global__ __launch_bounds__(P_THREADS) void gemm_tn_pp_kernel(const float* A, const float* B,
                                                               float* C, PpShape sh) {
  gemm_pp_body<KS, KS>(A, B, C, sh);
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
  sh.S = 1; sh.full = sh.tiles_m * sh.tiles_n; sh.sub = 1;
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
  const PpShape sh = pp_shape(M, N, K, K, N);
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
    RL_LAUNCH(gemm_tn_pp_kernel, dim3(pp_grid(sh)), dim3(P_THREADS), 0, s, a, b, c, sh);
    RL_LAUNCH_CHECK();
    return RLPYT_OK;
  }
  RL_CHECK_ARG(workspace && RL_ALIGNED16(workspace), RLPYT_EINVAL,
               "rlpyt_gemm_tn_f32: K >= 2048 needs the workspace of rlpyt_gemm_tn_workspace_bytes");
  tn_plan(sh);
  const int n_tiles = sh.tiles_m * sh.tiles_n;
  const int units = sh.full + (n_tiles - sh.full) * sh.sub;
  float* ws = static_cast<float*>(workspace);
  RL_LAUNCH(gemm_tn_pp_kernel, dim3(8 * units), dim3(P_THREADS), 0, s, a, b, ws, sh);
  RL_LAUNCH_CHECK();
  const int64_t n4 = M * N / 4;
  RL_LAUNCH(gemm_reduce_slots_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, s, ws, c, sh);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
