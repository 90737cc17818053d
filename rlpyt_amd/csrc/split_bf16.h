// f32 contractions on the bf16 matrix pipe: exact three-piece bf16 splits and the MFMA wrapper shared
// by the GEMM kernels (gemm.hip, gemm_tn.hip).  x == hi + mid + lo exactly for finite x (3 x 8
// significand bits, every remainder exact in f32); products of pieces are exact in the MFMA's f32
// accumulator (tests/test_split_arith.py pins the arithmetic on the CPU).
#pragma once
#include "common.h"

namespace rlpyt {

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
// (x0, x1) -> three packed bf16 pieces (x0 in the low half), hi + mid + lo == x exactly
__device__ __forceinline__ void split3_rn(float x0, float x1, uint32_t& hi, uint32_t& mid,
                                          uint32_t& lo) {
  hi = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(r0, r1);
  lo = cvt_pk_bf16(r0 - __uint_as_float(mid << 16), r1 - __uint_as_float(mid & 0xffff0000u));
}

// ds_read_b64_tr_b16, the LDS transpose read of gfx950: within a group of 16 lanes, lane s supplies
// the address of four contiguous 16-bit elements = "key s >> 2, columns 4 (s & 3) .. + 3" of a
// 4 x 16 block, and lane i receives column i of the block, keys 0..3 (scripts/debug/tr_probe.hip
// prints the map).  Two of them give a lane the 8 consecutive K of an MFMA operand from data that is
// stored K-major.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const uint8_t* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(uint2, v);
}

}  // namespace rlpyt
