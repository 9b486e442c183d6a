// exp_mfma.hpp — the blend exponent of 16 splats at the 64 pixels of an 8x8 block on the matrix cores (gfx950).
//
// log2(opacity * G) of a splat at a pixel is a quadratic in the pixel position.  In the frame of the block's centre
// (xl, yl in {-3.5 ... 3.5}):   e = c0 + c1 xl + c2 yl + c3 xl^2 + c4 xl yl + c5 yl^2,   with, for the record's exponent form
// (A, B, C, L of common.hpp) and (X, Y) = splat centre - block centre,
//     c0 = L + A X^2 + B X Y + C Y^2,   c1 = -(2 A X + B Y),   c2 = -(2 C Y + B X),   c3 = A,  c4 = B,  c5 = C.
// The six monomials are exact in bf16 (multiples of 1/4 below 16).  Each fp32 coefficient is cut into THREE bf16 terms by
// truncation, c = t0 + t1 + t2 exactly (8 + 8 + 8 = 24 mantissa bits; bf16 has fp32's exponent range, so no scaling is
// needed - an fp16 split would need one).  All products are therefore exact, the accumulation is the matrix core's fp32.
// No operand is held at reduced precision: this is the fp32 polynomial in another association order.
//
// One v_mfma_f32_32x32x16_bf16 per term (K = 16 = two blocks of 8: six monomials + two spare slots each).  The output tile
// D[32][32] puts column n on lane n + 32 h and rows {(i & 3) + 8 (i >> 2) + 4 h} in register i, so a lane only ever sees
// the rows of ITS half h.  Rows (i & 3) + 8 (i >> 2) + 4 h, h = 0, 1, both carry splat i; row set h has its coefficients
// in K-block h and zeros in the other, and K-block h of the B operand holds the monomials of pixels 32 h + n.  Then
//     register i of lane l  =  e(splat i, pixel l)          for all 16 splats and all 64 pixels,
// lane = pixel, exactly the layout the sequential blend wants - no transpose, no LDS round trip for the result.
// Half of the multiply-adds are zeros: 3 x 32 matrix-pipe cycles per 16 splats = 6 per (block, splat), against 7 VALU
// instructions (~22 issue cycles) per (block, splat) for the Horner form on the vector pipe.
#pragma once
#include "common.hpp"

namespace scorp {
#ifdef __HIPCC__
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Bf16Frag { bf16x8 v; uint4 q; uint32_t d[4]; };

// (hi16(a)) | (hi16(b) << 16): two truncated-bf16 values in one dword, a in the low half
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// c = t0 + t1 + t2 exactly, each t a bf16 (returned as fp32 bit patterns whose low 16 bits are zero)
__device__ __forceinline__ void split3_bf16(float c, uint32_t &t0, uint32_t &t1, uint32_t &t2) {
  t0 = __float_as_uint(c) & 0xFFFF0000u;
  const float r1 = c - __uint_as_float(t0);              // exact: the low 16 mantissa bits
  t1 = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(t1);             // exact: at most 8 significant bits left
  t2 = __float_as_uint(r2);
}

// The three A-operand terms of one splat for a block centred at (cx, cy): element j of term k = k-th bf16 term of c_j.
__device__ __forceinline__ void splat_block_coefs(float x, float y, float A, float B, float C, float L, float cx, float cy,
                                                  uint4 &k0, uint4 &k1, uint4 &k2) {
  const float X = x - cx, Y = y - cy;
  float c[6];
  c[0] = splat_exponent(X, Y, A, B, C, L);
  c[1] = -__builtin_fmaf(2.0f * A, X, B * Y);
  c[2] = -__builtin_fmaf(2.0f * C, Y, B * X);
  c[3] = A; c[4] = B; c[5] = C;
  uint32_t t[3][6];
#pragma unroll
  for (int j = 0; j < 6; j++) split3_bf16(c[j], t[0][j], t[1][j], t[2][j]);
  k0 = make_uint4(pack_hi16(t[0][0], t[0][1]), pack_hi16(t[0][2], t[0][3]), pack_hi16(t[0][4], t[0][5]), 0u);
  k1 = make_uint4(pack_hi16(t[1][0], t[1][1]), pack_hi16(t[1][2], t[1][3]), pack_hi16(t[1][4], t[1][5]), 0u);
  k2 = make_uint4(pack_hi16(t[2][0], t[2][1]), pack_hi16(t[2][2], t[2][3]), pack_hi16(t[2][4], t[2][5]), 0u);
}

// The reference skips a splat at a pixel where power > 0, i.e. where opacity * G exceeds the opacity itself.  For a
// positive-definite conic that never happens (up to rounding, within which exp2(e) = o (1 + 1e-5) is the right answer,
// not "skip"), and the hot loops carry no such test.  It can happen for an INDEFINITE conic (det < 0 passes the
// reference's cull), so groups that hold one run the guarded instantiation: exp2(e) <= lim with lim = the two-term bf16
// truncation of o (1 + 4e-5) (>= o (1 + 2.4e-5)), which rides in the spare fourth dword of the first coefficient term -
// K slots 6, 7 of the A operand, which meet zeros in the B operand (both halves are finite bf16 values, so 0 x them = 0).
__device__ __forceinline__ uint32_t guard_limit_pack(float opacity) {
  const float lim = opacity * 1.00004f;
  const uint32_t t0 = __float_as_uint(lim) & 0xFFFF0000u;
  const uint32_t t1 = __float_as_uint(lim - __uint_as_float(t0)) & 0xFFFF0000u;
  return pack_hi16(t0, t1);
}
__device__ __forceinline__ float guard_limit_unpack(uint32_t w) { return __uint_as_float(w << 16) + __uint_as_float(w & 0xFFFF0000u); }
// a record of SplatRec whose conic is not positive-definite: preprocess marks it with kcut = +inf
__device__ __forceinline__ bool rec_is_indefinite(float kcut) { return kcut == __builtin_inff(); }

// B operand of this lane: the six monomials of its own pixel (lane = pixel of the 8x8 block, row-major), bf16, exact.
__device__ __forceinline__ uint4 pixel_basis_frag(int lane) {
  const float xl = (float)(lane & 7) - 3.5f, yl = (float)(lane >> 3) - 3.5f;
  const uint32_t one = __float_as_uint(1.0f), ux = __float_as_uint(xl), uy = __float_as_uint(yl);
  const uint32_t uxx = __float_as_uint(xl * xl), uxy = __float_as_uint(xl * yl), uyy = __float_as_uint(yl * yl);
  return make_uint4(pack_hi16(one, ux), pack_hi16(uy, uxx), pack_hi16(uxy, uyy), 0u);
}

// Which splat of a 16-group this lane supplies to the A operand (its row r = lane & 31 belongs to splat (r & 3) + 4 (r >> 3)),
// and whether its K-block (lane >> 5) is the one that row set uses; the other lanes supply zeros.
__device__ __forceinline__ int a_operand_slot(int lane) { const int r = lane & 31; return (r & 3) + 4 * (r >> 3); }
__device__ __forceinline__ bool a_operand_active(int lane) { return ((lane >> 2) & 1) == (lane >> 5); }

// e[i] = log2(opacity * G) of splat i at this lane's pixel, i = 0..15, from the lane's three A fragments
__device__ __forceinline__ f32x16 block_exponents(uint4 a0, uint4 a1, uint4 a2, uint4 basis) {
  Bf16Frag fa0, fa1, fa2, fb;
  fa0.q = a0; fa1.q = a1; fa2.q = a2; fb.q = basis;
  f32x16 acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  // smallest terms first: the sum of the low-order terms is formed before it meets the leading one
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2.v, fb.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1.v, fb.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0.v, fb.v, acc, 0, 0, 0);
  return acc;
}
#endif
}  // namespace scorp
