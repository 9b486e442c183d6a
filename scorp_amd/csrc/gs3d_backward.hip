// gs3d_backward.hip — 3DGS backward for gfx950: per-tile back-to-front replay -> per-Gaussian chain rule.
//
// Replaces the backward half of `diff_gaussian_rasterization` (reached from `loss.backward()`,
// train_3dgs.py:152; colour-only at post_refine_gs.py:53-56; w.r.t. colors_precomp at utils/mask.py:47-70).
// Arithmetic follows oracle/gs3d_oracle.c::gs3d_oracle_backward.  The forward state and the pair buffer are
// only read, so the backward can be replayed on one forward.

#include <type_traits>

#include "common.hpp"
#include "exp_mfma.hpp"

namespace scorp {
namespace {

// ---------------------------------------------------------------------------------------------------------
// The blend backward: a back-to-front replay of each pixel's blend with the pixel->splat reduction done on the
// matrix cores.
//
// For one splat the ten sums over pixels factor as  sum_p v_p * {1, x_p, y_p, x_p^2, x_p y_p, y_p^2}  (geometry:
// v_p = G * opacity * dL/dalpha, pixel coordinates relative to the block centre) and  sum_p w_p * {dL/dr, dL/dg, dL/db,
// dL/ddepth}_p  (w_p = alpha * T).  With a wave's 64 pixels as the K dimension that is
// D[16 splats][16 columns] = [V | W] * [basis_v ; basis_w].
//
// Per wave and 16-splat group:
//   1a (straight-line, no dependence between splats): alpha and G*opacity of the 16 splats at this lane's pixel;
//   1b (the sequential part): T, the blended-behind recurrence and dL/dalpha.  A splat that does not contribute to
//      this pixel is carried through as alpha = 0, which leaves every recurrence bit-for-bit unchanged, so the
//      group is branch-free.  The five "accumulated colour/depth/alpha behind" recurrences of the textbook form
//      collapse into ONE, because only their dot product with this pixel's upstream gradient is ever used:
//      s = c.dL/dcolor + z*dL/ddepth + dL/dalpha,  R <- last_alpha * s_last + (1 - last_alpha) * R,
//      dL/dalpha_i = (s - R) * T - T_final / (1 - alpha) * (bg . dL/dcolor);
//   lane (= pixel) writes v, w into two wave-private LDS matrices [slot][pixel] (row stride 68 dwords: conflict-free
//   row writes, 16-byte-aligned rows for the transposed A-operand reads);
//   the 16x16 MFMA result holds block-frame moments, which sixteen lanes turn into the ten screen-space gradients;
//   they are flushed with float atomics shaped as whole 40-byte row segments (lanes = consecutive floats).
//
// Two forms of the reduction (template parameter kExact, entry point scorp_gs3d_backward_ex):
//   * SPLIT (default): v and w leave the lane as TWO fp16 terms each - h1 = rtz16(x), h2 = rtz16(x - h1), 22 bits
//     together (fp32 carries 24) - packed (v | w) into one dword per term, so a slot costs one ds_write2_b32.  With
//     K = (pixel, v-or-w) the whole reduction is 8 v_mfma_f32_16x16x32_f16 per group (two passes, one per term, over
//     ONE B operand): columns 0..5 hold the position basis (half-integer block coordinates and their products, exact
//     in fp16) against the v slots, columns 6..9 / 10..13 the high / low fp16 term of the upstream gradients against
//     the w slots.  The products are exact and accumulate in fp32.  Range: the pixel's upstream gradients are
//     pre-scaled by a power of two chosen per wave from the block's largest |dL/dpixel| (so v sits around 2^7 ... and
//     the conversion saturates, RTZ, instead of overflowing), w by 2^10; the sums are unscaled by exact powers of two.
//   * EXACT (kExact): v and w stay fp32, 16 + 16 v_mfma_f32_16x16x4_f32 per group (fp32 MFMAs run at the fp32 VALU
//     rate and their time is paid in full, scripts/mb_mfma_valu.hip: 608 -> 128 matrix cycles per group is what the
//     split form buys).  The parity tests compare the two forms with each other and each with the oracle.
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// Timing probes only (scripts/dev/lds_conflicts.py; profiles/r06_3d_backward_bank_conflicts.txt): SCORP_BWD_XSTRIDE=72 with
// SCORP_BWD_KOFF=4 is the conflict-free A-operand read of the 2-D kernel (its K mapping does not match this kernel's B operand:
// WRONG gradients, same instructions and LDS traffic), SCORP_BWD_LDS_PAD pads the other arm to the same LDS footprint.
#ifndef SCORP_BWD_XSTRIDE
#define SCORP_BWD_XSTRIDE 68
#endif
#ifndef SCORP_BWD_KOFF
#define SCORP_BWD_KOFF 16
#endif
#ifndef SCORP_BWD_LDS_PAD
#define SCORP_BWD_LDS_PAD 0
#endif
constexpr int kXStride = SCORP_BWD_XSTRIDE;  // dwords per slot row of the wave-private matrices: rows stay 16-byte
                                             // aligned for the A-operand's ds_read_b128; row writes are conflict-free
constexpr int kGroup = 16;                   // splats per MFMA group
#ifndef SCORP_BWD_DSTRIDE
#define SCORP_BWD_DSTRIDE 20
#endif
constexpr int kDStride = SCORP_BWD_DSTRIDE;   // floats per slot row of the 16 x 14 result tile: 20 (80 bytes, 16-byte aligned rows) keeps the
                                             // sixteen moment lanes' 16-byte row reads on distinct banks (stride 16: lanes 0, 4, 8, 12 collide)
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kVTargetExp = 7;               // split form: the block's largest |dL/dpixel| is scaled into [2^7, 2^8)
constexpr int kVTargetExpDA = 4;             // ... [2^4, 2^5) when depth / alpha gradients (depth values!) take part
constexpr float kWScale = 1024.0f;           // split form: blend weights (<= 1) are carried as w * 2^10

__device__ __forceinline__ uint32_t pack_rtz16(float lo, float hi) {   // (fp16 rtz(lo)) | (fp16 rtz(hi)) << 16
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lo, hi));
}
__device__ __forceinline__ float half_lo(uint32_t p) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(p & 0xFFFFu)); }
__device__ __forceinline__ float half_hi(uint32_t p) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(p >> 16)); }

// ---------------------------------------------------------------------------------------------------------
// B1w: the replay with ONE WAVE PER 8x8 PIXEL BLOCK as the unit of work (64-thread workgroups, no workgroup
// barriers, no waiting for the slowest wave of a tile).  A wave walks its block's HIT LIST (left by the forward: the
// splats that passed the exact ellipse-vs-block test, compacted, in blend order) back to front 64 hits at a time: each
// lane gathers one record (list entries are fetched two chunks ahead, records one) into a 64-slot staging buffer in the
// wave's own LDS, and the chunk's four groups of 16 go through the 1a / 1b / MFMA pipeline described above.  A hit's
// position in the list is arithmetic (no per-slot position array): slot i of a group whose first slot has 1-based
// position `top` belongs to the pixel's blend iff top - i <= last contributor of the pixel.  The 16x16 result is
// converted from block-frame moments to the ten gradients by lanes 0..15 and added to the per-Gaussian accumulators
// with float atomics (zero terms are skipped).  Blocks are numbered so that the four waves of a tile land on the same
// XCD (same L2).
// ---------------------------------------------------------------------------------------------------------
constexpr int kChunk = 64;

#ifndef SCORP_BWD_WAVES
#define SCORP_BWD_WAVES 4
#endif
// kHasDA false: no upstream gradient on the depth / alpha images (the photometric-loss-only step).
// kColorOnly: only dL/dcolour is wanted (post_refine_gs.py:53-56 freezes xyz / scale / rotation / opacity and trains SH0):
// no dL/dalpha, no "blended behind" recurrence, no v; the matrix rows carry w alone, three sums per splat leave.
// fp16 bits of the V columns of the B operand: kBasisV[bn][q] = column bn (1, x, y, x^2, xy, y^2 in the block frame, half-
// integer coordinates: exact) at pixel q of the 8x8 block.  A lane's sixteen values are consecutive (q = 16 bk .. + 15).
struct BasisTable { uint16_t v[6][64]; };
constexpr uint16_t f16_bits_small(float x) {   // exact for the multiples of 0.25 below 16 that occur here
  if (x == 0.0f) return 0;
  const uint16_t sign = x < 0.0f ? 0x8000u : 0u;
  float a = x < 0.0f ? -x : x;
  int e = 15;
  while (a >= 2.0f) { a *= 0.5f; e++; }
  while (a < 1.0f) { a *= 2.0f; e--; }
  return (uint16_t)(sign | (e << 10) | (uint16_t)((a - 1.0f) * 1024.0f));
}
constexpr BasisTable make_basis_table() {
  BasisTable t{};
  for (int q = 0; q < 64; q++) {
    const float xl = (float)(q & 7) - 3.5f, yl = (float)(q >> 3) - 3.5f;
    const float col[6] = {1.0f, xl, yl, xl * xl, xl * yl, yl * yl};
    for (int c = 0; c < 6; c++) t.v[c][q] = f16_bits_small(col[c]);
  }
  return t;
}
__device__ const BasisTable kBasisV = make_basis_table();

// kDet: SCORP_BACKWARD_DETERMINISTIC - the sums leave as plain rows partial[quad][hit-list position] instead of float atomics
template <bool kHasDA, bool kExact, bool kColorOnly = false, bool kDet = false>
__global__ void __launch_bounds__(64, SCORP_BWD_WAVES)
blend_backward_wave_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ hits,
                           const SplatRec *__restrict__ rec, uint32_t capacity, int W, int H, int tiles_x, int tiles,
                           const float *__restrict__ bg, const float *__restrict__ final_T,
                           const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dcolor,
                           const float *__restrict__ dL_ddepth, const float *__restrict__ dL_dalpha,
                           float *__restrict__ acc, float *__restrict__ partial, uint8_t *__restrict__ row_flags,
                           const uint32_t *__restrict__ pair_base, const BinRec *__restrict__ bin,
                           const uint64_t *__restrict__ tile_mask) {
  // staged hits: the three bf16 terms of the six block-frame coefficients of log2(opacity * G) (exp_mfma.hpp), the
  // blended values (r, g, b, depth) for the recurrence, and what the moment step needs: (x - cx, y - cy, A, B), (C, opacity), id
  __shared__ uint4 q_k[3][kChunk];
  __shared__ float4 q_col[kChunk], q_t0[kChunk];
  __shared__ float2 q_t1[kChunk];
  __shared__ __attribute__((aligned(16))) uint32_t q_id[kChunk];
  // [row][pixel] matrix of one half-group (8 splats).  split form: rows 0..7 = first fp16 terms (v | w << 16) of the 8
  // splats, rows 8..15 = second terms; exact form: rows 0..7 = v, rows 8..15 = w (fp32)
  __shared__ __attribute__((aligned(16))) uint32_t xm[16 * kXStride];
  float *dbuf = reinterpret_cast<float *>(xm);   // the 2 x 16 x 14 result tile reuses the matrix once the MFMAs have consumed it
  float *xs = reinterpret_cast<float *>(xm);     // prologue scratch: 64 x 4 floats
  static_assert(2 * kGroup * kDStride <= 13 * kXStride && 64 * 4 <= 13 * kXStride, "scratch fits below the zero fragments");
  static_assert(kXStride != 68 || SCORP_BWD_LDS_PAD != 0 ||
                sizeof(uint4) * 3 * kChunk + sizeof(float4) * 2 * kChunk + sizeof(float2) * kChunk + 4 * kChunk + 4 * 16 * kXStride == 10240,
                "10 KiB of LDS per wave: four waves per SIMD fill the CU's 160 KiB exactly");
#if SCORP_BWD_LDS_PAD
  __shared__ uint32_t lds_pad[SCORP_BWD_LDS_PAD];
  if (W < 0) lds_pad[threadIdx.x] = 1u;   // (never taken: keeps the padding allocated)
#endif
  const int lane = threadIdx.x;
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int tile = (kk >> 2) * 8 + xcd, quad = kk & 3;
  if (tile >= tiles) return;
  const int tx0 = (tile % tiles_x) * kTile, ty0 = (tile / tiles_x) * kTile;
  const int bx = tx0 + (quad & 1) * 8, by = ty0 + (quad >> 1) * 8;
  const int px = bx + (lane & 7), py = by + (lane >> 3);
  const bool inside = px < W && py < H;
  const float cx = (float)bx + 3.5f, cy = (float)by + 3.5f;  // block-frame origin of the exponent's coefficients and of the moments
  // A operand of the exponent MFMAs: half of the lanes feed zeros (exp_mfma.hpp).  The zero fragments live in the four
  // padding dwords of rows 13..15 of the [row][pixel] matrix (nothing ever writes there: rows are written [row][lane])
  const uint4 basis = pixel_basis_frag(lane);
  const bool a_on = a_operand_active(lane);
  const int a_slot = a_operand_slot(lane);
  if (lane < 3) *reinterpret_cast<uint4 *>(&xm[(13 + lane) * kXStride + 64]) = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t beg = min(tile_start[2 * tile], capacity), end = min(tile_start[2 * tile + 1], capacity);   // (start, end) per tile
  // deterministic mode: the sums leave as plain rows partial[4 * pair + quad], pair = the (Gaussian, tile) pair's ordinal
  // in GAUSSIAN-major order (pair_base[id] + the tile's rank in the Gaussian's tile mask), with a flag byte per row
  constexpr bool det = kDet;
  if (end == beg) return;
  const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
  // all of the pixel's loads are issued together (no load waits on `last`); pixels nothing was blended into drop
  // their upstream gradient afterwards by a select (it may be NaN: depth / alpha at empty pixels)
  float T_final = 0.0f, dpix0 = 0.0f, dpix1 = 0.0f, dpix2 = 0.0f, ddep = 0.0f, dalp = 0.0f;
  uint32_t last = 0u;
  if (inside) {
    T_final = final_T[pix];
    last = n_contrib[pix];
    dpix0 = dL_dcolor[pix]; dpix1 = dL_dcolor[HW + pix]; dpix2 = dL_dcolor[2 * HW + pix];
    if (kHasDA && dL_ddepth) ddep = dL_ddepth[pix];
    if (kHasDA && dL_dalpha) dalp = dL_dalpha[pix];
  }
  if (last == 0) { dpix0 = dpix1 = dpix2 = ddep = dalp = 0.0f; }
  const uint32_t todo = wave_max_u32(last);   // wave-uniform (SGPR): the chunk loop and the slot indices live in SGPRs
  if (todo == 0) return;
  // split form: one power-of-two scale per wave from the block's largest upstream gradient
  float sv = 1.0f, inv_sv = 1.0f;
  if constexpr (!kExact) {
    float amax = fmaxf(fmaxf(fabsf(dpix0), fabsf(dpix1)), fabsf(dpix2));
    if (kHasDA) amax = fmaxf(amax, fmaxf(fabsf(ddep), fabsf(dalp)));
    const int eb = (int)((wave_max_u32(__float_as_uint(amax)) >> 23) & 0xFFu);   // biased exponent (|x| orders like its bits); 0: zero / denormal
    int sb = eb == 0 ? 127 : 254 + (kHasDA ? kVTargetExpDA : kVTargetExp) - eb;   // biased exponent of the scale
    sb = min(max(sb, 1), 253);
    sv = __uint_as_float((uint32_t)sb << 23);
    inv_sv = __uint_as_float((uint32_t)(254 - sb) << 23);
  }
  // In the split form the recurrence runs on T' = 2^10 T (so that w' = alpha T' is the scaled blend weight at no cost)
  // and on upstream gradients scaled by sv / 2^10, which makes dL/dalpha come out scaled by sv:
  //   sv dL/dalpha = (s'' - R'') T' - (sv T_final bg.dL/dcolor) / (1 - alpha).
  const float tf_bg = sv * (T_final * (bg[0] * dpix0 + bg[1] * dpix1 + bg[2] * dpix2));
  const int bn = lane & 15, bk = lane >> 4;
  // Every lane already holds its own pixel's four gradients, so the B operand's W columns are exchanged through LDS
  // (the matrices are idle until the first group) instead of being gathered from global memory again.
  xs[lane * 4 + 0] = sv * dpix0; xs[lane * 4 + 1] = sv * dpix1; xs[lane * 4 + 2] = sv * dpix2; xs[lane * 4 + 3] = sv * ddep;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // B operand.  A lane supplies ONE column bn of the 16-column basis: columns 0..5 are the position basis of the V half
  // (1, x, y, x^2, xy, y^2 in the block frame), the following ones the upstream gradients (dL/dr, dL/dg, dL/db,
  // dL/ddepth) of the W half.
  union Frag { f16x8 v; uint4 q; uint32_t d[4]; };
  float bb[kExact ? 16 : 1];
  Frag bh[kExact ? 1 : 4];
  if constexpr (kExact) {
#pragma unroll
    for (int t = 0; t < 16; t++) {   // fp32 MFMA t covers pixels t, t + 16, t + 32, t + 48
      const int q = t + 16 * bk;
      const float xl = (float)(q & 7) - 3.5f, yl = (float)(q >> 3) - 3.5f;
      float v = 0.0f;
      v = bn == 0 ? 1.0f : v; v = bn == 1 ? xl : v; v = bn == 2 ? yl : v;
      v = bn == 3 ? xl * xl : v; v = bn == 4 ? xl * yl : v; v = bn == 5 ? yl * yl : v;
      v = (bn >= 6 && bn <= 9) ? xs[q * 4 + ((bn - 6) & 3)] : v;
      bb[t] = v;
    }
  } else {
    // fp16 MFMA m covers pixels 16 bk + 4 m + j, j = 0..3: element 2j = v slot, 2j + 1 = w slot.  V columns (bn < 6) come
    // from the constant table (two 16-byte loads per lane), W columns from the exchanged gradients.
    uint4 tv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (bn < 6) {
      const uint4 *tp = reinterpret_cast<const uint4 *>(&kBasisV.v[bn][16 * bk]);
      tv[0] = tp[0]; tv[1] = tp[1];
    }
    const uint32_t tw[8] = {tv[0].x, tv[0].y, tv[0].z, tv[0].w, tv[1].x, tv[1].y, tv[1].z, tv[1].w};
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 16 * bk + 4 * m + j, e = 4 * m + j;
        const uint32_t vbits = (e & 1) ? tw[e >> 1] >> 16 : tw[e >> 1] & 0xFFFFu;
        const float g = (bn >= 6 && bn <= 13) ? xs[q * 4 + ((bn - 6) & 3)] : 0.0f;
        const uint32_t g1 = pack_rtz16(g, 0.0f);
        const float gw = bn <= 9 ? half_lo(g1) : g - half_lo(g1);     // columns 6..9: first term, 10..13: the remainder
        bh[m].d[j] = vbits | (pack_rtz16(gw, 0.0f) << 16);
      }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if constexpr (!kExact) {
    const float sq = sv * (1.0f / kWScale);
    dpix0 *= sq; dpix1 *= sq; dpix2 *= sq; ddep *= sq; dalp *= sq;
  }
  float T = kExact ? T_final : kWScale * T_final, R = 0.0f, s_last = 0.0f, last_alpha = 0.0f;
  const int abase = (lane & 15) * kXStride + SCORP_BWD_KOFF * (lane >> 4);
  constexpr int kAStep = SCORP_BWD_KOFF == 16 ? 4 : 16;   // dwords between the lane's four 16-byte reads
  float park_v[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // a group's sums, parked until flush_sums
  uint32_t park_o[4] = {det ? 0xFFFFFFFFu : 0u, det ? 0xFFFFFFFFu : 0u, det ? 0xFFFFFFFFu : 0u, det ? 0xFFFFFFFFu : 0u};
                                                // ... and their float offsets in acc (N * 16 < 2^32, checked at the entry
                                                // point); det: offsets in `partial`, 0xFFFFFFFF = nothing parked
  auto flush_sums = [&]() {
    if (det) {   // rows are written whole (zeros included): nothing was cleared beforehand
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (park_o[k] != 0xFFFFFFFFu) partial[park_o[k]] = park_v[k];
        park_o[k] = 0xFFFFFFFFu;
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
#ifdef SCORP_BWD_NOATOMIC
      if (park_v[k] != 0.0f && capacity == 0xFFFFFFFFu) atomicAdd(acc + park_o[k], park_v[k]);
#else
      if (park_v[k] != 0.0f) atomicAdd(acc + park_o[k], park_v[k]);
#endif
      park_v[k] = 0.0f;
    }
  };
  // slots of a group: head + i, head a multiple of kGroup: one LDS base per array, immediate offsets.  `top` = 1-based
  // position in the hit list of the group's first slot (positions fall by one per slot).
  // A group is two halves of 8 splats.  Each half: 1a, 1b, then ONE pass over the matrix pipe whose 16 rows are
  // (8 slots) x (first term | second term) [exact form: (8 slots) x (v | w)], so the [row][pixel] matrix in LDS is
  // 16 x 64 dwords whatever the form; the two halves' results wait in registers and leave together.
  auto process_group = [&](auto full, auto clampy, int nslots, int head, int top) {
    flush_sums();   // (a later group of the same chunk: the previous one's sums leave now)
    constexpr bool kFull = decltype(full)::value;   // full groups run straight-line; only a wave's last one is partial
    constexpr bool kClamp = decltype(clampy)::value;   // the chunk holds a splat with opacity > 0.99 (the forward's rule:
                                                       // only such a splat can reach alpha = 0.99; the others skip the v_min)
    int hv = head;
    asm volatile("" : "+v"(hv));   // keep the group's LDS bases in VGPRs (else every ds_read re-moves an SGPR base)
    const float4 *gcol = q_col + hv, *gt0 = q_t0 + hv;
    const float2 *gt1 = q_t1 + hv;
    const int first = top - (int)last;   // slot i takes part in this pixel's blend iff top - i <= last, i.e. i >= first
    // log2(opacity * G) of the group's 16 hits at this lane's pixel: three MFMAs (the forward's, bit for bit)
    const uint4 *ka0 = a_on ? &q_k[0][hv + a_slot] : reinterpret_cast<const uint4 *>(&xm[13 * kXStride + 64]);
    const uint4 *ka1 = a_on ? &q_k[1][hv + a_slot] : reinterpret_cast<const uint4 *>(&xm[14 * kXStride + 64]);
    const uint4 *ka2 = a_on ? &q_k[2][hv + a_slot] : reinterpret_cast<const uint4 *>(&xm[15 * kXStride + 64]);
    const f32x16 ev = block_exponents(*ka0, *ka1, *ka2, basis);
    f32x4 dd[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (kFull || h * 8 < nslots) {   // wave-uniform
        // One straight-line pass per splat (alpha, then the recurrence): with four waves per SIMD an in-order wave's own
        // issue interval already covers the ALU latencies, so nothing is gained by separating a parallel alpha phase
        // from the sequential part - and keeping eight (alpha, record) sets live costs the registers that let the
        // staged records of the NEXT splats be fetched from LDS ahead of their use.
        // The staged records are fetched kAhead splats ahead of their use (explicit rotation: left alone the scheduler
        // issues each ds_read right in front of its use and the wave eats the LDS latency once per splat).
#ifndef SCORP_BWD_AHEAD
#define SCORP_BWD_AHEAD 3
#endif
        constexpr int kAhead = SCORP_BWD_AHEAD;
        float4 rc[kAhead];
#pragma unroll
        for (int j = 0; j < kAhead - 1; j++) rc[j] = gcol[h * 8 + j];
#pragma unroll
        for (int i8 = 0; i8 < 8; i8++) {
          const int i = h * 8 + i8;
          if (i8 + kAhead - 1 < 8) rc[(i8 + kAhead - 1) % kAhead] = gcol[i + kAhead - 1];
          if (kFull || i < nslots) {  // wave-uniform: stale staging entries beyond the group must not enter the recurrence
            const float4 col = rc[i8 % kAhead];   // r, g, b, depth
            const float g_o = __builtin_amdgcn_exp2f(ev[i]);
            // alpha = min(0.99, g_o) >= 1/255  <=>  g_o >= 1/255
            bool ok = (i >= first) & (g_o >= kAlphaMin);
            if constexpr (kClamp) ok = ok & (g_o <= guard_limit_unpack(q_k[0][hv + i].w));   // the forward's `power > 0` skip
            const float Go = ok ? g_o : 0.0f;
            const float alpha = kClamp ? fminf(kAlphaMax, Go) : Go;
            const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
            T *= rinv;
            const float w = alpha * T;
            float v = 0.0f;
            if constexpr (!kColorOnly) {
              R = last_alpha * (s_last - R) + R;
              const float sc = kHasDA ? col.x * dpix0 + col.y * dpix1 + col.z * dpix2 + col.w * ddep + dalp
                                      : col.x * dpix0 + col.y * dpix1 + col.z * dpix2;
              const float dL_dal = (sc - R) * T - tf_bg * rinv;
              s_last = sc;
              last_alpha = alpha;
              v = Go * dL_dal;
            }
            if constexpr (kExact) {
              xm[i8 * kXStride + lane] = __float_as_uint(v);
              xm[(8 + i8) * kXStride + lane] = __float_as_uint(w);
            } else if constexpr (kColorOnly) {
              const uint32_t p1 = pack_rtz16(0.0f, w);
              xm[i8 * kXStride + lane] = p1;
              xm[(8 + i8) * kXStride + lane] = pack_rtz16(0.0f, __builtin_fmaf(half_hi(p1), -1.0f, w));
            } else {   // x = h1 + h2, two fp16 terms (round toward zero, saturating); v and w share the dwords
              const uint32_t p1 = pack_rtz16(v, w);
              const uint32_t p2 = pack_rtz16(__builtin_fmaf(half_lo(p1), -1.0f, v), __builtin_fmaf(half_hi(p1), -1.0f, w));
              xm[i8 * kXStride + lane] = p1;
              xm[(8 + i8) * kXStride + lane] = p2;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // A operands: the lane's 16 pixels are consecutive in its row, fetched as four 16-byte reads issued together
        f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (kExact) {
          float4 av[4];
#pragma unroll
          for (int t4 = 0; t4 < 4; t4++) av[t4] = *reinterpret_cast<const float4 *>(&xm[abase + kAStep * t4]);
#ifdef SCORP_BWD_PROBE_HALF_MFMA
          if (h == 0)     // diagnostic build (wrong gradients): what halving the fp32 matrix work would buy
#endif
#pragma unroll
          for (int t4 = 0; t4 < 4; t4++) {
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4].x, bb[4 * t4], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4].y, bb[4 * t4 + 1], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4].z, bb[4 * t4 + 2], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4].w, bb[4 * t4 + 3], d, 0, 0, 0);
          }
        } else {
          Frag af[4];
#pragma unroll
          for (int m = 0; m < 4; m++) af[m].q = *reinterpret_cast<const uint4 *>(&xm[abase + kAStep * m]);
#pragma unroll
          for (int m = 0; m < 4; m++) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m].v, bh[m].v, d, 0, 0, 0);
        }
        dd[h] = d;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the next half (and the result tile) overwrite the matrix
      }
    }
    // result tile [term][slot][column] (it reuses the matrix): lane (bn, bk) holds rows 4 bk .. 4 bk + 3 of column bn
    if (bn < 14) {
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          dbuf[((bk >> 1) * kGroup + h * 8 + ((4 * bk + r) & 7)) * kDStride + bn] = dd[h][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef SCORP_BWD_NOTAIL
    if (false) {
#else
    if (lane < nslots) {  // block-frame moments -> the ten screen-space gradients of slot `lane`
#endif
      float *m = dbuf + lane * kDStride;
      const float *m2 = m + kGroup * kDStride;
      const float4 a = gt0[lane];     // x - cx, y - cy, A, B
      const float2 b = gt1[lane];     // C, opacity
      const float opac = b.y;
      const float cA = a.z * (-1.0f / kConicScale), cB = a.w * (-0.5f / kConicScale), cC = b.x * (-1.0f / kConicScale);
      const float xl = a.x, yl = a.y;
      float mm[10];
      {
        const float4 u0 = *reinterpret_cast<const float4 *>(m), u1 = *reinterpret_cast<const float4 *>(m + 4),
                     u2 = *reinterpret_cast<const float4 *>(m + 8), u3 = *reinterpret_cast<const float4 *>(m + 12);
        const float4 w0 = *reinterpret_cast<const float4 *>(m2), w1 = *reinterpret_cast<const float4 *>(m2 + 4),
                     w2 = *reinterpret_cast<const float4 *>(m2 + 8), w3 = *reinterpret_cast<const float4 *>(m2 + 12);
        if constexpr (kExact) {      // rows 0..7 carried v (columns 0..5), rows 8..15 w (columns 6..9)
          mm[0] = u0.x; mm[1] = u0.y; mm[2] = u0.z; mm[3] = u0.w; mm[4] = u1.x; mm[5] = u1.y;
          mm[6] = w1.z; mm[7] = w1.w; mm[8] = w2.x; mm[9] = w2.y;
        } else {                     // first + second term; W sums = (first-term + remainder columns) of the upstream gradients
          const float inv_w = inv_sv * (1.0f / kWScale);
          mm[0] = (u0.x + w0.x) * inv_sv; mm[1] = (u0.y + w0.y) * inv_sv; mm[2] = (u0.z + w0.z) * inv_sv;
          mm[3] = (u0.w + w0.w) * inv_sv; mm[4] = (u1.x + w1.x) * inv_sv; mm[5] = (u1.y + w1.y) * inv_sv;
          mm[6] = ((u1.z + w1.z) + (u2.z + w2.z)) * inv_w; mm[7] = ((u1.w + w1.w) + (u2.w + w2.w)) * inv_w;
          mm[8] = ((u2.x + w2.x) + (u3.x + w3.x)) * inv_w; mm[9] = ((u2.y + w2.y) + (u3.y + w3.y)) * inv_w;
        }
      }
      if constexpr (kColorOnly) {
        *reinterpret_cast<float4 *>(m) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *reinterpret_cast<float4 *>(m + 4) = make_float4(0.0f, 0.0f, mm[6], mm[7]);
        *reinterpret_cast<float2 *>(m + 8) = make_float2(mm[8], 0.0f);
      } else {
        const float m0 = mm[0], mx = mm[1], my = mm[2], mxx = mm[3], mxy = mm[4], myy = mm[5];
        const float svdx = xl * m0 - mx, svdy = yl * m0 - my;
        const float svdx2 = xl * xl * m0 - 2.0f * xl * mx + mxx;
        const float svdxdy = xl * yl * m0 - xl * my - yl * mx + mxy;
        const float svdy2 = yl * yl * m0 - 2.0f * yl * my + myy;
#ifdef SCORP_BWD_RAWMOM
        *reinterpret_cast<float4 *>(m) = make_float4(svdx, svdy, -0.5f * svdx2, -svdxdy);
        *reinterpret_cast<float4 *>(m + 4) = make_float4(-0.5f * svdy2, m0, mm[6], mm[7]);
#else
        *reinterpret_cast<float4 *>(m) = make_float4(0.5f * W * (-cA * svdx - cB * svdy), 0.5f * H * (-cC * svdy - cB * svdx),
                                                     -0.5f * svdx2, -svdxdy);
        *reinterpret_cast<float4 *>(m + 4) = make_float4(-0.5f * svdy2, m0 / opac, mm[6], mm[7]);
#endif
        *reinterpret_cast<float2 *>(m + 8) = make_float2(mm[8], mm[9]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // The 160 sums go out as four wave-wide atomic instructions (lane = float of a 64-byte accumulator row, four rows
    // per instruction, ten lanes of sixteen active: one memory-side request per splat) - but not yet: they are parked in registers and
    // issued after the next chunk's gathers have been picked up (flush_sums).  vmcnt counts loads and atomics alike
    // and the compiler waits with vmcnt(0) for the gathers, so atomics issued right here would be waited for, at
    // their full memory-side latency, at the top of the next chunk.
    static_assert(kAccStride == 16 && kDStride >= 16 && kDStride % 4 == 0, "a result row's first 16 floats map onto an accumulator row lane for lane");
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int sl = 4 * k + (lane >> 4), col = lane & 15;
      park_v[k] = 0.0f;
      if (det) {
        park_o[k] = 0xFFFFFFFFu;
        const uint32_t pair = sl < nslots ? q_id[head + sl] : 0xFFFFFFFFu;   // (det: the staging lane left the pair ordinal here)
        if (pair < capacity && col < 10) {   // (an ordinal beyond the reservation: an overflowed view, discarded anyway)
          const bool used = kColorOnly ? (col >= 6 && col < 9) : true;
          park_v[k] = used ? dbuf[sl * kDStride + col] : 0.0f;
          park_o[k] = (pair * 4u + (uint32_t)quad) * (uint32_t)kAccStride + col;
          if (col == 0) row_flags[pair * 4u + (uint32_t)quad] = 1;
        }
      } else if (sl < nslots && (kColorOnly ? (col >= 6 && col < 9) : col < 10)) {
        park_v[k] = dbuf[sl * kDStride + col];
        park_o[k] = q_id[head + sl] * (uint32_t)kAccStride + col;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // The chunk's gathers are a chain of dependent loads (hit-list entry -> record) of ~1 us each way; with three waves
  // per SIMD nothing hides them, so they are software-pipelined two chunks deep: while chunk c is replayed the records
  // of chunk c+1 and the list entries of chunk c+2 are already in flight.  Chunk c holds the hits at 0-based positions
  // todo - 64 c - 1 (slot 0) down to todo - 64 c - 64 (slot 63): slots ascend back to front.
  const uint32_t *my_hits = hits + (size_t)quad * capacity + beg;
  auto fetch_id = [&](uint32_t c_) {
    const int o = (int)todo - (int)(kChunk * c_) - 1 - lane;
    return o >= 0 ? my_hits[o] : 0xFFFFFFFFu;
  };
  auto fetch_rec = [&](uint32_t id_, float4 &a_, float4 &b_, float4 &c_) {
    if (id_ != 0xFFFFFFFFu) {
      const float4 *src = reinterpret_cast<const float4 *>(rec + id_);
      a_ = src[0]; b_ = src[1]; c_ = src[2];
    }
  };
  float4 a, b, c;
  uint32_t id = fetch_id(0);
  fetch_rec(id, a, b, c);
  uint32_t id1 = fetch_id(1);
  const uint32_t nchunks = (todo + kChunk - 1) / kChunk;
  for (uint32_t ch = 0; ch < nchunks; ch++) {
    float4 a1, b1, c1;
    fetch_rec(id1, a1, b1, c1);
    const uint32_t id2 = fetch_id(ch + 2);
    flush_sums();
    if (id != 0xFFFFFFFFu) {
      uint4 k0, k1, k2;
      splat_block_coefs(a.x, a.y, a.z, a.w, b.x, b.y, cx, cy, k0, k1, k2);
      k0.w = guard_limit_pack(c.w);   // the forward's bits (exp_mfma.hpp)
      q_k[0][lane] = k0; q_k[1][lane] = k1; q_k[2][lane] = k2;
      q_col[lane] = make_float4(b.z, b.w, c.x, c.y);
      q_t0[lane] = make_float4(a.x - cx, a.y - cy, a.z, a.w);
      q_t1[lane] = make_float2(b.x, c.w);
      if constexpr (det) {
        // the (Gaussian, tile) pair's ordinal, Gaussian-major: pair_base[id] + the rank of this tile among the tiles the
        // Gaussian reaches (for_each_tile's order: the set bits of its mask, or its whole rectangle row by row)
        const uint4 raw = reinterpret_cast<const uint4 *>(bin)[id];
        const BinRec br = *reinterpret_cast<const BinRec *>(&raw);
        const uint64_t mk = tile_mask[id];
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const uint32_t rank = mk == kMaskAll ? (uint32_t)((ty - br.y0) * (br.x1 - br.x0) + (tx - br.x0))
                                             : (uint32_t)__builtin_popcountll(mk & ((1ull << ((ty - br.y0) * 8 + (tx - br.x0))) - 1ull));
        q_id[lane] = pair_base[id] + rank;
      } else {
        q_id[lane] = id;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int top = (int)todo - (int)(kChunk * ch);          // 1-based position of slot 0
    const int n = top < kChunk ? top : kChunk;               // hits in this chunk (only the last chunk is short)
    // (b.y = log2(opacity); an indefinite conic takes the same instantiation for its `power > 0` guard)
    const bool hot = __ballot(id != 0xFFFFFFFFu && (b.y > kLog2AlphaMax - 1e-4f || rec_is_indefinite(c.z))) != 0;
    for (int head = 0; head < n; head += kGroup) {
      if (n - head < kGroup) process_group(std::false_type{}, std::true_type{}, n - head, head, top - head);
      else if (hot) process_group(std::true_type{}, std::true_type{}, kGroup, head, top - head);
      else process_group(std::true_type{}, std::false_type{}, kGroup, head, top - head);
    }
    id = id1; a = a1; b = b1; c = c1; id1 = id2;
  }
  flush_sums();
}


// ---------------------------------------------------------------------------------------------------------
// Deterministic mode.  The rows are addressed by the (Gaussian, tile) pair's ordinal in Gaussian-major order, so the rows
// of one Gaussian are CONTIGUOUS - partial[4 * pair_base[i] ... 4 * pair_base[i + 1]) - and the ordered per-Gaussian sum
// is a streaming read with no search (round 3 found a Gaussian's row in every block's depth-sorted hit list by binary
// search: ~190 dependent loads per Gaussian, 2.7 ms of a 3.5 ms view).
//   pair_count_kernel / pair_base_kernel : pair_base[i] = number of (Gaussian, tile) pairs of the Gaussians before i
//                                          (a two-level exclusive scan of the tile counts the binning used)
//   reduce_pair_rows_kernel              : sixteen lanes per Gaussian (lane = float of a row) add the flagged rows in the
//                                          fixed order tiles of the mask x blocks 0..3
// ---------------------------------------------------------------------------------------------------------
constexpr int kScanBlock = kPairScanBlock;   // Gaussians per workgroup of the pair-count scan
__device__ __forceinline__ uint32_t pairs_of(const BinRec &br, uint64_t mask) {
  if ((br.radius & kRadiusMask) == 0) return 0u;
  return mask == kMaskAll ? (uint32_t)((br.x1 - br.x0) * (br.y1 - br.y0)) : (uint32_t)__builtin_popcountll(mask);
}
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *s_red) {   // 256 threads
  for (int off = 32; off >= 1; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
__global__ void __launch_bounds__(256)
pair_count_kernel(int N, const BinRec *__restrict__ bin, const uint64_t *__restrict__ tile_mask, uint32_t *__restrict__ block_sums) {
  __shared__ uint32_t s_red[4];
  uint32_t v = 0;
  for (int k = 0; k < kScanBlock / 256; k++) {
    const int i = blockIdx.x * kScanBlock + k * 256 + threadIdx.x;
    if (i < N) {
      const uint4 raw = reinterpret_cast<const uint4 *>(bin)[i];
      v += pairs_of(*reinterpret_cast<const BinRec *>(&raw), tile_mask[i]);
    }
  }
  v = block_sum_u32(v, s_red);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = v;
}
__global__ void __launch_bounds__(256)
pair_base_kernel(int N, const BinRec *__restrict__ bin, const uint64_t *__restrict__ tile_mask,
                 const uint32_t *__restrict__ block_sums, uint32_t *__restrict__ pair_base) {
  __shared__ uint32_t s_red[4], s_wave[4];
  // the pairs of the workgroups before this one (at most ~1000 words for a million Gaussians: every workgroup adds them itself)
  uint32_t before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += block_sums[b];
  uint32_t run = block_sum_u32(before, s_red);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int k = 0; k < kScanBlock / 256; k++) {
    const int i = blockIdx.x * kScanBlock + k * 256 + threadIdx.x;
    uint32_t c = 0;
    if (i < N) {
      const uint4 raw = reinterpret_cast<const uint4 *>(bin)[i];
      c = pairs_of(*reinterpret_cast<const BinRec *>(&raw), tile_mask[i]);
    }
    uint32_t inc = c;   // inclusive prefix inside the wave
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
      if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wv; w++) wbase += s_wave[w];
    if (i < N) pair_base[i] = run + wbase + inc - c;
    run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) pair_base[N] = run;
}

__global__ void __launch_bounds__(256)
reduce_pair_rows_kernel(int N, const uint32_t *__restrict__ pair_base, uint32_t capacity, const uint8_t *__restrict__ row_flags,
                        const float *__restrict__ partial, float *__restrict__ acc) {
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4), col = threadIdx.x & 15;
  if (i >= N) return;
  const uint32_t r0 = min(pair_base[i], capacity) * 4u, r1 = min(pair_base[i + 1], capacity) * 4u;
  float sum = 0.0f;
  // four (Gaussian, tile) pairs = sixteen rows per step: the four flag words first, then every flagged row, all loads in
  // flight together (a Gaussian has 2.4 pairs on average: one step); the additions keep the fixed order pair, block
  for (uint32_t r = r0; r < r1; r += 16) {
    uint32_t f[4];
#pragma unroll
    for (int p = 0; p < 4; p++) f[p] = r + 4 * p < r1 ? *reinterpret_cast<const uint32_t *>(row_flags + r + 4 * p) : 0u;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const bool on = (f[k >> 2] >> (8 * (k & 3))) & 0xFFu;
      v[k] = on ? partial[(size_t)(r + k) * kAccStride + col] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) sum += v[k];   // (an absent row adds an exact zero)
  }
  acc[(size_t)i * kAccStride + col] = col < 10 ? sum : 0.0f;
}

}  // namespace
}  // namespace scorp

// pair_base[i] = the number of (Gaussian, tile) pairs of the Gaussians before i (pair_base[N] = all of them), from the tile
// rectangles / masks the binning used; shared by the 3-D and the 2-D deterministic backward (the 2-D state holds the same
// BinRec / tile-mask arrays)
int scorp::launch_pair_base(int N, const BinRec *bin, const uint64_t *tile_mask, uint32_t *block_sums, uint32_t *pair_base,
                            hipStream_t stream) {
  const int blocks = (N + kScanBlock - 1) / kScanBlock;
  pair_count_kernel<<<blocks, 256, 0, stream>>>(N, bin, tile_mask, block_sums);
  pair_base_kernel<<<blocks, 256, 0, stream>>>(N, bin, tile_mask, block_sums, pair_base);
  return SCORP_OK;
}

using namespace scorp;

extern "C" size_t scorp_gs3d_backward_scratch_bytes_ex(int32_t N, int32_t W, int32_t H, uint64_t capacity, uint32_t flags) {
  (void)W; (void)H;
  if (flags & SCORP_BACKWARD_DETERMINISTIC) return DetLayout(N, capacity, kAccStride).total;
  return align_up((size_t)(N > 0 ? N : 1) * kAccStride * sizeof(float), 256);
}

extern "C" size_t scorp_gs3d_backward_scratch_bytes(int32_t N) {
  return align_up((size_t)(N > 0 ? N : 1) * kAccStride * sizeof(float), 256);
}

extern "C" int scorp_gs3d_backward(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                                   const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                                   const ScorpGs3dGrads *grads, void *scratch, size_t scratch_bytes,
                                   scorp_stream_t stream_) {
  return scorp_gs3d_backward_ex(in, state, pairs, capacity, dL_dcolor, dL_ddepth, dL_dalpha, grads, scratch, scratch_bytes,
                                0u, stream_);
}

extern "C" int scorp_gs3d_backward_ex(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                                      const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                                      const ScorpGs3dGrads *grads, void *scratch, size_t scratch_bytes, uint32_t flags,
                                      scorp_stream_t stream_) {
  return backward3d_impl(in, state, pairs, capacity, dL_dcolor, dL_ddepth, dL_dalpha, grads, scratch, scratch_bytes, flags, stream_,
                         nullptr);
}

// `adam` (scorp_gs3d_train_view): the per-Gaussian kernel applies the optimizer step and the view's statistics itself
int scorp::backward3d_impl(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                           const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                           const ScorpGs3dGrads *grads, void *scratch, size_t scratch_bytes, uint32_t flags,
                           scorp_stream_t stream_, const AdamEpi *adam) {
  if (!in || !state || !pairs || !grads || !scratch) { set_error("NULL argument to scorp_gs3d_backward"); return SCORP_ERR_INVALID; }
  if (!dL_dcolor) { set_error("dL_dcolor is NULL"); return SCORP_ERR_INVALID; }
  if (in->num_views > 1) { set_error("num_views > 1 is forward only"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const int N = in->num_gaussians, W = in->image_width, H = in->image_height;
  if (N <= 0) return SCORP_OK;
  if ((uint64_t)N * kAccStride > 0xFFFFFFFFull) {   // the blend kernel addresses accumulators with 32-bit float offsets
    set_error("num_gaussians %d above the %llu the backward supports", N, 0xFFFFFFFFull / kAccStride);
    return SCORP_ERR_INVALID;
  }
  const StateLayout L(N, W, H);
  const PairLayout P(capacity);
  const bool det = (flags & SCORP_BACKWARD_DETERMINISTIC) != 0;
  const size_t need = scorp_gs3d_backward_scratch_bytes_ex(N, W, H, capacity, flags);
  if (det && capacity * 4 * kAccStride > 0xFFFFFFFFull) { set_error("capacity too large for the deterministic backward"); return SCORP_ERR_INVALID; }
  if (scratch_bytes < need || ((uintptr_t)scratch & 15)) {
    set_error("backward scratch too small or misaligned (%zu < %zu)", scratch_bytes, need);
    return SCORP_ERR_INVALID;
  }
  const char *base = (const char *)state, *pb = (const char *)pairs;
  float *acc = (float *)scratch;
  float *partial = nullptr;
  uint8_t *row_flags = nullptr;
  uint32_t *pair_base = nullptr;
  const DetLayout DL(N, capacity, kAccStride);
  if (det) {
    char *p = (char *)scratch;
    partial = (float *)(p + DL.partial);
    row_flags = (uint8_t *)(p + DL.flags);
    pair_base = (uint32_t *)(p + DL.pair_base);
    uint32_t *block_sums = (uint32_t *)(p + DL.block_sums);
    SCORP_HIP_CHECK(hipMemsetAsync(row_flags, 0, (size_t)(capacity > 0 ? capacity : 1) * 4, stream));
    launch_pair_base(N, (const BinRec *)(base + L.bin), (const uint64_t *)(base + L.tile_mask), block_sums, pair_base, stream);
    SCORP_KERNEL_CHECK("pair_base", in->debug, stream);
  } else if (!(flags & SCORP_BACKWARD_SCRATCH_ZEROED)) {
    SCORP_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)N * kAccStride * sizeof(float), stream));
  }
  {
    ProfScope prof(kKBlendBackward, stream);
    const int blocks = ((L.tiles + 7) / 8) * 8 * 4;
    const bool da = dL_ddepth || dL_dalpha, exact = (flags & SCORP_BACKWARD_EXACT_FP32) != 0;
    // nothing but colour gradients wanted (every geometry / opacity output NULL): the colour-only replay
    const bool adam_geom = adam && adam->on && (adam->m[0] || adam->m[3] || adam->m[4] || adam->m[5] || adam->accum);
    const bool color_only = !exact && !adam_geom && !grads->means3D && !grads->means2D && !grads->opacities && !grads->scales &&
                            !grads->rotations && !grads->cov3D_precomp;
    auto wk = det ? (color_only ? blend_backward_wave_kernel<false, false, true, true>
                     : exact ? (da ? blend_backward_wave_kernel<true, true, false, true> : blend_backward_wave_kernel<false, true, false, true>)
                             : (da ? blend_backward_wave_kernel<true, false, false, true> : blend_backward_wave_kernel<false, false, false, true>))
              : color_only ? blend_backward_wave_kernel<false, false, true>
              : exact ? (da ? blend_backward_wave_kernel<true, true> : blend_backward_wave_kernel<false, true>)
                      : (da ? blend_backward_wave_kernel<true, false> : blend_backward_wave_kernel<false, false>);
    wk<<<blocks, 64, 0, stream>>>(
        (const uint32_t *)(base + L.tile_start), (const uint32_t *)(pb + P.hits), (const SplatRec *)(base + L.rec),
        (uint32_t)capacity, W, H, L.tiles_x, L.tiles, in->bg, (const float *)(base + L.final_T),
        (const uint32_t *)(base + L.n_contrib), dL_dcolor, dL_ddepth, dL_dalpha, acc, partial, row_flags, pair_base,
        (const BinRec *)(base + L.bin), (const uint64_t *)(base + L.tile_mask));
  }
  SCORP_KERNEL_CHECK("blend_backward", in->debug, stream);
  if (det) {
    reduce_pair_rows_kernel<<<(N + 15) / 16, 256, 0, stream>>>(N, pair_base, (uint32_t)capacity, row_flags, partial, acc);
    SCORP_KERNEL_CHECK("reduce_pair_rows", in->debug, stream);
  }
  {
    ProfScope prof(kKPreprocessBackward, stream);
    launch_preprocess_backward(in, L, (const BinRec *)(base + L.bin), acc, grads, stream, adam);
  }
  SCORP_KERNEL_CHECK("preprocess_backward", in->debug, stream);
  return SCORP_OK;
}
