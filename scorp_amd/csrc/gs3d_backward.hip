// gs3d_backward.hip — 3DGS backward for gfx950: per-tile back-to-front replay -> per-Gaussian chain rule.
//
// Replaces the backward half of `diff_gaussian_rasterization` (reached from `loss.backward()`,
// train_3dgs.py:152; colour-only at post_refine_gs.py:53-56; w.r.t. colors_precomp at utils/mask.py:47-70).
// Arithmetic follows oracle/gs3d_oracle.c::gs3d_oracle_backward.  The forward state and the pair buffer are
// only read, so the backward can be replayed on one forward.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace scorp {
namespace {

// Sum over the 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------
// B1: per-tile replay, back to front. Same tiling as the forward blend (4 wave64 x 8x8 pixels, LDS batches of
// 256 records, per-wave ballot cull). Each lane produces 10 partial gradients per splat; they are summed over
// the wave and added to the per-Gaussian accumulator with float atomics (one add per wave per value).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void
blend_backward_body(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                      const SplatRec *__restrict__ rec, uint32_t capacity, int W, int H, int tiles_x,
                      const float *__restrict__ bg, const float *__restrict__ final_T,
                      const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dcolor,
                      const float *__restrict__ dL_ddepth, const float *__restrict__ dL_dalpha,
                      float *__restrict__ acc) {
  __shared__ float4 s_a[256], s_b[256], s_c[256];
  __shared__ uint32_t s_id[256];
  __shared__ uint32_t s_max;
  const int tile = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bx = (tile % tiles_x) * kTile + (wave & 1) * 8, by = (tile / tiles_x) * kTile + (wave >> 1) * 8;
  const int px = bx + (lane & 7), py = by + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const float bx0 = (float)bx, bx1 = (float)(bx + 7), by0 = (float)by, by1 = (float)(by + 7);
  const uint32_t beg = min(tile_start[tile], capacity), end = min(tile_start[tile + 1], capacity);
  if (end == beg) return;
  const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
  const float T_final = inside ? final_T[pix] : 0.0f;
  const uint32_t last = inside ? n_contrib[pix] : 0u;
  float dpix0 = 0.0f, dpix1 = 0.0f, dpix2 = 0.0f, ddep = 0.0f, dalp = 0.0f;
  if (inside) {
    dpix0 = dL_dcolor[pix]; dpix1 = dL_dcolor[HW + pix]; dpix2 = dL_dcolor[2 * HW + pix];
    if (dL_ddepth) ddep = dL_ddepth[pix];
    if (dL_dalpha) dalp = dL_dalpha[pix];
  }
  const float bg_dot = bg[0] * dpix0 + bg[1] * dpix1 + bg[2] * dpix2;
  // Nothing past the deepest contributor of any pixel of the tile was blended: start the replay there.
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  atomicMax(&s_max, last);
  __syncthreads();
  const uint32_t todo = s_max;  // 1-based count of list entries to replay
  float T = T_final;
  float acc_c0 = 0.0f, acc_c1 = 0.0f, acc_c2 = 0.0f, acc_d = 0.0f, acc_a = 0.0f;
  float last_alpha = 0.0f, last_c0 = 0.0f, last_c1 = 0.0f, last_c2 = 0.0f, last_d = 0.0f;
  // Batches walk the list from position todo-1 down to 0; slot s of a batch holds list position top-s.
  for (uint32_t done_n = 0; done_n < todo; done_n += 256) {
    __syncthreads();
    const uint32_t top = todo - 1 - done_n;  // list position (0-based) held by slot 0
    const int cnt = (int)min(256u, todo - done_n);
    if ((int)threadIdx.x < cnt) {
      const uint32_t id = point_list[beg + top - threadIdx.x];
      const float4 *src = reinterpret_cast<const float4 *>(rec + id);
      s_a[threadIdx.x] = src[0];
      s_b[threadIdx.x] = src[1];
      s_c[threadIdx.x] = src[2];
      s_id[threadIdx.x] = id;
    }
    __syncthreads();
    for (int q = 0; q < cnt; q += 64) {
      const int j = q + lane;
      bool hit = false;
      if (j < cnt) {
        const float4 a = s_a[j];
        hit = conic_min_over_box(a.x, a.y, a.z, a.w, s_b[j].x, bx0, bx1, by0, by1) <= s_c[j].z;
      }
      uint64_t mask = __ballot(hit);
      while (mask) {
        const int jj = q + __builtin_ctzll(mask);
        mask &= mask - 1;
        const uint32_t pos1 = top - (uint32_t)jj + 1u;  // 1-based list position of this splat
        const float4 a = s_a[jj], b = s_b[jj], c = s_c[jj];
        const float dx = a.x - pxf, dy = a.y - pyf;
        const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
        const float G = __expf(power);
        const float alpha = fminf(kAlphaMax, b.y * G);
        const bool valid = pos1 <= last && power <= 0.0f && alpha >= kAlphaMin;
        if (__ballot(valid) == 0) continue;
        float g_x = 0.0f, g_y = 0.0f, g_A = 0.0f, g_B = 0.0f, g_C = 0.0f, g_o = 0.0f, g_r = 0.0f, g_g = 0.0f,
              g_b = 0.0f, g_z = 0.0f;
        if (valid) {
          T = T / (1.0f - alpha);
          const float w = alpha * T;
          float dL_dal = 0.0f;
          acc_c0 = last_alpha * last_c0 + (1.0f - last_alpha) * acc_c0; last_c0 = b.z;
          acc_c1 = last_alpha * last_c1 + (1.0f - last_alpha) * acc_c1; last_c1 = b.w;
          acc_c2 = last_alpha * last_c2 + (1.0f - last_alpha) * acc_c2; last_c2 = c.x;
          dL_dal += (b.z - acc_c0) * dpix0 + (b.w - acc_c1) * dpix1 + (c.x - acc_c2) * dpix2;
          g_r = w * dpix0; g_g = w * dpix1; g_b = w * dpix2;
          acc_d = last_alpha * last_d + (1.0f - last_alpha) * acc_d; last_d = c.y;
          dL_dal += (c.y - acc_d) * ddep;
          g_z = w * ddep;
          acc_a = last_alpha + (1.0f - last_alpha) * acc_a;
          dL_dal += (1.0f - acc_a) * dalp;
          dL_dal *= T;
          last_alpha = alpha;
          dL_dal += (-T_final / (1.0f - alpha)) * bg_dot;
          const float dL_dG = b.y * dL_dal;
          const float gdx = G * dx, gdy = G * dy;
          g_x = dL_dG * (-gdx * a.z - gdy * a.w) * (0.5f * W);
          g_y = dL_dG * (-gdy * b.x - gdx * a.w) * (0.5f * H);
          g_A = -0.5f * gdx * dx * dL_dG;
          g_B = -gdx * dy * dL_dG;
          g_C = -0.5f * gdy * dy * dL_dG;
          g_o = G * dL_dal;
        }
        g_x = wave_sum(g_x); g_y = wave_sum(g_y); g_A = wave_sum(g_A); g_B = wave_sum(g_B); g_C = wave_sum(g_C);
        g_o = wave_sum(g_o); g_r = wave_sum(g_r); g_g = wave_sum(g_g); g_b = wave_sum(g_b); g_z = wave_sum(g_z);
        // lanes 0..9 add the ten sums to ten consecutive floats of the splat's accumulator row
        float v = g_x;
        v = lane == 1 ? g_y : v; v = lane == 2 ? g_A : v; v = lane == 3 ? g_B : v; v = lane == 4 ? g_C : v;
        v = lane == 5 ? g_o : v; v = lane == 6 ? g_r : v; v = lane == 7 ? g_g : v; v = lane == 8 ? g_b : v;
        v = lane == 9 ? g_z : v;
        if (lane < 10) atomicAdd(acc + (size_t)s_id[jj] * kAccStride + lane, v);
      }
    }
  }
}

__global__ void __launch_bounds__(256)
blend_backward_kernel_abl(const uint32_t *tile_start, const uint32_t *point_list, const SplatRec *rec, uint32_t capacity,
                          int W, int H, int tiles_x, const float *bg, const float *final_T, const uint32_t *n_contrib,
                          const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha, float *acc, int) {
  // same signature as the MFMA kernel so the launcher can switch between them
  blend_backward_body(tile_start, point_list, rec, capacity, W, H, tiles_x, bg, final_T, n_contrib, dL_dcolor, dL_ddepth,
                      dL_dalpha, acc);
}

// ---------------------------------------------------------------------------------------------------------
// B1m: the same replay with the pixel->splat reduction done on the matrix cores.
//
// For one splat the ten sums over pixels factor as  sum_p v_p * {1, x_p, y_p, x_p^2, x_p y_p, y_p^2}  (geometry:
// v_p = G * opacity * dL/dalpha, pixel coordinates relative to the tile centre) and  sum_p w_p * {dL/dr, dL/dg, dL/db,
// dL/ddepth}_p  (w_p = alpha * T).  With a wave's 64 pixels as the K dimension that is D[16 splats][16] =
// [V | W](16 x 128) * [basis_v ; basis_w](128 x 16): 32 exact-fp32 v_mfma_f32_16x16x4_f32 per 16 splats instead of
// 60 cross-lane shuffles per splat.
//
// Per wave and 16-splat group:
//   1a (straight-line, no dependence between splats): alpha and G*opacity of the 16 splats at this lane's pixel;
//   1b (the sequential part): T, the blended-behind recurrence and dL/dalpha.  A splat that does not contribute to
//      this pixel is carried through as alpha = 0, which leaves every recurrence bit-for-bit unchanged, so the
//      group is branch-free.  The five "accumulated colour/depth/alpha behind" recurrences of the textbook form
//      collapse into ONE, because only their dot product with this pixel's upstream gradient is ever used:
//      s = c.dL/dcolor + z*dL/ddepth + dL/dalpha,  R <- last_alpha * s_last + (1 - last_alpha) * R,
//      dL/dalpha_i = (s - R) * T - T_final / (1 - alpha) * (bg . dL/dcolor);
//   lane (= pixel) writes v, w into a wave-private LDS matrix [slot][pixel] (row stride 65 floats: the row writes and
//   the transposed A-operand reads are both conflict-free); the per-pixel basis is loop-invariant and lives in 32
//   registers as the B operand;
//   the 16x16 MFMA result is added to per-tile LDS accumulators (one row per splat of the batch).
// After the batch the rows are turned from tile-frame moments into the ten gradients and flushed with float atomics
// shaped as whole 40-byte row segments (lanes = consecutive floats).
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kXStride = 65;                 // floats per slot row of the wave-private v / w matrices
constexpr int kGroup = 16;                   // splats per MFMA group
constexpr float kLog2e = 1.4426950408889634f;

template <int kBatch>
__global__ void __launch_bounds__(256, 3)
blend_backward_mfma_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                           const SplatRec *__restrict__ rec, uint32_t capacity, int W, int H, int tiles_x,
                           const float *__restrict__ bg, const float *__restrict__ final_T,
                           const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dcolor,
                           const float *__restrict__ dL_ddepth, const float *__restrict__ dL_dalpha,
                           float *__restrict__ acc, int ablate) {
  // s_a = (x, y, A', B'), s_b = (C', opacity, r, g), s_c = (b, depth, kcut, -): conic pre-scaled for exp2
  __shared__ float4 s_a[kBatch], s_b[kBatch], s_c[kBatch];
  __shared__ uint32_t s_id[kBatch];
  __shared__ float s_acc[kBatch * kAccStride];
  __shared__ uint32_t s_touched[kBatch];
  __shared__ float s_xv[4][kGroup * kXStride], s_xw[4][kGroup * kXStride];
  __shared__ uint16_t s_list[4][kBatch + kGroup];
  __shared__ uint32_t s_max;
  const int tile = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tx0 = (tile % tiles_x) * kTile, ty0 = (tile / tiles_x) * kTile;
  const int bx = tx0 + (wave & 1) * 8, by = ty0 + (wave >> 1) * 8;
  const int px = bx + (lane & 7), py = by + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const float bx0 = (float)bx, bx1 = (float)(bx + 7), by0 = (float)by, by1 = (float)(by + 7);
  const float cx = (float)tx0 + 7.5f, cy = (float)ty0 + 7.5f;  // tile-frame origin
  const uint32_t beg = min(tile_start[tile], capacity), end = min(tile_start[tile + 1], capacity);
  if (end == beg) return;
  const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
  const float T_final = inside ? final_T[pix] : 0.0f;
  const uint32_t last = inside ? n_contrib[pix] : 0u;
  float dpix0 = 0.0f, dpix1 = 0.0f, dpix2 = 0.0f, ddep = 0.0f, dalp = 0.0f;
  if (last > 0) {  // a pixel nothing was blended into never reads its upstream gradient (it may hold NaN: 0/0 of depth/alpha)
    dpix0 = dL_dcolor[pix]; dpix1 = dL_dcolor[HW + pix]; dpix2 = dL_dcolor[2 * HW + pix];
    if (dL_ddepth) ddep = dL_ddepth[pix];
    if (dL_dalpha) dalp = dL_dalpha[pix];
  }
  const float tf_bg = T_final * (bg[0] * dpix0 + bg[1] * dpix1 + bg[2] * dpix2);
  // B operand of the MFMAs: lane (n = lane&15, k = lane>>4) holds basis[pixel q = t + 16k][n] for K-step t
  const int bn = lane & 15, bk = lane >> 4;
  float bv[16], bw[16];
#pragma unroll
  for (int t = 0; t < 16; t++) {
    const int q = t + 16 * bk;
    const int qx = bx + (q & 7), qy = by + (q >> 3);
    const float xl = (float)qx - cx, yl = (float)qy - cy;
    float v = 0.0f;
    v = bn == 0 ? 1.0f : v; v = bn == 1 ? xl : v; v = bn == 2 ? yl : v;
    v = bn == 3 ? xl * xl : v; v = bn == 4 ? xl * yl : v; v = bn == 5 ? yl * yl : v;
    bv[t] = v;
    float w = 0.0f;
    if (qx < W && qy < H && bn >= 6 && bn <= 9) {
      const size_t qp = (size_t)qy * W + qx;
      if (n_contrib[qp] > 0) {  // same guard as above: 0 * NaN would poison the matrix product
        if (bn <= 8) w = dL_dcolor[(size_t)(bn - 6) * HW + qp];
        else w = dL_ddepth ? dL_ddepth[qp] : 0.0f;
      }
    }
    bw[t] = w;
  }
  if (threadIdx.x == 0) s_max = 0;
  for (int e = threadIdx.x; e < kBatch * kAccStride; e += 256) s_acc[e] = 0.0f;
  if ((int)threadIdx.x < kBatch) s_touched[threadIdx.x] = 0;
  __syncthreads();
  atomicMax(&s_max, last);
  __syncthreads();
  const uint32_t todo = s_max;
  float T = T_final, R = 0.0f, s_last = 0.0f, last_alpha = 0.0f;
  float *xv = s_xv[wave], *xw = s_xw[wave];
  uint16_t *list = s_list[wave];
  const int abase = (lane & 15) * kXStride + 16 * (lane >> 4);

  for (uint32_t done_n = 0; done_n < todo; done_n += kBatch) {
    const uint32_t top = todo - 1 - done_n;  // list position (0-based) held by batch slot 0
    const int cnt = (int)min((uint32_t)kBatch, todo - done_n);
    if ((int)threadIdx.x < cnt) {
      const uint32_t id = point_list[beg + top - threadIdx.x];
      const float4 *src = reinterpret_cast<const float4 *>(rec + id);
      const float4 a = src[0], b = src[1], c = src[2];
      s_a[threadIdx.x] = make_float4(a.x, a.y, -0.5f * kLog2e * a.z, -kLog2e * a.w);
      s_b[threadIdx.x] = make_float4(-0.5f * kLog2e * b.x, b.y, b.z, b.w);
      s_c[threadIdx.x] = c;
      s_id[threadIdx.x] = id;
    }
    __syncthreads();
    // compact the batch slots whose alpha >= 1/255 disc reaches this wave's 8x8 block
    int nh = 0;
    for (int q = 0; q < cnt; q += 64) {
      const int j = q + lane;
      bool hit = false;
      if (j < cnt) {
        const float4 a = s_a[j];  // conic is pre-scaled for exp2: undo it for the cull test
        hit = conic_min_over_box(a.x, a.y, a.z * (-2.0f / kLog2e), a.w * (-1.0f / kLog2e), s_b[j].x * (-2.0f / kLog2e), bx0,
                                 bx1, by0, by1) <= s_c[j].z;
      }
      const uint64_t m = __ballot(hit);
      if (hit) list[nh + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint16_t)j;
      nh += __builtin_popcountll(m);
    }
    if (lane < kGroup) list[nh + lane] = 0;  // padding entries of the last group point at a real slot
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (int g0 = 0; g0 < nh; g0 += kGroup) {
      const int nslots = min(kGroup, nh - g0);
      float Go[kGroup], al[kGroup];
      uint32_t jjs[kGroup];
      bool any = false;
      // ---- 1a: independent per splat ----
#pragma unroll
      for (int i = 0; i < kGroup; i++) {
        const uint32_t jj = __builtin_amdgcn_readfirstlane((uint32_t)list[g0 + i]);
        jjs[i] = jj;
        const float4 a = s_a[jj];
        const float2 co = *reinterpret_cast<const float2 *>(&s_b[jj]);
        const float dx = a.x - pxf, dy = a.y - pyf;
        const float p2 = a.z * dx * dx + co.x * dy * dy + a.w * dx * dy;  // log2 of the Gaussian falloff
        const float G = __builtin_amdgcn_exp2f(p2);
        const float alpha = fminf(kAlphaMax, co.y * G);
        const bool ok = i < nslots && (top - jj + 1u) <= last && p2 <= 0.0f && alpha >= kAlphaMin;
        al[i] = ok ? alpha : 0.0f;
        Go[i] = ok ? G * co.y : 0.0f;
        any |= ok;
      }
      if (__ballot(any) == 0 || (ablate & 2)) continue;  // nothing of this group reaches any pixel of the block
      // ---- 1b: the sequential recurrence, branch-free (alpha = 0 is an exact no-op) ----
#pragma unroll
      for (int i = 0; i < kGroup; i++) {
        const uint32_t jj = jjs[i];
        const float2 rg = *reinterpret_cast<const float2 *>(&s_b[jj].z);
        const float2 bz = *reinterpret_cast<const float2 *>(&s_c[jj]);
        const float alpha = al[i];
        const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
        T *= rinv;
        const float w = alpha * T;
        R = last_alpha * (s_last - R) + R;
        const float sc = rg.x * dpix0 + rg.y * dpix1 + bz.x * dpix2 + bz.y * ddep + dalp;
        const float dL_dal = (sc - R) * T - tf_bg * rinv;
        s_last = sc;
        last_alpha = alpha;
        xv[i * kXStride + lane] = Go[i] * dL_dal;
        xw[i * kXStride + lane] = w;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (ablate & 4) continue;
      // ---- MFMA: D[slot][n] = sum over the 64 pixels ----
      f32x4 d0 = {0.0f, 0.0f, 0.0f, 0.0f}, d1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[abase + t], bv[t], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[abase + t + 1], bv[t + 1], d1, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[abase + t], bw[t], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[abase + t + 1], bw[t + 1], d1, 0, 0, 0);
      }
      const f32x4 d = d0 + d1;
      // D: this lane holds column n = lane&15 of rows (slots) 4*(lane>>4) + r
      if (bn < 10) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int sl = 4 * bk + r;
          if (sl < nslots) {
            const uint32_t jj = list[g0 + sl];
            atomicAdd(&s_acc[jj * kAccStride + bn], d[r]);
            if (bn == 0) s_touched[jj] = 1u;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // moments (tile frame) -> the ten screen-space gradients, in place
    if ((int)threadIdx.x < cnt && s_touched[threadIdx.x]) {
      float *m = s_acc + threadIdx.x * kAccStride;
      const float4 a = s_a[threadIdx.x], b = s_b[threadIdx.x];
      const float cA = a.z * (-2.0f / kLog2e), cB = a.w * (-1.0f / kLog2e), cC = b.x * (-2.0f / kLog2e);
      const float xl = a.x - cx, yl = a.y - cy;
      const float m0 = m[0], mx = m[1], my = m[2], mxx = m[3], mxy = m[4], myy = m[5];
      const float svdx = xl * m0 - mx, svdy = yl * m0 - my;
      const float svdx2 = xl * xl * m0 - 2.0f * xl * mx + mxx;
      const float svdxdy = xl * yl * m0 - xl * my - yl * mx + mxy;
      const float svdy2 = yl * yl * m0 - 2.0f * yl * my + myy;
      m[0] = 0.5f * W * (-cA * svdx - cB * svdy);
      m[1] = 0.5f * H * (-cC * svdy - cB * svdx);
      m[2] = -0.5f * svdx2;
      m[3] = -svdxdy;
      m[4] = -0.5f * svdy2;
      m[5] = m0 / b.y;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * kAccStride; e += 256) {
      const int row = e / kAccStride, col = e - row * kAccStride;
      if (col < 10 && s_touched[row] && !(ablate & 1)) atomicAdd(acc + (size_t)s_id[row] * kAccStride + col, s_acc[e]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * kAccStride; e += 256) s_acc[e] = 0.0f;
    if ((int)threadIdx.x < kBatch) s_touched[threadIdx.x] = 0;
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------------------
// B1w: the MFMA replay with ONE WAVE PER 8x8 PIXEL BLOCK as the unit of work (64-thread workgroups, no workgroup
// barriers, no waiting for the slowest wave of a tile).  A wave walks its tile's sorted list back to front 32 entries
// at a time: each lane gathers one record and runs the exact ellipse-vs-block test on it, the survivors are
// compacted (ballot + popcount) into a 64-entry ring in the wave's own LDS, and whenever 16 are queued they go
// through the 1a / 1b / MFMA pipeline of the kernel above.  The 16x16 result is converted from block-frame moments
// to the ten gradients by lanes 0..15 and added to the per-Gaussian accumulators with float atomics (zero terms are
// skipped).  Blocks are numbered so that the four waves of a tile land on the same XCD (same L2).
// ---------------------------------------------------------------------------------------------------------
constexpr int kRing = 64, kChunk = 32;

template <bool kHasDA>  // false: no upstream gradient on the depth / alpha images (the photometric-loss-only step)
__global__ void __launch_bounds__(64, 3)
blend_backward_wave_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                           const SplatRec *__restrict__ rec, uint32_t capacity, int W, int H, int tiles_x, int tiles,
                           const float *__restrict__ bg, const float *__restrict__ final_T,
                           const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dcolor,
                           const float *__restrict__ dL_ddepth, const float *__restrict__ dL_dalpha,
                           float *__restrict__ acc) {
  // ring entries: (x, y, A', B'), (C', log2 opacity, r, g), (b, depth, opacity, -): conic pre-scaled so that
  // opacity * G = exp2(A' dx^2 + C' dy^2 + B' dx dy + log2 opacity)
  __shared__ float4 q_a[kRing], q_b[kRing], q_c[kRing];
  __shared__ __attribute__((aligned(16))) uint32_t q_id[kRing], q_pos[kRing];
  __shared__ float xv[kGroup * kXStride], xw[kGroup * kXStride];
  float *dbuf = xw;  // the 16x12 result tile reuses the w matrix once the MFMAs have consumed it
  const int lane = threadIdx.x;
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int tile = (kk >> 2) * 8 + xcd, quad = kk & 3;
  if (tile >= tiles) return;
  const int tx0 = (tile % tiles_x) * kTile, ty0 = (tile / tiles_x) * kTile;
  const int bx = tx0 + (quad & 1) * 8, by = ty0 + (quad >> 1) * 8;
  const int px = bx + (lane & 7), py = by + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const float bx0 = (float)bx, bx1 = (float)(bx + 7), by0 = (float)by, by1 = (float)(by + 7);
  const float cx = (float)bx + 3.5f, cy = (float)by + 3.5f;  // block-frame origin of the moments
  const uint32_t beg = min(tile_start[tile], capacity), end = min(tile_start[tile + 1], capacity);
  if (end == beg) return;
  const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
  const float T_final = inside ? final_T[pix] : 0.0f;
  const uint32_t last = inside ? n_contrib[pix] : 0u;
  float dpix0 = 0.0f, dpix1 = 0.0f, dpix2 = 0.0f, ddep = 0.0f, dalp = 0.0f;
  if (last > 0) {
    dpix0 = dL_dcolor[pix]; dpix1 = dL_dcolor[HW + pix]; dpix2 = dL_dcolor[2 * HW + pix];
    if (kHasDA && dL_ddepth) ddep = dL_ddepth[pix];
    if (kHasDA && dL_dalpha) dalp = dL_dalpha[pix];
  }
  const float tf_bg = T_final * (bg[0] * dpix0 + bg[1] * dpix1 + bg[2] * dpix2);
  uint32_t todo = last;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) todo = max(todo, (uint32_t)__shfl_xor((int)todo, off, 64));
  todo = (uint32_t)__builtin_amdgcn_readfirstlane((int)todo);   // tell the compiler it is wave-uniform: the chunk loop,
  if (todo == 0) return;                                        // ring head / count and slot indices then live in SGPRs
  const int bn = lane & 15, bk = lane >> 4;
  float bv[16], bw[16];
#pragma unroll
  for (int t = 0; t < 16; t++) {
    const int q = t + 16 * bk;
    const int qx = bx + (q & 7), qy = by + (q >> 3);
    const float xl = (float)qx - cx, yl = (float)qy - cy;
    float v = 0.0f;
    v = bn == 0 ? 1.0f : v; v = bn == 1 ? xl : v; v = bn == 2 ? yl : v;
    v = bn == 3 ? xl * xl : v; v = bn == 4 ? xl * yl : v; v = bn == 5 ? yl * yl : v;
    bv[t] = v;
    float w = 0.0f;
    if (qx < W && qy < H && bn >= 6 && bn <= 9) {
      const size_t qp = (size_t)qy * W + qx;
      if (n_contrib[qp] > 0) {
        if (bn <= 8) w = dL_dcolor[(size_t)(bn - 6) * HW + qp];
        else w = (kHasDA && dL_ddepth) ? dL_ddepth[qp] : 0.0f;
      }
    }
    bw[t] = w;
  }
  float T = T_final, R = 0.0f, s_last = 0.0f, last_alpha = 0.0f;
  const int abase = (lane & 15) * kXStride + 16 * (lane >> 4);
  int head = 0, count = 0;

  // head stays a multiple of kGroup (only a wave's final group is partial), so the slots of a group are head + i
  // without wrap-around: one LDS base per array, immediate offsets
  auto process_group = [&](auto full, int nslots) {
    constexpr bool kFull = decltype(full)::value;   // full groups run straight-line; only a wave's last one is partial
    {
      int hv = head;
      asm volatile("" : "+v"(hv));   // keep the group's LDS bases in VGPRs (else every ds_read re-moves an SGPR base)
      const float4 *ga = q_a + hv, *gb = q_b + hv, *gc = q_c + hv;
      const uint32_t *gp = q_pos + hv;
      // 1a + 1b in two halves of 8 splats: the straight-line part keeps only 8 (alpha, G*opacity) pairs live
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float Go[8], al[8];
        const uint4 pl = *reinterpret_cast<const uint4 *>(gp + h * 8), ph = *reinterpret_cast<const uint4 *>(gp + h * 8 + 4);
        const uint32_t pos8[8] = {pl.x, pl.y, pl.z, pl.w, ph.x, ph.y, ph.z, ph.w};
#pragma unroll
        for (int i8 = 0; i8 < 8; i8++) {
          const int i = h * 8 + i8;
          const float4 a = ga[i];
          const float2 co = *reinterpret_cast<const float2 *>(&gb[i]);
          const uint32_t pos = pos8[i8];
          const float dx = a.x - pxf, dy = a.y - pyf;
          const float e = co.y + a.z * dx * dx + co.x * dy * dy + a.w * dx * dy;   // log2(opacity * G)
          const float g_o = __builtin_amdgcn_exp2f(e);
          const float alpha = fminf(kAlphaMax, g_o);
          const bool ok = (kFull || i < nslots) & (pos <= last) & (e <= co.y) & (alpha >= kAlphaMin);
          al[i8] = ok ? alpha : 0.0f;
          Go[i8] = ok ? g_o : 0.0f;
        }
#pragma unroll
        for (int i8 = 0; i8 < 8; i8++) {
          const int i = h * 8 + i8;
          if (kFull || i < nslots) {  // wave-uniform: stale ring entries beyond the group must not enter the recurrence
            const float2 rg = *reinterpret_cast<const float2 *>(&gb[i].z);
            const float2 bz = *reinterpret_cast<const float2 *>(&gc[i]);
            const float alpha = al[i8];
            const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
            T *= rinv;
            const float w = alpha * T;
            R = last_alpha * (s_last - R) + R;
            const float sc = kHasDA ? rg.x * dpix0 + rg.y * dpix1 + bz.x * dpix2 + bz.y * ddep + dalp
                                    : rg.x * dpix0 + rg.y * dpix1 + bz.x * dpix2;
            const float dL_dal = (sc - R) * T - tf_bg * rinv;
            s_last = sc;
            last_alpha = alpha;
            xv[i * kXStride + lane] = Go[i8] * dL_dal;
            xw[i * kXStride + lane] = w;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      f32x4 d0 = {0.0f, 0.0f, 0.0f, 0.0f}, d1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[abase + t], bv[t], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[abase + t + 1], bv[t + 1], d1, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[abase + t], bw[t], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[abase + t + 1], bw[t + 1], d1, 0, 0, 0);
      }
      const f32x4 d = d0 + d1;
      if (bn < 10) {
#pragma unroll
        for (int r = 0; r < 4; r++) dbuf[(4 * bk + r) * kAccStride + bn] = d[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < nslots) {  // block-frame moments -> the ten screen-space gradients of slot `lane`
        float *m = dbuf + lane * kAccStride;
        const float4 a = ga[lane];
        const float4 b = gb[lane];
        const float opac = gc[lane].z;
        const float cA = a.z * (-2.0f / kLog2e), cB = a.w * (-1.0f / kLog2e), cC = b.x * (-2.0f / kLog2e);
        const float xl = a.x - cx, yl = a.y - cy;
        const float m0 = m[0], mx = m[1], my = m[2], mxx = m[3], mxy = m[4], myy = m[5];
        const float svdx = xl * m0 - mx, svdy = yl * m0 - my;
        const float svdx2 = xl * xl * m0 - 2.0f * xl * mx + mxx;
        const float svdxdy = xl * yl * m0 - xl * my - yl * mx + mxy;
        const float svdy2 = yl * yl * m0 - 2.0f * yl * my + myy;
        m[0] = 0.5f * W * (-cA * svdx - cB * svdy);
        m[1] = 0.5f * H * (-cC * svdy - cB * svdx);
        m[2] = -0.5f * svdx2;
        m[3] = -svdxdy;
        m[4] = -0.5f * svdy2;
        m[5] = m0 / opac;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int f = lane + 64 * k;
        const int sl = f / 10, col = f - sl * 10;
        if (sl < nslots) {
          const float v = dbuf[sl * kAccStride + col];
          if (v != 0.0f) atomicAdd(acc + (size_t)q_id[head + sl] * kAccStride + col, v);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    head = (head + kGroup) & (kRing - 1);   // a partial group is the wave's last one
    count -= nslots;
  };

  for (uint32_t done_n = 0; done_n < todo; done_n += kChunk) {
    bool hit = false;
    float4 a, b, c;
    uint32_t id = 0, pos1 = 0;
    if (lane < kChunk && done_n + lane < todo) {
      const uint32_t pos0 = todo - 1 - done_n - lane;
      pos1 = pos0 + 1;
      id = point_list[beg + pos0];
      const float4 *src = reinterpret_cast<const float4 *>(rec + id);
      a = src[0]; b = src[1]; c = src[2];
      hit = conic_min_over_box(a.x, a.y, a.z, a.w, b.x, bx0, bx1, by0, by1) <= c.z;
    }
    const uint64_t m = __ballot(hit);
    if (hit) {
      const int qi = (head + count + __builtin_popcountll(m & ((1ull << lane) - 1ull))) & (kRing - 1);
      q_a[qi] = make_float4(a.x, a.y, -0.5f * kLog2e * a.z, -kLog2e * a.w);
      q_b[qi] = make_float4(-0.5f * kLog2e * b.x, __builtin_amdgcn_logf(b.y), b.z, b.w);
      q_c[qi] = make_float4(c.x, c.y, b.y, 0.0f);
      q_id[qi] = id;
      q_pos[qi] = pos1;
    }
    count += __builtin_popcountll(m);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool last_chunk = done_n + kChunk >= todo;
    while (count >= kGroup) process_group(std::true_type{}, kGroup);
    if (last_chunk && count > 0) process_group(std::false_type{}, count);
  }
}

}  // namespace
}  // namespace scorp

using namespace scorp;

extern "C" size_t scorp_gs3d_backward_scratch_bytes(int32_t N) {
  return align_up((size_t)(N > 0 ? N : 1) * kAccStride * sizeof(float), 256);
}

extern "C" int scorp_gs3d_backward(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                                   const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                                   const ScorpGs3dGrads *grads, void *scratch, size_t scratch_bytes,
                                   scorp_stream_t stream_) {
  if (!in || !state || !pairs || !grads || !scratch) { set_error("NULL argument to scorp_gs3d_backward"); return SCORP_ERR_INVALID; }
  if (!dL_dcolor) { set_error("dL_dcolor is NULL"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const int N = in->num_gaussians, W = in->image_width, H = in->image_height;
  if (N <= 0) return SCORP_OK;
  const StateLayout L(N, W, H);
  const PairLayout P(capacity);
  const size_t need = scorp_gs3d_backward_scratch_bytes(N);
  if (scratch_bytes < need || ((uintptr_t)scratch & 15)) {
    set_error("backward scratch too small or misaligned (%zu < %zu)", scratch_bytes, need);
    return SCORP_ERR_INVALID;
  }
  const char *base = (const char *)state, *pb = (const char *)pairs;
  float *acc = (float *)scratch;
  SCORP_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)N * kAccStride * sizeof(float), stream));
  static const bool use_shuffle = getenv("SCORP_BWD_SHUFFLE") != nullptr;  // A/B switch: the pre-MFMA reduction
  {
  ProfScope prof(kKBlendBackward, stream);
  static const bool batch256 = getenv("SCORP_BWD_BATCH256") != nullptr;
  static const int ablate = getenv("SCORP_BWD_ABLATE") ? atoi(getenv("SCORP_BWD_ABLATE")) : 0;  // timing experiments only
  static const bool per_tile = getenv("SCORP_BWD_PER_TILE") != nullptr;  // A/B switch: the workgroup-per-tile form
  if (!use_shuffle && !per_tile) {
    const int blocks = ((L.tiles + 7) / 8) * 8 * 4;
    auto wk = (dL_ddepth || dL_dalpha) ? blend_backward_wave_kernel<true> : blend_backward_wave_kernel<false>;
    wk<<<blocks, 64, 0, stream>>>(
        (const uint32_t *)(base + L.tile_start), (const uint32_t *)(pb + P.list), (const SplatRec *)(base + L.rec),
        (uint32_t)capacity, W, H, L.tiles_x, L.tiles, in->bg, (const float *)(base + L.final_T),
        (const uint32_t *)(base + L.n_contrib), dL_dcolor, dL_ddepth, dL_dalpha, acc);
  } else {
  auto kern = use_shuffle ? blend_backward_kernel_abl : (batch256 ? blend_backward_mfma_kernel<256> : blend_backward_mfma_kernel<128>);
  kern<<<L.tiles, 256, 0, stream>>>(
      (const uint32_t *)(base + L.tile_start), (const uint32_t *)(pb + P.list), (const SplatRec *)(base + L.rec),
      (uint32_t)capacity, W, H, L.tiles_x, in->bg, (const float *)(base + L.final_T),
      (const uint32_t *)(base + L.n_contrib), dL_dcolor, dL_ddepth, dL_dalpha, acc, ablate);
  }
  }
  SCORP_KERNEL_CHECK("blend_backward", in->debug, stream);
  {
    ProfScope prof(kKPreprocessBackward, stream);
    launch_preprocess_backward(in, L, (const BinRec *)(base + L.bin), acc, grads, stream);
  }
  SCORP_KERNEL_CHECK("preprocess_backward", in->debug, stream);
  return SCORP_OK;
}
