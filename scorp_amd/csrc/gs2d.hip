// gs2d.hip — 2D Gaussian splatting (surfel) path for gfx950: forward and backward.
//
// Replaces `diff_surfel_rasterization` as called at gs2dgs/gaussian_renderer/__init__.py:111-120 (outputs
// color[3,H,W], radii[N], allmap[7,H,W]; channel order :131-148).  Arithmetic follows oracle/gs2d_oracle.c: each
// surfel is a 3x3 matrix T (rows Tu,Tv,Tw) mapping its local (u,v,1) to (x*w, y*w, w) in pixels; a pixel intersects
// the surfel plane in uv space, alpha = o * exp(-0.5 * min(u^2+v^2, 2*|pixel - centre|^2)).
// Binning / per-tile depth sort are the 3DGS ones (common.hpp); tiles and 8x8 pixel blocks are culled by the exact
// footprint of {alpha >= 1/255} (an ellipse united with the low-pass disc).
#include <stdlib.h>

#include <type_traits>

#include "pergaussian.hpp"

namespace scorp {
namespace {

constexpr float kFarZ = 100.0f;
constexpr float kCutoff = 3.0f;
constexpr float kFilterSize = 0.707106f;
constexpr float kFilterInvSq = 2.0f;
constexpr float kExtentFloor = 0.0001f;
constexpr int kAcc2Stride = 20;  // d/d(pa, pb, pc)[9], d/dD, d/dTw.z, gxy[2], gnormal[3], gopacity, grgb[3]

struct alignas(16) Surfel {  // 96 bytes, gathered as six 16-byte loads
  float4 r0;  // Tu.x Tu.y Tu.z Tv.x
  float4 r1;  // Tv.y Tv.z Tw.x Tw.y
  float4 r2;  // Tw.z cx cy opacity
  float4 r3;  // n.x n.y n.z r
  float4 r4;  // g b C lp2   | footprint of {alpha >= 1/255}: the ellipse A dx^2 + 2B dx dy + C dy^2 <= 1 about (ex, ey)
  float4 r5;  // ex ey A B   | united with the disc |p - (cx,cy)|^2 <= lp2; A == 0: unknown, never cull
};
static_assert(sizeof(Surfel) == 96, "Surfel must be 96 bytes");

// Exact culling for surfels.  alpha >= 1/255 needs min(rho3d, rho2d) <= kk = 2 ln(255 o).  {rho3d <= kk} is the
// projection of the surfel's uv-disc of radius sqrt(kk): with p = (x Tw - Tu) x (y Tw - Tv) linear in the pixel,
// rho3d = (p0^2 + p1^2) / p2^2, so the region is the conic p0^2 + p1^2 - kk p2^2 <= 0 — an ellipse whenever the disc
// stays in front of the camera; preprocess2d_kernel normalises it (with 1% + 0.02 of slack on kk).  {rho2d <= kk} is
// a disc about the low-pass centre.  A pixel box that neither reaches cannot hold a contributing pixel, so dropping
// the (box, surfel) pair changes no output bit.
__device__ __forceinline__ bool surfel_reaches_box(float cx, float cy, const float4 r4, const float4 r5, float bx0, float bx1,
                                                   float by0, float by1) {
  if (r5.z == 0.0f) return true;
  const float ddx = fmaxf(fmaxf(bx0 - cx, cx - bx1), 0.0f), ddy = fmaxf(fmaxf(by0 - cy, cy - by1), 0.0f);
  if (ddx * ddx + ddy * ddy <= r4.w) return true;
  return conic_min_over_box(r5.x, r5.y, r5.z, r5.w, r4.z, bx0, bx1, by0, by1) <= 1.0f;
}

struct Pg2Args {
  int N, K, W, H, tiles_x, tiles_y, raw, count_with_atomics;
  float scale_mod;
  const float *view, *proj, *campos;
  const float *means3D, *shs, *shs_rest, *colors_precomp, *opacities, *scales, *rotations, *transmat;
};

__device__ __forceinline__ void pixel_rows(const float *pm, int W, int H, float Q[3][4]) {
#pragma clang fp contract(off)   // the splat-to-pixel transform feeds ceil(extent): same roundings as the CPU oracle
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const float p0 = pm[c * 4 + 0], p1 = pm[c * 4 + 1], p3 = pm[c * 4 + 3];
    Q[0][c] = 0.5f * W * p0 + 0.5f * (W - 1) * p3;
    Q[1][c] = 0.5f * H * p1 + 0.5f * (H - 1) * p3;
    Q[2][c] = p3;
  }
}

__device__ __forceinline__ void quat_R(float4 q, float *R) {
#pragma clang fp contract(off)
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z);     R[2] = 2 * (x * z + r * y);
  R[3] = 2 * (x * y + r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
  R[6] = 2 * (x * z - r * y);     R[7] = 2 * (y * z + r * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// ---------------------------------------------------------------------------------------------------------
// LIN: full workgroups of the training layout (dc / rest split, K = 16, scales + rotations) - parameters fetched with
// untracked loads AHEAD of the SH stream (direct global -> LDS loads) and picked up at vmcnt(12); see pergaussian.hpp.
template <int DEG, bool SPLIT, bool LIN>
__device__ __forceinline__ void preprocess2d_body(const Pg2Args &a, float *s_sh, Surfel *__restrict__ rec,
                                                  BinRec *__restrict__ bin, uint64_t *__restrict__ tile_mask,
                                                  int32_t *__restrict__ radii, uint32_t *__restrict__ tile_count) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = LIN || i < a.N;
  constexpr int NFL = 3 * (DEG + 1) * (DEG + 1);
  const size_t i0 = (size_t)blockIdx.x * 256;
  const int nrows_f = LIN ? 256 : min(256, a.N - (int)i0);
  constexpr bool lin = LIN;
  float pre[11];   // LIN: means 0-2, rotation 3-6, scale 7-8, opacity 10
  if constexpr (LIN) {
    RawParams r;
    raw_issue_params<2>(r, a.means3D + 3 * (size_t)i, a.rotations + 4 * (size_t)i, a.scales + 2 * (size_t)i, a.opacities + i);
    stage_sh_linear_async<SCORP_NT_SH ? 2 : 0>(s_sh, a.shs, a.shs_rest, i0);   // (nontemporal: see preprocess_body, gs3d_pergaussian.hip)
    raw_take_params(r, pre);
  }
  float vm[16], pm[16];
#pragma unroll
  for (int q = 0; q < 16; q++) { vm[q] = ((const CFloat *)a.view)[q]; pm[q] = ((const CFloat *)a.proj)[q]; }   // scalar cache
  BinRec br;
  br.x0 = br.y0 = br.x1 = br.y1 = 0; br.depth_bits = 0; br.radius = 0;
  bool vis = false;
  float T[9], nv[3] = {0, 0, 1}, p[3] = {0, 0, 0}, cx = 0, cy = 0, op = 0, depth = 0, mult = 1;
  float eA = 0, eB = 0, eC = 0, ex = 0, ey = 0, lp2 = 0;
  uint64_t mask = kMaskAll;
  int radius = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  if (active) {
    if constexpr (LIN) { p[0] = pre[0]; p[1] = pre[1]; p[2] = pre[2]; }
    else { p[0] = a.means3D[3 * (size_t)i]; p[1] = a.means3D[3 * (size_t)i + 1]; p[2] = a.means3D[3 * (size_t)i + 2]; }
    const float pvx = vm[0] * p[0] + vm[4] * p[1] + vm[8] * p[2] + vm[12];
    const float pvy = vm[1] * p[0] + vm[5] * p[1] + vm[9] * p[2] + vm[13];
    depth = __builtin_fmaf(vm[10], p[2], __builtin_fmaf(vm[6], p[1], __builtin_fmaf(vm[2], p[0], vm[14])));
    if (depth > kNearZ) {
      if (!LIN && a.transmat) {
#pragma unroll
        for (int q = 0; q < 9; q++) T[q] = a.transmat[9 * (size_t)i + q];
      } else {
        float Q[3][4], R[9], invn;
        pixel_rows(pm, a.W, a.H, Q);
        const float4 q_in = LIN ? make_float4(pre[3], pre[4], pre[5], pre[6]) : reinterpret_cast<const float4 *>(a.rotations)[i];
        quat_R(act_quat(q_in, a.raw, &invn), R);
        const float sx = a.scale_mod * act_scale(LIN ? pre[7] : a.scales[2 * (size_t)i], a.raw),
                    sy = a.scale_mod * act_scale(LIN ? pre[8] : a.scales[2 * (size_t)i + 1], a.raw);
#pragma unroll
        for (int r = 0; r < 3; r++) {
          T[r * 3 + 0] = sx * (Q[r][0] * R[0] + Q[r][1] * R[3] + Q[r][2] * R[6]);
          T[r * 3 + 1] = sy * (Q[r][0] * R[1] + Q[r][1] * R[4] + Q[r][2] * R[7]);
          T[r * 3 + 2] = Q[r][0] * p[0] + Q[r][1] * p[1] + Q[r][2] * p[2] + Q[r][3];
        }
#pragma unroll
        for (int r = 0; r < 3; r++) nv[r] = vm[0 * 4 + r] * R[2] + vm[1 * 4 + r] * R[5] + vm[2 * 4 + r] * R[8];
      }
      const float cosv = -(pvx * nv[0] + pvy * nv[1] + depth * nv[2]);
      const float *Tu = T, *Tv = T + 3, *Tw = T + 6;
      const float c2 = kCutoff * kCutoff;
      const float dd = c2 * Tw[0] * Tw[0] + c2 * Tw[1] * Tw[1] + -1.0f * Tw[2] * Tw[2];
      if (cosv != 0.0f && dd != 0.0f) {
        mult = cosv > 0.0f ? 1.0f : -1.0f;
        const float f0 = c2 / dd, f1 = c2 / dd, f2 = -1.0f / dd;
        cx = f0 * Tu[0] * Tw[0] + f1 * Tu[1] * Tw[1] + f2 * Tu[2] * Tw[2];
        cy = f0 * Tv[0] * Tw[0] + f1 * Tv[1] * Tw[1] + f2 * Tv[2] * Tw[2];
        const float tx = f0 * Tu[0] * Tu[0] + f1 * Tu[1] * Tu[1] + f2 * Tu[2] * Tu[2];
        const float ty = f0 * Tv[0] * Tv[0] + f1 * Tv[1] * Tv[1] + f2 * Tv[2] * Tv[2];
        const float extx = sqrtf(fmaxf(kExtentFloor, cx * cx - tx)), exty = sqrtf(fmaxf(kExtentFloor, cy * cy - ty));
        radius = (int)ceilf(fmaxf(fmaxf(extx, exty), kCutoff * kFilterSize));
        x0 = min(a.tiles_x, max(0, (int)((cx - radius) / kTile)));
        y0 = min(a.tiles_y, max(0, (int)((cy - radius) / kTile)));
        x1 = min(a.tiles_x, max(0, (int)((cx + radius + kTile - 1) / kTile)));
        y1 = min(a.tiles_y, max(0, (int)((cy + radius + kTile - 1) / kTile)));
        if ((x1 - x0) * (y1 - y0) > 0) {
          vis = true;
          op = act_opacity(LIN ? pre[10] : a.opacities[i], a.raw);
          // footprint of {alpha >= 1/255} (see surfel_reaches_box), in the frame centred on (cx, cy)
          const float kk = 1.01f * 2.0f * logf(fmaxf(255.0f * op, 1.0f)) + 0.02f;
          lp2 = 0.5f * kk;
          const float ddk = kk * Tw[0] * Tw[0] + kk * Tw[1] * Tw[1] - Tw[2] * Tw[2];
          if (ddk < 0.0f) {  // the uv-disc of radius sqrt(kk) stays in front of the camera plane: bounded projection
            const float pa[3] = {Tv[1] * Tw[2] - Tv[2] * Tw[1], Tv[2] * Tw[0] - Tv[0] * Tw[2], Tv[0] * Tw[1] - Tv[1] * Tw[0]};
            const float pb[3] = {Tw[1] * Tu[2] - Tw[2] * Tu[1], Tw[2] * Tu[0] - Tw[0] * Tu[2], Tw[0] * Tu[1] - Tw[1] * Tu[0]};
            float pc[3] = {Tu[1] * Tv[2] - Tu[2] * Tv[1], Tu[2] * Tv[0] - Tu[0] * Tv[2], Tu[0] * Tv[1] - Tu[1] * Tv[0]};
#pragma unroll
            for (int q = 0; q < 3; q++) pc[q] += pa[q] * cx + pb[q] * cy;
            const float Mxx = pa[0] * pa[0] + pa[1] * pa[1] - kk * pa[2] * pa[2];
            const float Mxy = pa[0] * pb[0] + pa[1] * pb[1] - kk * pa[2] * pb[2];
            const float Myy = pb[0] * pb[0] + pb[1] * pb[1] - kk * pb[2] * pb[2];
            const float Mx1 = pa[0] * pc[0] + pa[1] * pc[1] - kk * pa[2] * pc[2];
            const float My1 = pb[0] * pc[0] + pb[1] * pc[1] - kk * pb[2] * pc[2];
            const float M11 = pc[0] * pc[0] + pc[1] * pc[1] - kk * pc[2] * pc[2];
            const float dM = Mxx * Myy - Mxy * Mxy;
            // A surfel seen nearly edge-on projects to a sliver (a major axis of 100 px on a minor axis of 0.05 px occurs):
            // rotated on the pixel grid, its conic has Mxx Myy ~ Mxy^2 and fp32 leaves dM - and with it the centre and the
            // normalisation - no digits, which dropped blocks the sliver does cross (an isolated pixel of alpha 0.24 in
            // a fuzz case).  The footprint is trusted only while every difference below keeps at least two digits;
            // otherwise the surfel is never culled inside its rectangle (the published rasterizer's own behaviour).
#ifndef SCORP_2D_SOUND_M
#define SCORP_2D_SOUND_M 1e-3f
#endif
#ifndef SCORP_2D_SOUND_D
#define SCORP_2D_SOUND_D 1e-4f
#endif
            const bool sound = Mxx > SCORP_2D_SOUND_M * (pa[0] * pa[0] + pa[1] * pa[1] + kk * pa[2] * pa[2]) &&
                               Myy > SCORP_2D_SOUND_M * (pb[0] * pb[0] + pb[1] * pb[1] + kk * pb[2] * pb[2]) &&
                               dM > SCORP_2D_SOUND_D * (Mxx * Myy + Mxy * Mxy);
            if (sound) {
              const float ox = (My1 * Mxy - Mx1 * Myy) / dM, oy = (Mx1 * Mxy - My1 * Mxx) / dM;
              const float t1 = Mx1 * ox, t2 = My1 * oy;
              const float F = M11 + t1 + t2;
              // trust the normalisation only when F did not lose its digits to cancellation and the centre is sane
              if (F < 0.0f && -F > 1e-3f * (fabsf(M11) + fabsf(t1) + fabsf(t2)) && fabsf(ox) < 4096.0f && fabsf(oy) < 4096.0f) {
                const float nf = -1.0f / F;
                eA = Mxx * nf; eB = Mxy * nf; eC = Myy * nf; ex = cx + ox; ey = cy + oy;
                if (!(eA > 0.0f && eC > 0.0f && eA < 3.0e38f && eC < 3.0e38f)) eA = eB = eC = 0.0f;
              }
            }
          }
          if (eA > 0.0f && x1 - x0 <= 8 && y1 - y0 <= 8) {
            const float4 t4 = make_float4(0.0f, 0.0f, eC, lp2), t5 = make_float4(ex, ey, eA, eB);
            mask = 0;
            for (int ty = y0; ty < y1; ty++)
              for (int tx = x0; tx < x1; tx++)
                if (surfel_reaches_box(cx, cy, t4, t5, (float)(tx * kTile), (float)(tx * kTile + kTile - 1), (float)(ty * kTile),
                                       (float)(ty * kTile + kTile - 1)))
                  mask |= 1ull << ((ty - y0) * 8 + (tx - x0));
          }
        }
      }
    }
  }
  float rgb[3] = {0.0f, 0.0f, 0.0f};
  int clamp_bits = 0;
  if (a.shs) {
    if (lin) stage_sh_wait();
    if (__syncthreads_or(vis ? 1 : 0)) {
      if (!lin) {
        stage_sh_rows<NFL, SPLIT>(s_sh, a.shs, a.shs_rest, a.K, i0, nrows_f);
        __syncthreads();
      }
      if (vis) {
        const float dx = p[0] - ((const CFloat *)a.campos)[0], dy = p[1] - ((const CFloat *)a.campos)[1], dz = p[2] - ((const CFloat *)a.campos)[2];
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        sh_row_to_rgb<DEG>(sh_row(s_sh, threadIdx.x, lin), dx * inv, dy * inv, dz * inv, rgb);
#pragma unroll
        for (int q = 0; q < 3; q++) {
          if (rgb[q] < 0.0f) clamp_bits |= 1 << q;
          rgb[q] = fmaxf(rgb[q], 0.0f);
        }
      }
    }
  } else if (vis) {
#pragma unroll
    for (int q = 0; q < 3; q++) rgb[q] = a.colors_precomp[3 * (size_t)i + q];
  }
  if (!active) return;
  int radius_out = 0;
  if (vis) {
    float4 *dst = reinterpret_cast<float4 *>(rec + i);
    dst[0] = make_float4(T[0], T[1], T[2], T[3]);
    dst[1] = make_float4(T[4], T[5], T[6], T[7]);
    dst[2] = make_float4(T[8], cx, cy, op);
    dst[3] = make_float4(mult * nv[0], mult * nv[1], mult * nv[2], rgb[0]);
    dst[4] = make_float4(rgb[1], rgb[2], eC, lp2);
    dst[5] = make_float4(ex, ey, eA, eB);
    br.x0 = (uint16_t)x0; br.y0 = (uint16_t)y0; br.x1 = (uint16_t)x1; br.y1 = (uint16_t)y1;
    br.depth_bits = __float_as_uint(depth);
    br.radius = radius | (clamp_bits << kClampShift) | ((mult < 0.0f ? 1 : 0) << kFlipBit);
    radius_out = radius;
    if (a.count_with_atomics)
      for_each_tile(x0, y0, x1, y1, mask, a.tiles_x, [&](int t) { atomicAdd(&tile_count[t], 1u); });
  }
  reinterpret_cast<uint4 *>(bin)[i] = *reinterpret_cast<const uint4 *>(&br);
  tile_mask[i] = mask;
  radii[i] = radius_out;
}

template <int DEG, bool SPLIT>
__global__ void __launch_bounds__(256)
preprocess2d_kernel(Pg2Args a, Surfel *__restrict__ rec, BinRec *__restrict__ bin, uint64_t *__restrict__ tile_mask,
                    int32_t *__restrict__ radii, uint32_t *__restrict__ tile_count) {
  __shared__ __attribute__((aligned(16))) float s_sh[256 * kShStride];   // direct global->LDS loads land 16-byte words
  if constexpr (SPLIT && DEG == 3) {
    if (a.shs != nullptr && a.K == 16 && !a.transmat && a.N - (int)blockIdx.x * 256 >= 256) {
      preprocess2d_body<DEG, SPLIT, true>(a, s_sh, rec, bin, tile_mask, radii, tile_count);
      return;
    }
  }
  preprocess2d_body<DEG, SPLIT, false>(a, s_sh, rec, bin, tile_mask, radii, tile_count);
}

// ---------------------------------------------------------------------------------------------------------
// Blend kernels: one wave per 8x8 pixel block (64-thread workgroups, no workgroup barriers), the four blocks of a tile
// numbered onto the same XCD — the 2DGS twins of blend_forward_wave_kernel / blend_backward_wave_kernel.  Each lane
// of a chunk gathers one surfel of the tile's list and tests its exact footprint against the block
// (surfel_reaches_box); survivors are compacted into a per-wave LDS ring.
// ---------------------------------------------------------------------------------------------------------
// The ray-surfel intersection in linear form.  With k = x Tw - Tu, l = y Tw - Tv the reference intersects with
// p = k x l and s = (p0, p1) / p2.  p is LINEAR in the pixel, and p . Tw = det[Tu; Tv; Tw] =: D for every pixel, so
// the hit depth s0 Tw0 + s1 Tw1 + Tw2 = D / p2.  A wave expands the form about the CENTRE (bxc, byc) OF ITS 8x8 BLOCK:
// with kb = bxc Tw - Tu, lb = byc Tw - Tv (each component ONE fma, so the near-cancellation of bxc Tw2 against Tu2
// costs a single rounding of the small result),
//     p(x, y) = (x - bxc) pa + (y - byc) pb + pc,    pa = Tv x Tw,  pb = Tw x Tu,  pc = kb x lb,
// and (x - bxc, y - byc) is a per-lane constant in {-3.5 .. 3.5}: six FMAs per pixel, no reciprocal for the depth (nor
// for 1 / depth: p2 / D).  Expanded about the image origin instead (pc = Tu x Tv, round 1) p0 and p1 were differences
// of terms ~10^2 times their size, and 2 % of random scenes held a surfel whose gradient missed the oracle's by more
// than its tolerance.  The lane that inserts a surfel into a wave's ring computes pa, pb, pc, D once.
// The backward accumulates the gradients of (pa, pb, pc, D) about ONE point per surfel whatever the block - its centre
// (cx, cy) clamped into the image, (ex, ey) - as products of dp with (x - ex, y - ey, 1); preprocess2d_backward_kernel
// chains them back to T with pc = (ex Tw - Tu) x (ey Tw - Tv), (ex, ey) held fixed (p as a function of T does not
// depend on where it is expanded).  A splat-centred frame keeps those sums and their cross products with T free of
// the (x, y)-weighted against (cx, cy)-weighted cancellation of the origin form; the clamp keeps a centre far outside
// the image (a large surfel seen from close by) from re-creating it.
struct SurfelLin { float4 e0, e1, e2, e3; };   // (pa, pb0) (pb1, pb2, pc0, pc1) (pc2, D, cx, cy) (log2 o, Tw2, 1/D, 1/Tw2)
__device__ __forceinline__ SurfelLin surfel_lin(const float4 r0, const float4 r1, const float4 r2, float bxc, float byc) {
#pragma clang fp contract(off)
  const float Tu[3] = {r0.x, r0.y, r0.z}, Tv[3] = {r0.w, r1.x, r1.y}, Tw[3] = {r1.z, r1.w, r2.x};
  float pa[3], pb[3], pc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    pa[i] = __builtin_fmaf(Tv[j], Tw[k], -(Tv[k] * Tw[j]));
    pb[i] = __builtin_fmaf(Tw[j], Tu[k], -(Tw[k] * Tu[j]));
  }
  float kc[3], lc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { kc[i] = __builtin_fmaf(bxc, Tw[i], -Tu[i]); lc[i] = __builtin_fmaf(byc, Tw[i], -Tv[i]); }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    pc[i] = __builtin_fmaf(kc[j], lc[k], -(kc[k] * lc[j]));
  }
  const float D = __builtin_fmaf(Tu[0], pa[0], __builtin_fmaf(Tu[1], pa[1], Tu[2] * pa[2]));
  SurfelLin L;
  L.e0 = make_float4(pa[0], pa[1], pa[2], pb[0]);
  L.e1 = make_float4(pb[1], pb[2], pc[0], pc[1]);
  L.e2 = make_float4(pc[2], D, r2.y, r2.z);
  L.e3 = make_float4(__builtin_amdgcn_logf(r2.w), Tw[2], __builtin_amdgcn_rcpf(D), __builtin_amdgcn_rcpf(Tw[2]));
  return L;
}

struct Eval2 { float s0, s1, pz, rz, dx, dy, depth, rdepth, Go, alpha; bool use3d; };
// Same decisions in forward and backward: every product-sum is written as an explicit fma and contraction is off, so
// the two kernels cannot round the intersection differently.  Go = opacity * G (alpha before the 0.99 clamp).
__device__ __forceinline__ bool eval_surfel(const float4 e0, const float4 e1, const float4 e2, const float4 e3, float qx,
                                            float qy, float pxf, float pyf, Eval2 &h) {   // (qx, qy) = pixel - block centre
#pragma clang fp contract(off)
  const float p0 = __builtin_fmaf(e0.x, qx, __builtin_fmaf(e0.w, qy, e1.z));
  const float p1 = __builtin_fmaf(e0.y, qx, __builtin_fmaf(e1.x, qy, e1.w));
  h.pz = __builtin_fmaf(e0.z, qx, __builtin_fmaf(e1.y, qy, e2.x));
#ifdef SCORP_2D_PROBE_P
  {   // measurement build only: the six multiply-adds of p a second time (what taking them off the vector pipe could save)
    const float x0 = __builtin_fmaf(e0.y, qx, __builtin_fmaf(e0.z, qy, e1.w));
    const float x1 = __builtin_fmaf(e0.w, qx, __builtin_fmaf(e1.y, qy, e2.x));
    const float x2 = __builtin_fmaf(e0.x, qx, __builtin_fmaf(e1.x, qy, e1.z));
    asm volatile("" ::"v"(x0), "v"(x1), "v"(x2));
  }
#endif
  h.rz = __builtin_amdgcn_rcpf(h.pz);
  h.s0 = p0 * h.rz; h.s1 = p1 * h.rz;
  const float rho3d = __builtin_fmaf(h.s0, h.s0, h.s1 * h.s1);
  h.dx = e2.z - pxf; h.dy = e2.w - pyf;
  const float rho2d = kFilterInvSq * __builtin_fmaf(h.dx, h.dx, h.dy * h.dy);
  h.use3d = rho3d <= rho2d;
  const float rho = fminf(rho3d, rho2d);            // rho2d is finite, so a NaN / inf rho3d (pz == 0) falls back to it
  h.depth = h.use3d ? e2.y * h.rz : e3.y;
  h.rdepth = h.use3d ? h.pz * e3.z : e3.w;          // 1 / depth without a reciprocal
  h.Go = __builtin_amdgcn_exp2f(__builtin_fmaf(-0.5f * 1.4426950408889634f, rho, e3.x));
  h.alpha = fminf(kAlphaMax, h.Go);
  return (h.pz != 0.0f) & (h.depth >= kNearZ) & (h.alpha >= kAlphaMin);
}

// Hits per straight-line group and the waves per SIMD the kernel is held to.  Groups of 8 behind a 128-slot ring (rounds
// 2 - 3) cost 164 registers and 11.8 KB of LDS per wave: three waves per SIMD, 13 per CU by LDS.  Groups of 4 behind a
// 68-slot ring: 96 registers, 6.3 KB - five waves per SIMD.  Same box, rocprof, S6: 298 -> 272 us (render-only form
// 293 -> 262); groups of 8 squeezed into 128 registers (spills): 334; groups of 2 need more registers than groups of 4.
// The blend order does not depend on the group size (bit-identical images and hit lists).
#ifndef SCORP_2D_FGROUP
#define SCORP_2D_FGROUP 4
#endif
#ifndef SCORP_2D_FWAVES
#define SCORP_2D_FWAVES 5
#endif
constexpr int k2FChunk = 64, k2FGroup = SCORP_2D_FGROUP;
// ring: at most k2FGroup - 1 left-over hits + 64 new ones, a multiple of the group (a group's slots never wrap)
constexpr int k2FRing = (k2FChunk + k2FGroup - 1 + k2FGroup - 1) / k2FGroup * k2FGroup;

// kForBackward = false (scorp_gs2d_render_image): no cull verdicts, per-pixel state or contributor bookkeeping is left
// behind for a backward pass.
template <bool kForBackward>
__global__ void __launch_bounds__(64, SCORP_2D_FWAVES)
blend2d_forward_wave_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                            const Surfel *__restrict__ rec, uint32_t capacity, int W, int H, int tiles_x, int tiles,
                            const float *__restrict__ bg, float *__restrict__ out_color, float *__restrict__ allmap,
                            float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, uint32_t *__restrict__ hits) {
  __shared__ float4 q0[k2FRing], q1[k2FRing], q2[k2FRing], q3[k2FRing], q4[k2FRing];   // SurfelLin + (normal, r)
  __shared__ float2 q5[k2FRing];                                                          // (g, b)
  __shared__ __attribute__((aligned(16))) uint32_t q_pos[k2FRing];
  const int lane = threadIdx.x;
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int tile = (kk >> 2) * 8 + xcd, quad = kk & 3;
  if (tile >= tiles) return;
  const int bx = (tile % tiles_x) * kTile + (quad & 1) * 8, by = (tile / tiles_x) * kTile + (quad >> 1) * 8;
  const int px = bx + (lane & 7), py = by + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const float bx0 = (float)bx, bx1 = (float)(bx + 7), by0 = (float)by, by1 = (float)(by + 7);
  const float bxc = (float)bx + 3.5f, byc = (float)by + 3.5f;                    // the linear form's expansion point (surfel_lin)
  const float qxb = (float)(lane & 7) - 3.5f, qyb = (float)(lane >> 3) - 3.5f;   // this pixel about it
  const uint32_t beg = min(tile_start[2 * tile], capacity), end = min(tile_start[2 * tile + 1], capacity);   // (start, end) per tile
  const uint32_t n = end - beg;
  const float fn = kFarZ / (kFarZ - kNearZ);
  float T = inside ? 1.0f : -1.0f, C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0, Dp = 0, M1 = 0, M2 = 0, dist = 0, med = 0;
  uint32_t last = 0, med_c = 0;
  int head = 0, count = 0;
  uint32_t nh = 0;   // hits found so far (wave-uniform): a hit's 1-based position in the block's hit list is its `pos`
  uint32_t *my_hits = hits + (size_t)quad * capacity + beg;
  // The chunk's gathers (list entry -> 96-byte record) are dependent loads; software-pipelined: while chunk c is blended
  // the records of chunk c+1 and the list entries of chunk c+2 are in flight.  The entries that pass the footprint test
  // are left for the backward as the block's HIT LIST (ids, compacted, in blend order) in the pair buffer's key region
  // (dead after the sort), as in the 3-D kernels: the backward replays it and never looks at the tile's list again.
  auto fetch_id = [&](uint32_t bs) { return (bs + lane < n) ? point_list[beg + bs + lane] : 0xFFFFFFFFu; };
  auto fetch_rec = [&](uint32_t id_, float4 &a0_, float4 &a1_, float4 &a2_, float4 &a3_, float4 &a4_, float4 &a5_) {
    if (id_ != 0xFFFFFFFFu) {
      const float4 *src = reinterpret_cast<const float4 *>(rec + id_);
      a0_ = src[0]; a1_ = src[1]; a2_ = src[2]; a3_ = src[3]; a4_ = src[4]; a5_ = src[5];
    }
  };
  float4 r0, r1, r2, r3, r4, r5;
  uint32_t id0 = fetch_id(0);
  fetch_rec(id0, r0, r1, r2, r3, r4, r5);
  uint32_t id1 = fetch_id(k2FChunk);
  for (uint32_t base = 0; base < n; base += k2FChunk) {
    if (__ballot(T > 0.0f) == 0) break;
    float4 nx0, nx1, nx2, nx3, nx4, nx5;   // next chunk's records
    fetch_rec(id1, nx0, nx1, nx2, nx3, nx4, nx5);
    const uint32_t id2 = fetch_id(base + 2 * k2FChunk);
    bool hit = false;
    if (id0 != 0xFFFFFFFFu) hit = surfel_reaches_box(r2.y, r2.z, r4, r5, bx0, bx1, by0, by1);
    const uint64_t m = __ballot(hit);
    if (hit) {
      const uint32_t rank = nh + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      int qi = head + count + (int)(rank - nh);
      qi = qi >= k2FRing ? qi - k2FRing : qi;
      const SurfelLin L = surfel_lin(r0, r1, r2, bxc, byc);
      q0[qi] = L.e0; q1[qi] = L.e1; q2[qi] = L.e2; q3[qi] = L.e3; q4[qi] = r3; q5[qi] = make_float2(r4.x, r4.y);
      q_pos[qi] = rank + 1u;
      if constexpr (kForBackward) my_hits[rank] = id0;   // (rank < n: inside this tile's slice of the region)
    }
    count += __builtin_popcountll(m);
    nh += (uint32_t)__builtin_popcountll(m);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool last_chunk = base + k2FChunk >= n;
    // full groups run straight-line; only a wave's final group is partial, so head stays a multiple of k2FGroup and
    // the slots of a group are head + i without wrap-around (one LDS base per array, immediate offsets)
    auto blend_group = [&](auto full, int nslots) {
      constexpr bool kFull = decltype(full)::value;
      int hv = head;
      asm volatile("" : "+v"(hv));   // keep the group's LDS bases in VGPRs
      const float4 *g0 = q0 + hv, *g1 = q1 + hv, *g2 = q2 + hv, *g3 = q3 + hv, *g4 = q4 + hv;
      const float2 *g5 = q5 + hv;
      const uint32_t *gp = q_pos + hv;
      static_assert(k2FGroup == 8 || k2FGroup == 4 || k2FGroup == 2, "positions are fetched as 16- or 8-byte LDS reads");
      uint32_t pos[k2FGroup];
      if constexpr (k2FGroup == 2) {
        const uint2 pl = *reinterpret_cast<const uint2 *>(gp);
        pos[0] = pl.x; pos[1] = pl.y;
      } else {
        const uint4 pl = *reinterpret_cast<const uint4 *>(gp);
        pos[0] = pl.x; pos[1] = pl.y; pos[2] = pl.z; pos[3] = pl.w;
        if constexpr (k2FGroup == 8) {
          const uint4 ph = *reinterpret_cast<const uint4 *>(gp + 4);
          pos[4] = ph.x; pos[5] = ph.y; pos[6] = ph.z; pos[7] = ph.w;
        }
      }
      float al[k2FGroup], dz[k2FGroup], mm[k2FGroup];
      bool okv[k2FGroup];   // (wave masks in scalar registers: the straight-line group keeps them there)
#pragma unroll
      for (int i = 0; i < k2FGroup; i++) {
        Eval2 h;
        const bool ok = eval_surfel(g0[i], g1[i], g2[i], g3[i], qxb, qyb, pxf, pyf, h) & (kFull || i < nslots);
        okv[i] = ok;
        al[i] = ok ? h.alpha : 0.0f;
        dz[i] = ok ? h.depth : 1.0f;
        mm[i] = __builtin_fmaf(-fn * kNearZ, ok ? h.rdepth : 1.0f, fn);   // the backward's m_d, to the bit
      }
#pragma unroll
      for (int i = 0; i < k2FGroup; i++) {
        if (kFull || i < nslots) {  // wave-uniform
          const float4 nr = g4[i];
          const float2 gb = g5[i];
          // saturation is latched by the sign of T (see blend_forward_wave_kernel)
          const float alpha = al[i];
          const float test_T = T * (1.0f - alpha);
          const bool ok = test_T >= kTMin;
          const float ae = ok ? alpha : 0.0f;
          const float w = ae * T;
          const float A = 1.0f - T, mz = mm[i];
          dist += (mz * mz * A + M2 - 2.0f * mz * M1) * w;
          Dp += dz[i] * w; M1 += mz * w; M2 += mz * mz * w;
          const bool contributes = okv[i] & ok;   // = (ae > 0): a hit that passed eval_surfel has alpha >= 1 / 255 (two scalar masks)
          const bool is_med = contributes & (T > 0.5f);
          med = is_med ? dz[i] : med;
          if constexpr (kForBackward) med_c = is_med ? pos[i] : med_c;
          N0 += nr.x * w; N1 += nr.y * w; N2 += nr.z * w;
          C0 += nr.w * w; C1 += gb.x * w; C2 += gb.y * w;
          T = ok ? test_T : -fabsf(T);
          if constexpr (kForBackward) last = contributes ? pos[i] : last;
        }
      }
      head = head + k2FGroup == k2FRing ? 0 : head + k2FGroup;
      count -= nslots;
    };
    while (count >= k2FGroup) blend_group(std::true_type{}, k2FGroup);
    if (last_chunk && count > 0) blend_group(std::false_type{}, count);
    id0 = id1; r0 = nx0; r1 = nx1; r2 = nx2; r3 = nx3; r4 = nx4; r5 = nx5; id1 = id2;
  }
  if (inside) {
    const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
    T = fabsf(T);
    if constexpr (kForBackward) {
      final_T[pix] = T; final_T[HW + pix] = M1; final_T[2 * HW + pix] = M2;
      n_contrib[pix] = last; n_contrib[HW + pix] = med_c;
    }
    out_color[pix] = C0 + T * bg[0]; out_color[HW + pix] = C1 + T * bg[1]; out_color[2 * HW + pix] = C2 + T * bg[2];
    allmap[pix] = Dp; allmap[HW + pix] = 1.0f - T;
    allmap[2 * HW + pix] = N0; allmap[3 * HW + pix] = N1; allmap[4 * HW + pix] = N2;
    allmap[5 * HW + pix] = med; allmap[6 * HW + pix] = dist;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The pixel -> surfel reduction of the 2-D backward on the matrix cores (round 4; it was a twenty-value butterfly of
// v_permlane32 / 16_swap + DPP adds: 15 swaps at 8.4 issue cycles + 27 adds per hit, 44 % of the kernel).
//
// The twenty sums of a hit factor, like the ten of the 3-D kernel, into a handful of per-pixel VALUES times per-pixel
// BASES that do not depend on the hit:
//     dp0, dp1, dp2  x {x, y, 1}                (nine sums: d/d pa, pb, pc; x, y = the pixel about the block centre - the offset to
//                                                the surfel's accumulation point is a per-hit constant applied afterwards)
//     zr, z2, t x {1};   t2 x {x, y, 1}         (depth, low-pass depth, opacity; the low-pass centre gradient)
//     w x {dL/dn (3), dL/drgb (3)}              (normal and colour)
// Eight values per (hit, pixel).  Each leaves the lane as TWO fp16 terms (x = h1 + h2, h1 = rtz16(x), h2 = rtz16(x - h1):
// 22 bits, exact products, fp32 accumulation), both in ONE dword, so a hit costs eight ds_write_b32; rows of the matrix
// are (hit of the pair, kind), its K dimension (pixel, term); B holds each basis value twice (once per term: the MFMA
// adds the two terms by itself) - columns 0 = 1, 1 = x, 2 = y, 3..8 = the first fp16 term of the six upstream gradients,
// 9..14 = their remainder.  Two hits per pass, four v_mfma_f32_16x16x32_f16 per pass.
// Range: block floating point, three exact powers of two.  The upstream gradients are pre-scaled per wave from the
// block's largest |dL/dpixel| (everything the recurrence produces is linear in them: the B operand's gradient columns
// sit high in the fp16 range); the four values that contain 1 / p.z (p.z ~ surfel extent^2 in pixels: anything from 1e-3
// to 1e3) are scaled per hit by the power of two below |p.z| at the block centre; and the three per-pixel roots of a
// hit - all eight values are linear in them - by the power of two that brings the largest of them over the hit's pixels
// into [2^4, 2^5) (one wave maximum per hit), so that the pair's 22 bits hold down to 2^-17 of the hit's largest value
// whatever T, alpha and the depth scale are.  The blend weight is carried as w * 2^10.  The conversion saturates
// (round toward zero) instead of overflowing.  The sums are unscaled by exact powers of two when they are read out.
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4_2d __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_2d __attribute__((ext_vector_type(8)));
union Frag2 { f16x8_2d v; uint4 q; uint32_t d[4]; };
// Row strides chosen by the LDS banking rules (MI355X_MICROARCH.md, section LDS; scripts/dev/lds_conflicts.py does the arithmetic):
// a ds_read_b128 is served in four 16-lane groups that each hold all sixteen matrix rows, eight from one K group of the MFMA
// operand and eight from the next (+ 4 dwords).  With 68 dwords per row the sixteen 16-byte slots of a group collided pairwise
// (8 LDS cycles per read instead of 4); with 72 a row starts two slots after the previous one, the rows of one K group take the
// even slots and those of the other the odd ones.  The result tile's rows are 20 floats apart so that the four rows a
// ds_write_b32 touches per 32-lane group start 16 banks apart (17: 4 cycles per write instead of 2).  Round 5's counters had
// 43.8 M conflict cycles per launch on S6 (44 % of the LDS-active cycles): 16 per pair from the reads, 8 from the writes.
#ifndef SCORP_2D_XSTRIDE
#define SCORP_2D_XSTRIDE 72
#endif
#ifndef SCORP_2D_DSTRIDE
#define SCORP_2D_DSTRIDE 20
#endif
constexpr int k2XStride = SCORP_2D_XSTRIDE;  // dwords per row of the [16][64] value matrix (16-byte aligned rows, conflict-free row writes)
constexpr int k2DStride = SCORP_2D_DSTRIDE;  // floats per row of the 16 x 16 result tile
#ifndef SCORP_2D_TARGET_EXP
#define SCORP_2D_TARGET_EXP 12
#endif
constexpr int k2TargetExp = SCORP_2D_TARGET_EXP;   // the block's largest |upstream gradient| is scaled into [2^12, 2^13): the B operand's
                                             // gradient columns (two fp16 terms) keep 22 bits for a pixel whose gradient is down to 2^-14 of the
                                             // block's largest.  The A side does not depend on it any more: every hit's values are brought to a
                                             // fixed range by the hit's own power of two (`sg` in the kernel; until round 5 they rode on this
                                             // scale alone, [1, 2) with a factor of nine of head room for a depth-loss-dominated block at the
                                             // far plane - tests/test_gs2d_gpu.py "far_depth" - and an absolute floor of 2^-24 under them)
constexpr float k2WScale = 1024.0f;
__device__ __forceinline__ uint32_t pack_rtz16_2d(float lo, float hi) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
__device__ __forceinline__ float half_lo_2d(uint32_t p) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(p & 0xFFFFu)); }
// x -> (h1(x) | h2(x) << 16), h1 = rtz16(x), h2 = rtz16(x - h1).  h1 as an fp32 value is x with the low thirteen mantissa
// bits cleared (fp16 carries ten), so the remainder needs no conversion back and ONE v_cvt_pkrtz makes the whole dword:
// three instructions per value (it was nine per two: two packs, two conversions back, two fma, two v_perm).  Below the
// fp16 normal range (2^-14; the hit's largest value sits at 2^4 .. 2^14, see `sg`) the pair ends at the same 2^-24.
// Round 6: TWO instructions - v_cvt_pkrtz writes h1 into the low half, v_fma_mixhi_f16 forms x - h1 in fp32 (exact: the
// difference is representable) straight from that half and rounds it to fp16 into the high half of the same register.  The
// remainder is rounded to nearest where it was truncated (either way x = h1 + h2 to 2^-22 relative); a value beyond the fp16
// range (not reachable: the hit's largest is held below 1.4e4, see `sg`) now ends as inf instead of a silently wrong pair.
__device__ __forceinline__ uint32_t split_one(float x) {
  uint32_t d = pack_rtz16_2d(x, 0.0f);
  asm("v_fma_mixhi_f16 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(x));
  return d;
}
// What lane `o` (0..19 of either half of the wave) reads out of the result tile: out = alpha * D[row][c1] + beta * D[row][c2],
// beta one of {0, 1, ox, oy, kF * offx, kF * offy} by `bsel`; `cls`: which unscaling applies (0: 1 / (sv Sh), 1: 1 / sv, 2: 1 / (sv 2^10))
struct ReadOut { int row, c1, c2, bsel, cls; float alpha; };
__device__ __forceinline__ ReadOut read_out_of(int o) {
  ReadOut r;
  r.row = 0; r.c1 = 0; r.c2 = 0; r.bsel = 0; r.cls = 1; r.alpha = 1.0f;
  if (o < 3) { r.row = o; r.c1 = 1; r.bsel = 2; r.cls = 0; }                       // d/d pa = sum dp (x + ox)
  else if (o < 6) { r.row = o - 3; r.c1 = 2; r.bsel = 3; r.cls = 0; }              // d/d pb = sum dp (y + oy)
  else if (o < 9) { r.row = o - 6; r.cls = 0; }                                    // d/d pc
  else if (o == 9) { r.row = 3; r.cls = 0; }                                       // d/d D (zr)
  else if (o == 10) { r.row = 4; }                                                 // low-pass depth (z2)
  else if (o == 11) { r.row = 5; r.c1 = 1; r.bsel = 4; r.alpha = -kFilterInvSq; }  // kF t2 (offx - x)
  else if (o == 12) { r.row = 5; r.c1 = 2; r.bsel = 5; r.alpha = -kFilterInvSq; }
  else if (o < 16) { r.row = 7; r.c1 = 3 + (o - 13); r.c2 = 9 + (o - 13); r.bsel = 1; r.cls = 2; }    // w x dL/dn (first term + remainder)
  else if (o == 16) { r.row = 6; r.alpha = -1.0f; }                                // - sum t
  else if (o < 20) { r.row = 7; r.c1 = 6 + (o - 17); r.c2 = 12 + (o - 17); r.bsel = 1; r.cls = 2; }   // w x dL/drgb
  return r;
}

#ifndef SCORP_2D_BCHUNK
#define SCORP_2D_BCHUNK 32
#endif
#ifndef SCORP_2D_MFMA_CHAINS
#define SCORP_2D_MFMA_CHAINS 1
#endif
constexpr int k2BChunk = SCORP_2D_BCHUNK;   // hits staged per chunk: 32 (the staging arrays are 112 bytes per hit; with 64 the wave's 11.5 KB of
                                            // LDS held the kernel at 13 waves per CU where its 124 registers allow 16: 706 -> 665 us)

// Four waves per SIMD (128 registers; every instantiation fits without scratch when asked to - left at three by the launch
// bounds the split form allocated 130).  Held to five (96 registers) it spills 26 dwords: 673 -> 740 us (same box, round 4).
#ifndef SCORP_2D_BWAVES
#define SCORP_2D_BWAVES 4
#endif
// Three forms, as for the 3-D kernel (scorp_gs2d_backward_ex, include/scorp_gs.h):
//   * SPLIT (default): the description above.
//   * kExact (SCORP_BACKWARD_EXACT_FP32): the eight values of a hit stay fp32 in the matrix, the B operand holds the basis and
//     the six upstream gradients in fp32 (nine columns), the reduction is 16 v_mfma_f32_16x16x4_f32 per pair of hits.  No
//     fp16 terms, so none of the three power-of-two scales either (wave, hit, 1 / p.z): every operand is the fp32 number the
//     reference's arithmetic would carry.  The fp32 MFMA runs at the vector rate (34 cycles each): this
//     form pays 8 of them per hit where the split form pays 2 fp16 ones, and saves the split's ~40 instructions.
//   * kDet (SCORP_BACKWARD_DETERMINISTIC): the twenty sums of a (block, hit) leave as one plain row
//     partial[4 * pair + block] (pair = the (surfel, tile) pair's ordinal in surfel-major order, gs3d_backward.hip) with a
//     flag byte; reduce_pair_rows2d_kernel adds a surfel's rows in a fixed order.  No float atomics.
template <bool kHasMap, bool kExact = false, bool kDet = false>
__global__ void __launch_bounds__(64, SCORP_2D_BWAVES)
blend2d_backward_wave_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                             const Surfel *__restrict__ rec, uint32_t capacity, int W, int H, int tiles_x, int tiles,
                             const float *__restrict__ bg, const float *__restrict__ final_T,
                             const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dcolor,
                             const float *__restrict__ dL_dallmap, float *__restrict__ acc,
                             const uint32_t *__restrict__ hits, float *__restrict__ partial, uint8_t *__restrict__ row_flags,
                             const uint32_t *__restrict__ pair_base, const BinRec *__restrict__ bin,
                             const uint64_t *__restrict__ tile_mask) {
  __shared__ float4 q0[k2BChunk], q1[k2BChunk], q2[k2BChunk], q3[k2BChunk], q4[k2BChunk];
  __shared__ float2 q5[k2BChunk];                        // q0..q3: SurfelLin, q4: (normal, r), q5: (g, b)
  __shared__ uint32_t q_id[k2BChunk], q_pos[k2BChunk];
  __shared__ float q6[k2BChunk];                         // Sh: the hit's power-of-two scale for the 1 / p.z values (0: taken at replay)
  // what the read-out lanes multiply with, one row of eight floats per hit, formed once by the staging lane:
  // [0, 1, ox, oy, kF (cx - bxc), kF (cy - byc), 1 / Sh, 1] - a lane picks its two entries by index (it was five selects)
  __shared__ __attribute__((aligned(16))) float q7[k2BChunk * 8];
  // [row = hit of the pair x kind][pixel] matrix of (h1 | h2 << 16) dwords; the 16 x 16 result tile reuses its first rows
  __shared__ __attribute__((aligned(16))) uint32_t xm2[16 * k2XStride];
  float *dbuf = reinterpret_cast<float *>(xm2);
  static_assert(16 * k2DStride <= 16 * k2XStride && 64 * 6 <= 16 * k2XStride, "the result tile and the prologue scratch fit the matrix");
  const int lane = threadIdx.x;
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int tile = (kk >> 2) * 8 + xcd, quad = kk & 3;
  if (tile >= tiles) return;
  const int bx = (tile % tiles_x) * kTile + (quad & 1) * 8, by = (tile / tiles_x) * kTile + (quad >> 1) * 8;
  const int px = bx + (lane & 7), py = by + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const float bxc = (float)bx + 3.5f, byc = (float)by + 3.5f;                    // the linear form's expansion point (surfel_lin)
  const float qxb = (float)(lane & 7) - 3.5f, qyb = (float)(lane >> 3) - 3.5f;   // this pixel about it
  const uint32_t beg = min(tile_start[2 * tile], capacity), end = min(tile_start[2 * tile + 1], capacity);   // (start, end) per tile
  if (end == beg) return;
  const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
  // all of the pixel's loads are issued together; pixels nothing was blended into drop their upstream gradient
  // afterwards by a select (it may be NaN)
  float T_final = 0, final_D = 0, final_D2 = 0, dpix0 = 0, dpix1 = 0, dpix2 = 0, ddep = 0, dacc = 0, dn0 = 0, dn1 = 0, dn2 = 0,
        dmed = 0, dreg = 0;
  uint32_t last = 0, med_c = 0;
  if (inside) {
    T_final = final_T[pix];
    last = n_contrib[pix];
    dpix0 = dL_dcolor[pix]; dpix1 = dL_dcolor[HW + pix]; dpix2 = dL_dcolor[2 * HW + pix];
    if (kHasMap) {
      final_D = final_T[HW + pix]; final_D2 = final_T[2 * HW + pix];
      med_c = n_contrib[HW + pix];
      ddep = dL_dallmap[pix]; dacc = dL_dallmap[HW + pix];
      dn0 = dL_dallmap[2 * HW + pix]; dn1 = dL_dallmap[3 * HW + pix]; dn2 = dL_dallmap[4 * HW + pix];
      dmed = dL_dallmap[5 * HW + pix]; dreg = dL_dallmap[6 * HW + pix];
    }
  }
  if (last == 0) { final_D = final_D2 = dpix0 = dpix1 = dpix2 = ddep = dacc = dn0 = dn1 = dn2 = dmed = dreg = 0.0f; med_c = 0; }
  const float final_A = 1.0f - T_final;
  const float fn = kFarZ / (kFarZ - kNearZ);
  uint32_t todo = last;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) todo = max(todo, (uint32_t)__shfl_xor((int)todo, off, 64));
  todo = (uint32_t)__builtin_amdgcn_readfirstlane((int)todo);   // wave-uniform: keeps the chunk loop's counters in SGPRs
  // one power-of-two scale per wave from the block's largest upstream gradient (exact; everything below is linear in them)
  float sv = 1.0f, inv_sv = 1.0f;
  if constexpr (!kExact) {
    float amax = fmaxf(fmaxf(fabsf(dpix0), fabsf(dpix1)), fabsf(dpix2));
    if (kHasMap) amax = fmaxf(fmaxf(fmaxf(amax, fabsf(ddep)), fmaxf(fabsf(dacc), fabsf(dmed))),
                              fmaxf(fmaxf(fabsf(dn0), fabsf(dn1)), fmaxf(fabsf(dn2), fabsf(dreg))));
    const int eb = (int)((wave_max_u32(__float_as_uint(amax)) >> 23) & 0xFFu);   // biased exponent; 0: zero / denormal
    const int sb = eb == 0 ? 127 : min(max(254 + k2TargetExp - eb, 1), 253);
    sv = __uint_as_float((uint32_t)sb << 23);
    inv_sv = __uint_as_float((uint32_t)(254 - sb) << 23);
  }
  dpix0 *= sv; dpix1 *= sv; dpix2 *= sv; ddep *= sv; dacc *= sv; dn0 *= sv; dn1 *= sv; dn2 *= sv; dmed *= sv; dreg *= sv;
  // (of the SCALED gradients, like everything below); per-pixel constants of the recurrence formed once: T_final bg . dL/dc,
  // and the distortion term's final_A, final_D, final_D2 with the map's upstream gradient folded in
  const float tf_bg = T_final * (bg[0] * dpix0 + bg[1] * dpix1 + bg[2] * dpix2);
  const float far_ = final_A * dreg, fdr_ = final_D * dreg, fd2r_ = final_D2 * dreg;
  // B operand.  A lane supplies ONE column bn of the basis for the pixels 16 m + 4 bk + j (MFMA m, j = 0..3), each value twice
  // (once per fp16 term of the A side).  The six gradient columns come from the other lanes through LDS (the matrix is idle).
  Frag2 bh[kExact ? 1 : 4];
  float bb[kExact ? 16 : 1];   // exact form: fp32 MFMA 4 m + j covers the pixels 16 m + 4 bk + j (one per K slot bk)
  {
    float *xs = reinterpret_cast<float *>(xm2);
    xs[lane * 6 + 0] = dn0; xs[lane * 6 + 1] = dn1; xs[lane * 6 + 2] = dn2;
    xs[lane * 6 + 3] = dpix0; xs[lane * 6 + 4] = dpix1; xs[lane * 6 + 5] = dpix2;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int bn = lane & 15, bk = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 16 * m + 4 * bk + j;
        float b = 0.0f;
        b = bn == 0 ? 1.0f : b;
        b = bn == 1 ? (float)(q & 7) - 3.5f : b;
        b = bn == 2 ? (float)(q >> 3) - 3.5f : b;
        if constexpr (kExact) {
          if (bn >= 3 && bn <= 8) b = xs[q * 6 + (bn - 3)];   // columns 3..8: the six upstream gradients, whole; 9..15: zero
          bb[4 * m + j] = b;
        } else {
          if (bn >= 3 && bn <= 14) {
            const float g = xs[q * 6 + (bn - 3) % 6];
            const float g1 = half_lo_2d(pack_rtz16_2d(g, 0.0f));
            b = bn <= 8 ? g1 : g - g1;                       // columns 3..8: first fp16 term, 9..14: the remainder
          }
          const uint32_t hb = pack_rtz16_2d(b, 0.0f) & 0xFFFFu;
          bh[m].d[j] = hb | (hb << 16);
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const ReadOut ro = read_out_of(lane & 31);
  const int ro_beta = ro.bsel, ro_us = kExact ? 7 : (ro.cls == 0 ? 6 : 7);   // this lane's entries of a hit's q7 row
  constexpr float kWCarry = kExact ? 1.0f : k2WScale;   // the blend weight travels as w * 2^10 in the split form only
  const float ro_unscale = ro.cls == 2 ? inv_sv * (1.0f / kWCarry) : inv_sv;
  const int abase = (lane & 15) * k2XStride + 4 * (lane >> 4);
  // Two hits per pass over the matrix pipe: `pend` halves of the matrix are filled (slots pend_s[0], pend_s[1] of the chunk)
  int pend = 0, pend_s0 = 0, pend_s1 = 0;
  float pend_iw0 = 1.0f, pend_iw1 = 1.0f;   // 1 / (the hit's block-floating-point scale, see `sg` below)
  uint64_t pend_live = 0;                   // exact form: the pixels either hit of the pair is live on (wave-uniform)
  auto flush_pair = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4_2d d = {0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (kExact) {
      float4 av[4];
#pragma unroll
      for (int m = 0; m < 4; m++) av[m] = *reinterpret_cast<const float4 *>(&xm2[abase + 16 * m]);
      // MFMAs 4 m .. 4 m + 3 cover the pixels 16 m .. 16 m + 15 (two rows of the block).  A surfel's footprint is a few pixels
      // across: where neither hit of the pair is live on those two rows every A value is an exact zero and the four fp32
      // MFMAs (34 cycles each at the vector rate) are skipped - a wave-uniform branch on the pair's live-pixel mask.
#ifndef SCORP_2D_EXACT_NOSKIP
#define SCORP_2D_EXACT_NOSKIP 0
#endif
#pragma unroll
      for (int m = 0; m < 4; m++) {
        if (SCORP_2D_EXACT_NOSKIP || ((pend_live >> (16 * m)) & 0xFFFFull) != 0) {
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].x, bb[4 * m], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].y, bb[4 * m + 1], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].z, bb[4 * m + 2], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].w, bb[4 * m + 3], d, 0, 0, 0);
        }
      }
    } else {
      Frag2 af[4];
#pragma unroll
      for (int m = 0; m < 4; m++) af[m].q = *reinterpret_cast<const uint4 *>(&xm2[abase + 16 * m]);
#if SCORP_2D_MFMA_CHAINS == 2   // two independent accumulation chains of two MFMAs each (latency) + four adds
      f32x4_2d d1 = {0.0f, 0.0f, 0.0f, 0.0f};
      d = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0].v, bh[0].v, d, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1].v, bh[1].v, d1, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2].v, bh[2].v, d, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[3].v, bh[3].v, d1, 0, 0, 0);
      d += d1;
#else
#pragma unroll
      for (int m = 0; m < 4; m++) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m].v, bh[m].v, d, 0, 0, 0);
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();   // every lane has its A operands: the result tile may overwrite the matrix
#pragma unroll
    for (int i = 0; i < 4; i++) dbuf[(4 * (lane >> 4) + i) * k2DStride + (lane & 15)] = d[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int hh = lane >> 5;
    if ((lane & 31) < kAcc2Stride && hh < pend) {   // lanes 0..19: the first hit of the pair, 32..51: the second
      const int sl = hh ? pend_s1 : pend_s0;
      const float beta = q7[sl * 8 + ro_beta];
      const float *row = dbuf + (ro.row + 8 * hh) * k2DStride;
      float v = ro.alpha * row[ro.c1] + beta * row[ro.c2];
      if constexpr (!kExact) v = v * (ro_unscale * q7[sl * 8 + ro_us]) * (hh ? pend_iw1 : pend_iw0);
      if constexpr (kDet) {   // (q_id holds the pair's ordinal; one beyond the reservation: an overflowed view, discarded anyway)
        const uint32_t pair = q_id[sl];
        if (pair < capacity) {
          const size_t r = (size_t)pair * 4u + (uint32_t)quad;
          partial[r * kAcc2Stride + (lane & 31)] = v;   // whole rows, zeros included: nothing was cleared beforehand
          if ((lane & 31) == 0) row_flags[r] = 1;
        }
      } else {
        atomicAdd(acc + (size_t)q_id[sl] * kAcc2Stride + (lane & 31), v);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the next hits' rows overwrite the result tile
    pend = 0;
  };
  float T = T_final, R = 0.0f, s_last = 0.0f, last_alpha = 0.0f, last_dL_dT = 0.0f;
  // Block floating point per hit.  The fp16 pair x = h1 + h2 has an ABSOLUTE floor (2^-24, the last denormal): against the
  // block's largest upstream gradient scaled into [1, 2) a hit at the rim of a faint surfel (alpha ~ 0.004, T ~ 0.3: values
  // ~ 1e-3) was left with 14 bits, a hit deep in a list with fewer - invisible in a tensor's max norm, percent-level in
  // the row of a surfel that only such pixels see (2-D fuzz, seeds 63 / 64: tests/test_gs2d_gpu.py).  All eight values
  // are linear in the three per-pixel roots (t, dL/dz, w); the largest |root| over the hit's pixels (one wave maximum)
  // gives the power of two `sg` that brings it into [2^4, 2^5) - wave-uniform, exact, undone at the read-out - so the
  // pair keeps its 22 bits down to 2^-17 of the hit's largest value, whatever T, alpha or the depth scale are.
  // Range: |dp| <= 12 |t| (|1 / p.z| Sh <= 4, |s| <= 3), |dp2| <= 36 |t| + 400 |dL/dz| (depth <= kFarZ): 1.4e4 at most;
  // w * 2^10 <= 2^15.
  // gathers software-pipelined two chunks deep over the block's hit list (left by the forward), back to front: lane l of
  // the chunk that starts dn hits from the end takes the hit at 0-based position todo - 1 - dn - l
  const uint32_t *my_hits = hits + (size_t)quad * capacity + beg;
  auto fetch_idx = [&](uint32_t dn, bool &hit_, uint32_t &id_) {
    hit_ = lane < k2BChunk && dn + lane < todo;
    id_ = hit_ ? my_hits[todo - 1 - dn - lane] : 0u;
  };
  auto fetch_rec = [&](bool hit_, uint32_t id_, float4 &a0_, float4 &a1_, float4 &a2_, float4 &a3_, float4 &a4_) {
    if (hit_) {
      const float4 *src = reinterpret_cast<const float4 *>(rec + id_);
      a0_ = src[0]; a1_ = src[1]; a2_ = src[2]; a3_ = src[3]; a4_ = src[4];
    }
  };
  bool hit, hit1;
  uint32_t id, id1;
  float4 r0, r1, r2, r3, r4;
  fetch_idx(0, hit, id);
  fetch_rec(hit, id, r0, r1, r2, r3, r4);
  fetch_idx(k2BChunk, hit1, id1);
  for (uint32_t done_n = 0; done_n < todo; done_n += k2BChunk) {
    const uint32_t top = todo - 1 - done_n;
    float4 nx0, nx1, nx2, nx3, nx4;   // next chunk's records
    fetch_rec(hit1, id1, nx0, nx1, nx2, nx3, nx4);
    bool hit2;
    uint32_t id2;
    fetch_idx(done_n + 2 * k2BChunk, hit2, id2);
    __builtin_amdgcn_wave_barrier();   // every lane is past the previous chunk's reads of the ring
    if (hit) {   // (the chunk's hits are lanes 0 .. cnt-1: a hit list has no gaps)
      const int qi = lane;
      const SurfelLin L = surfel_lin(r0, r1, r2, bxc, byc);
      q0[qi] = L.e0; q1[qi] = L.e1; q2[qi] = L.e2; q3[qi] = L.e3; q4[qi] = r3;
      // (ox, oy) = block centre - the surfel's accumulation point (its centre clamped into the image)
      q5[qi] = make_float2(r4.x, r4.y);
      float sh_stage = 1.0f, ish_stage = 1.0f;
      if constexpr (kDet) {
        // the (surfel, tile) pair's ordinal, surfel-major: pair_base[id] + the rank of this tile among the tiles the surfel
        // reaches (for_each_tile's order: the set bits of its mask, or its whole rectangle row by row)
        const uint4 raw = reinterpret_cast<const uint4 *>(bin)[id];
        const BinRec br = *reinterpret_cast<const BinRec *>(&raw);
        const uint64_t mk = tile_mask[id];
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const uint32_t rank = mk == kMaskAll ? (uint32_t)((ty - br.y0) * (br.x1 - br.x0) + (tx - br.x0))
                                             : (uint32_t)__builtin_popcountll(mk & ((1ull << ((ty - br.y0) * 8 + (tx - br.x0))) - 1ull));
        q_id[qi] = pair_base[id] + rank;
      } else {
        q_id[qi] = id;
      }
      q_pos[qi] = top - (uint32_t)lane + 1u;
      // The power of two below |p.z| at the block centre (L.e2.x): 1 / p.z times it stays within [1/3, 4] over the block as
      // long as p.z = e2.x + e0.z qx + e1.y qy (|qx|, |qy| <= 3.5) stays within half of its centre value.  Where it does not - a
      // large surfel whose horizon passes near the block - the centre says nothing about the pixels that count, and the
      // scale is taken from the hit's largest |1 / p.z| over its valid pixels when the hit is replayed (Sh = 0 asks for it).
#ifndef SCORP_2D_STEADY_FRAC
#define SCORP_2D_STEADY_FRAC 0.5f
#endif
      if constexpr (!kExact) {
        const float pzc = fabsf(L.e2.x);
        const bool steady = 3.5f * (fabsf(L.e0.z) + fabsf(L.e1.y)) <= SCORP_2D_STEADY_FRAC * pzc;
        const uint32_t eb = min(max(__float_as_uint(pzc) & 0x7F800000u, 0x10000000u), 0x6F000000u);
        sh_stage = steady ? __uint_as_float(eb) : 0.0f;
        ish_stage = __uint_as_float(0x7F000000u - eb);
        q6[qi] = sh_stage;
      }
      float4 *tb = reinterpret_cast<float4 *>(q7 + 8 * qi);
      tb[0] = make_float4(0.0f, 1.0f, bxc - fminf(fmaxf(r2.y, 0.0f), (float)(W - 1)), byc - fminf(fmaxf(r2.z, 0.0f), (float)(H - 1)));
      tb[1] = make_float4(kFilterInvSq * (L.e2.z - bxc), kFilterInvSq * (L.e2.w - byc), ish_stage, 1.0f);
    }
    const int cnt = (int)min(todo - done_n, (uint32_t)k2BChunk);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int s_ = 0; s_ < cnt; s_++) {
      int s = s_;
      asm volatile("" : "+v"(s));   // one VGPR slot index: the ds_reads below share it instead of re-moving SGPR bases
      const uint32_t pos1 = q_pos[s];
      const float4 a0 = q0[s], a1 = q1[s], a2 = q2[s], a3 = q3[s];
      Eval2 h;
      const bool valid = eval_surfel(a0, a1, a2, a3, qxb, qyb, pxf, pyf, h) & (pos1 <= last);
      const uint64_t live = __ballot(valid);
      if (live == 0) continue;
      // The per-pixel recurrence runs under `valid`; it leaves three scalars (blend weight w, t = dL/dG * (-G), dL/dz)
      // that are zero on the other lanes, and the twenty sums are formed from them outside the branch with the
      // geometry zeroed where it is not used (pz ~ 0 makes s, rz, depth non-finite there), so no lane ever needs
      // its twenty accumulators cleared first.
      float w = 0.0f, t = 0.0f, dL_dz = 0.0f;
      const float4 nr = q4[s];
      const float2 gb = q5[s];
      if (valid) {
        const float rinv = __builtin_amdgcn_rcpf(1.0f - h.alpha);
        T *= rinv;
        w = h.alpha * T;
        // the "blended behind" recurrences (colour, depth, alpha, normal) only ever appear dotted with this pixel's
        // upstream gradient, so one scalar recurrence carries them all (see gs3d_backward.hip)
        R = last_alpha * (s_last - R) + R;
        float sc = nr.w * dpix0 + gb.x * dpix1 + gb.y * dpix2;
        if (kHasMap) sc += h.depth * ddep + dacc + nr.x * dn0 + nr.y * dn1 + nr.z * dn2;
        float dL_dal = sc - R;
        s_last = sc;
        if (kHasMap) {
          // distortion: dL/dweight = dreg (D2 + m^2 A - 2 m D) = fd2r + m (u - fdr), u = m far - fdr (also the depth term's factor)
          const float rd = h.rdepth;
          const float m_d = __builtin_fmaf(-fn * kNearZ, rd, fn);
          const float dmd_dd = ((kFarZ * kNearZ / (kFarZ - kNearZ)) * rd) * rd;
          const float u = __builtin_fmaf(m_d, far_, -fdr_);
          const float dL_dweight = __builtin_fmaf(m_d, u - fdr_, fd2r_);
          dL_dz = (pos1 == med_c) ? dmed : 0.0f;
          dL_dal += dL_dweight - last_dL_dT;
          last_dL_dT = dL_dweight * h.alpha + (1.0f - h.alpha) * last_dL_dT;
          dL_dz = __builtin_fmaf(w + w, u * dmd_dd, dL_dz);
          dL_dz = __builtin_fmaf(w, ddep, dL_dz);
        }
        dL_dal *= T;
        last_alpha = h.alpha;
        dL_dal = __builtin_fmaf(-tf_bg, rinv, dL_dal);
        t = -h.Go * dL_dal;          // dL/dG * (-G), G = Go / opacity
      }
      float inv_sg = 1.0f;
      if constexpr (!kExact) {   // the hit's power-of-two scale (see above); a hit whose roots are all zero or denormal keeps 1
        const uint32_t eb = wave_max_u32(__float_as_uint(fmaxf(fmaxf(fabsf(t), fabsf(dL_dz)), w))) >> 23;
        const uint32_t sb = eb == 0u ? 127u : min(258u - eb, 250u);   // 2^(4 - (eb - 127)), biased
        const float sg = __uint_as_float(sb << 23);
        inv_sg = __uint_as_float((254u - sb) << 23);
        t *= sg; dL_dz *= sg; w *= sg;
      }
      // The eight per-pixel values of this hit (zero on the lanes it does not touch), as two fp16 terms each, into the
      // matrix half `pend`; the products with the bases and the sums over the block's pixels are the matrix cores' work.
      {
        const bool u3 = valid & h.use3d;
        const float s0 = u3 ? h.s0 : 0.0f, s1 = u3 ? h.s1 : 0.0f, dep = u3 ? h.depth : 0.0f;
        float Sh = 1.0f;
        if constexpr (!kExact) Sh = q6[s];                                  // (wave-uniform: one LDS broadcast)
        if (!kExact && Sh == 0.0f) {   // no steady scale for this hit (see the staging): 2^-e of its largest |1 / p.z| over the valid 3-D pixels
          const uint32_t em = min(max(wave_max_u32(u3 ? (__float_as_uint(h.rz) & 0x7F800000u) : 0u), 0x10000000u), 0x6F000000u);
          Sh = __uint_as_float(0x7F000000u - em);
          if (lane == 0) q7[8 * s_ + 6] = __uint_as_float(em);                // the read-out unscales with it
        }
        const float rzs = u3 ? h.rz * Sh : 0.0f;                            // 1 / p.z times the hit's power-of-two scale
        const float t2 = h.use3d ? 0.0f : t, z2 = h.use3d ? 0.0f : dL_dz;   // low-pass branch (t, dL_dz are 0 if !valid)
        const float tr = t * rzs;
        const float dp0 = tr * s0, dp1 = tr * s1;
        const float zr = dL_dz * rzs;                                       // depth = D / pz
        const float dp2 = -(dp0 * s0 + dp1 * s1) - zr * dep;
        uint32_t *rowp = xm2 + (8 * pend) * k2XStride + lane;
        auto term = [](float x) { return kExact ? __float_as_uint(x) : split_one(x); };
        rowp[0] = term(dp0); rowp[k2XStride] = term(dp1);
        rowp[2 * k2XStride] = term(dp2); rowp[3 * k2XStride] = term(zr);
        rowp[4 * k2XStride] = term(z2); rowp[5 * k2XStride] = term(t2);
        rowp[6 * k2XStride] = term(t); rowp[7 * k2XStride] = term(w * kWCarry);
      }
      if (pend == 0) { pend_s0 = s_; pend_iw0 = inv_sg; pend_live = live; } else { pend_s1 = s_; pend_iw1 = inv_sg; pend_live |= live; }
      pend++;
      if (pend == 2) flush_pair();
    }
    if (pend) flush_pair();   // (before the next chunk's staging overwrites the slots the read-out looks at)
    hit = hit1; id = id1; r0 = nx0; r1 = nx1; r2 = nx2; r3 = nx3; r4 = nx4;
    hit1 = hit2; id1 = id2;
  }
}

// ---------------------------------------------------------------------------------------------------------
// LIN: full workgroups of the training layout.  Everything a surfel needs (radius and depth words, the 20-float
// accumulator row, the nine floats of its transform, its parameters) is requested unconditionally BEFORE the SH stream is
// issued, so that one memory latency covers it all instead of the chain radius -> visible? -> rows -> parameters behind
// the stream.  Plain 16-byte loads here: the 44 dwords as untracked one-dword loads (the forward's way, which would also
// let the maths start before the stream has landed) cost more in the texture pipe than they gain (121 vs 118 us).
template <int DEG, bool SPLIT, bool LIN>
__device__ __forceinline__ void preprocess2d_backward_body(const Pg2Args &a, float *s_sh, const Surfel *__restrict__ rec,
                                                           const BinRec *__restrict__ bin, const float *__restrict__ acc,
                                                           const ScorpGs3dGrads &g, const AdamEpi &ad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = LIN || i < a.N;
  const size_t i0 = (size_t)blockIdx.x * 256;
  const int nrows = LIN ? 256 : min(256, a.N - (int)i0);
  constexpr int NFL = 3 * (DEG + 1) * (DEG + 1);
  // LIN: pre = means 0-2, rotation 3-6, scale 7-8, opacity 10 | acc_lo = accumulators 0-10 | acc_hi = accumulators
  // 11-19 in slots 0-8, radius word in slot 10 | rec_lo = transform 0-8, depth word in slot 10
  float pre[11], acc_lo[11], acc_hi[11], rec_lo[11];
  float lin_cx = 0.0f, lin_cy = 0.0f, lin_op = 1.0f;   // LIN: the surfel's centre and opacity (record r2.y, r2.z, r2.w)
  int32_t rad_bits = 0;
  if constexpr (LIN) {
    static_assert(kAcc2Stride == 20, "accumulator row of 20 floats");
    const float4 *ap = reinterpret_cast<const float4 *>(acc + (size_t)i * kAcc2Stride);
    const float4 *rp = reinterpret_cast<const float4 *>(rec + i);
    const uint4 bw = reinterpret_cast<const uint4 *>(bin)[i];
    const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3], a4 = ap[4], r0 = rp[0], r1 = rp[1];
    const float4 r2v = rp[2];   // Tw.z, cx, cy, (opacity)
    const float4 q_in = reinterpret_cast<const float4 *>(a.rotations)[i];
    const float s_in0 = a.scales[2 * (size_t)i], s_in1 = a.scales[2 * (size_t)i + 1];
    pre[0] = a.means3D[3 * (size_t)i]; pre[1] = a.means3D[3 * (size_t)i + 1]; pre[2] = a.means3D[3 * (size_t)i + 2];
    pre[10] = a.opacities[i];
    // (nontemporal unless the optimizer step in the epilogue reads the rows again: see preprocess_backward_body, gs3d_pergaussian.hip)
    if (SCORP_NT_SH && !(SPLIT && ad.on != 0)) stage_sh_linear_async<2>(s_sh, a.shs, a.shs_rest, i0);
    else stage_sh_linear_async<0>(s_sh, a.shs, a.shs_rest, i0);
    pre[3] = q_in.x; pre[4] = q_in.y; pre[5] = q_in.z; pre[6] = q_in.w; pre[7] = s_in0; pre[8] = s_in1;
    acc_lo[0] = a0.x; acc_lo[1] = a0.y; acc_lo[2] = a0.z; acc_lo[3] = a0.w; acc_lo[4] = a1.x; acc_lo[5] = a1.y; acc_lo[6] = a1.z;
    acc_lo[7] = a1.w; acc_lo[8] = a2.x; acc_lo[9] = a2.y; acc_lo[10] = a2.z;
    acc_hi[0] = a2.w; acc_hi[1] = a3.x; acc_hi[2] = a3.y; acc_hi[3] = a3.z; acc_hi[4] = a3.w; acc_hi[5] = a4.x; acc_hi[6] = a4.y;
    acc_hi[7] = a4.z; acc_hi[8] = a4.w;
    rec_lo[0] = r0.x; rec_lo[1] = r0.y; rec_lo[2] = r0.z; rec_lo[3] = r0.w; rec_lo[4] = r1.x; rec_lo[5] = r1.y; rec_lo[6] = r1.z;
    rec_lo[7] = r1.w; rec_lo[8] = r2v.x;
    lin_cx = r2v.y; lin_cy = r2v.z; lin_op = r2v.w;
    rec_lo[10] = __uint_as_float(bw.z);   // BinRec: x0y0 | x1y1 | depth word | radius word
    acc_hi[10] = __uint_as_float(bw.w);
    rad_bits = __float_as_int(acc_hi[10]);
  } else {
    rad_bits = active ? bin[i].radius : 0;
  }
  const bool visible = (rad_bits & kRadiusMask) != 0;
  // the optimizer step inside the view (scorp_gs2d_train_view with `adam`; see preprocess_backward_body in gs3d_pergaussian.hip)
  const bool adam_on = SPLIT && ad.on != 0;
  const bool adam_sh = adam_on && (ad.m[1] != nullptr || ad.m[2] != nullptr);
  const bool want_sh_grad = a.shs != nullptr && (g.shs != nullptr || adam_sh);
  bool staged = false;
  constexpr bool lin = LIN;
  if (a.shs) {
    if constexpr (lin) {
      staged = true;
    } else {
      if (__syncthreads_or(visible ? 1 : 0)) {
        stage_sh_rows<NFL, SPLIT>(s_sh, a.shs, a.shs_rest, a.K, i0, nrows);
        staged = true;
      }
      __syncthreads();
    }
  }
  float shx = 0, shy = 0, shz = 0, shinv = 0, gr3[3] = {0, 0, 0};
  float vm[16], pm[16];
#pragma unroll
  for (int q = 0; q < 16; q++) { vm[q] = ((const CFloat *)a.view)[q]; pm[q] = ((const CFloat *)a.proj)[q]; }   // scalar cache
  float gm[3] = {0, 0, 0}, gs[2] = {0, 0}, gq[4] = {0, 0, 0, 0}, gT[9], gm2[2] = {0, 0}, grgb[3] = {0, 0, 0}, g_op = 0;
#pragma unroll
  for (int q = 0; q < 9; q++) gT[q] = 0.0f;
  const ShRow row = sh_row(s_sh, threadIdx.x, lin);
  if (visible) {
    float4 a0, a1, a2, a3, a4, r0, r1, r2;
    if constexpr (LIN) {
      a0 = make_float4(acc_lo[0], acc_lo[1], acc_lo[2], acc_lo[3]); a1 = make_float4(acc_lo[4], acc_lo[5], acc_lo[6], acc_lo[7]);
      a2 = make_float4(acc_lo[8], acc_lo[9], acc_lo[10], acc_hi[0]); a3 = make_float4(acc_hi[1], acc_hi[2], acc_hi[3], acc_hi[4]);
      a4 = make_float4(acc_hi[5], acc_hi[6], acc_hi[7], acc_hi[8]);
      r0 = make_float4(rec_lo[0], rec_lo[1], rec_lo[2], rec_lo[3]); r1 = make_float4(rec_lo[4], rec_lo[5], rec_lo[6], rec_lo[7]);
      r2 = make_float4(rec_lo[8], lin_cx, lin_cy, lin_op);
    } else {
      const float4 *ap = reinterpret_cast<const float4 *>(acc + (size_t)i * kAcc2Stride);
      a0 = ap[0]; a1 = ap[1]; a2 = ap[2]; a3 = ap[3]; a4 = ap[4];
      const float4 *rp = reinterpret_cast<const float4 *>(rec + i);
      r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
    }
    const float ga[3] = {a0.x, a0.y, a0.z}, gb[3] = {a0.w, a1.x, a1.y}, gc[3] = {a1.z, a1.w, a2.x};
    const float gD = a2.y, gTw2 = a2.z, gx = a2.w, gy = a3.x;
    const float gn[3] = {a3.y, a3.z, a3.w};
    g_op = a4.x / r2.w;   // the blend kernel summed opacity * G * dL/dalpha
    grgb[0] = a4.y; grgb[1] = a4.z; grgb[2] = a4.w;
    const float Tu[3] = {r0.x, r0.y, r0.z}, Tv[3] = {r0.w, r1.x, r1.y}, Tw[3] = {r1.z, r1.w, r2.x};
    // the blend kernels hand over the gradients of the linear form p = (x - ex) pa + (y - ey) pb + pc, depth = D / p.z
    // (see surfel_lin): pa = Tv x Tw, pb = Tw x Tu, pc = kc x lc with kc = ex Tw - Tu, lc = ey Tw - Tv, D = Tu . pa,
    // (ex, ey) = the centre clamped into the image, held fixed.  For c = u x v: dL/du = v x g, dL/dv = g x u.
    const float ecx = fminf(fmaxf(r2.y, 0.0f), (float)(a.W - 1)), ecy = fminf(fmaxf(r2.z, 0.0f), (float)(a.H - 1));
    float kc[3], lc[3], gkc[3], glc[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { kc[q] = __builtin_fmaf(ecx, Tw[q], -Tu[q]); lc[q] = __builtin_fmaf(ecy, Tw[q], -Tv[q]); }
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int j = (q + 1) % 3, k = (q + 2) % 3;
      gkc[q] = lc[j] * gc[k] - lc[k] * gc[j];   // d/dkc = lc x gc
      glc[q] = gc[j] * kc[k] - gc[k] * kc[j];   // d/dlc = gc x kc
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int j = (q + 1) % 3, k = (q + 2) % 3;
      const float pa_q = Tv[j] * Tw[k] - Tv[k] * Tw[j], pb_q = Tw[j] * Tu[k] - Tw[k] * Tu[j], tuv_q = Tu[j] * Tv[k] - Tu[k] * Tv[j];
      gT[q] = (gb[j] * Tw[k] - gb[k] * Tw[j]) - gkc[q] + gD * pa_q;                                  // d/dTu
      gT[3 + q] = (Tw[j] * ga[k] - Tw[k] * ga[j]) - glc[q] + gD * pb_q;                              // d/dTv
      gT[6 + q] = (ga[j] * Tv[k] - ga[k] * Tv[j]) + (Tu[j] * gb[k] - Tu[k] * gb[j]) + (ecx * gkc[q] + ecy * glc[q])
                  + gD * tuv_q;                                                                      // d/dTw (dD/dTw = Tu x Tv)
    }
    gT[8] += gTw2;
    const float depth = LIN ? rec_lo[10] : __uint_as_float(bin[i].depth_bits);
    gm2[0] = gT[2] * depth * 0.5f * a.W;   // densification statistic (gs2dgs/scene/gaussian_model.py:495 consumes it)
    gm2[1] = gT[5] * depth * 0.5f * a.H;
    if (gx != 0.0f || gy != 0.0f) {        // the low-pass centre is the centre of the 3-sigma box, a function of T
      const float c2 = kCutoff * kCutoff;
      const float t[3] = {c2, c2, -1.0f};
      const float dd = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
      float dLdd = 0.0f;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const float f = t[q] / dd;
        gT[q] += gx * f * Tw[q];
        gT[3 + q] += gy * f * Tw[q];
        gT[6 + q] += gx * f * Tu[q] + gy * f * Tv[q];
        dLdd += (gx * Tu[q] * Tw[q] + gy * Tv[q] * Tw[q]) * f;
      }
      dLdd *= -1.0f / dd;
#pragma unroll
      for (int q = 0; q < 3; q++) gT[6 + q] += dLdd * 2.0f * t[q] * Tw[q];
    }
    if (a.raw & 1) {
      const float o = act_opacity(LIN ? pre[10] : a.opacities[i], a.raw);
      g_op *= o * (1.0f - o);
    }
    float p0, p1, p2;
    if constexpr (LIN) { p0 = pre[0]; p1 = pre[1]; p2 = pre[2]; }
    else { p0 = a.means3D[3 * (size_t)i]; p1 = a.means3D[3 * (size_t)i + 1]; p2 = a.means3D[3 * (size_t)i + 2]; }
    if (LIN || !a.transmat) {
      float Q[3][4], R[9], inv_qn;
      pixel_rows(pm, a.W, a.H, Q);
      const float4 q_in = LIN ? make_float4(pre[3], pre[4], pre[5], pre[6]) : reinterpret_cast<const float4 *>(a.rotations)[i];
      const float4 qn = act_quat(q_in, a.raw, &inv_qn);
      quat_R(qn, R);
      const float sa0 = act_scale(LIN ? pre[7] : a.scales[2 * (size_t)i], a.raw),
                  sa1 = act_scale(LIN ? pre[8] : a.scales[2 * (size_t)i + 1], a.raw);
      const float sx = a.scale_mod * sa0, sy = a.scale_mod * sa1;
      float gh[3][3];
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int c = 0; c < 3; c++) gh[j][c] = gT[0 * 3 + j] * Q[0][c] + gT[1 * 3 + j] * Q[1][c] + gT[2 * 3 + j] * Q[2][c];
#pragma unroll
      for (int c = 0; c < 3; c++) gm[c] = gh[2][c];
      const float mult = ((rad_bits >> kFlipBit) & 1) ? -1.0f : 1.0f;
      float gtn[3];
#pragma unroll
      for (int c = 0; c < 3; c++) gtn[c] = mult * (vm[c * 4 + 0] * gn[0] + vm[c * 4 + 1] * gn[1] + vm[c * 4 + 2] * gn[2]);
      float gR[9];
#pragma unroll
      for (int r = 0; r < 3; r++) { gR[r * 3 + 0] = gh[0][r] * sx; gR[r * 3 + 1] = gh[1][r] * sy; gR[r * 3 + 2] = gtn[r]; }
      gs[0] = a.scale_mod * (gh[0][0] * R[0] + gh[0][1] * R[3] + gh[0][2] * R[6]);
      gs[1] = a.scale_mod * (gh[1][0] * R[1] + gh[1][1] * R[4] + gh[1][2] * R[7]);
      if (a.raw & 2) { gs[0] *= sa0; gs[1] *= sa1; }
      const float r_ = qn.x, x = qn.y, y = qn.z, z = qn.w;
      gq[0] = 2 * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
      gq[1] = 2 * (y * gR[1] + z * gR[2] + y * gR[3] - 2 * x * gR[4] - r_ * gR[5] + z * gR[6] + r_ * gR[7] - 2 * x * gR[8]);
      gq[2] = 2 * (-2 * y * gR[0] + x * gR[1] + r_ * gR[2] + x * gR[3] + z * gR[5] - r_ * gR[6] + z * gR[7] - 2 * y * gR[8]);
      gq[3] = 2 * (-2 * z * gR[0] - r_ * gR[1] + x * gR[2] + r_ * gR[3] - 2 * z * gR[4] + y * gR[5] + x * gR[6] + y * gR[7]);
      if (a.raw & 4) {
        const float dotq = r_ * gq[0] + x * gq[1] + y * gq[2] + z * gq[3];
        gq[0] = (gq[0] - r_ * dotq) * inv_qn; gq[1] = (gq[1] - x * dotq) * inv_qn;
        gq[2] = (gq[2] - y * dotq) * inv_qn; gq[3] = (gq[3] - z * dotq) * inv_qn;
      }
    }
    if (a.shs) {   // the SH part itself runs after the rows have landed in LDS (below)
      const float d0 = p0 - ((const CFloat *)a.campos)[0], d1 = p1 - ((const CFloat *)a.campos)[1], d2_ = p2 - ((const CFloat *)a.campos)[2];
      shinv = 1.0f / sqrtf(d0 * d0 + d1 * d1 + d2_ * d2_);
      shx = d0 * shinv; shy = d1 * shinv; shz = d2_ * shinv;
#pragma unroll
      for (int ch = 0; ch < 3; ch++) gr3[ch] = ((rad_bits >> (kClampShift + ch)) & 1) ? 0.0f : grgb[ch];
    }
  }
  if (a.shs && lin) { stage_sh_wait(); __syncthreads(); }
  AdamGeomMoments am;   // asked for here, used in the epilogue: the SH phase in between hides the latency
  if (adam_on && active && !a.transmat) {
    adam_moments_load<3>(ad, 0, 3 * (size_t)i, am.m, am.v);
    adam_moments_load<1>(ad, 3, (size_t)i, am.m + 3, am.v + 3);
    adam_moments_load<2>(ad, 4, 2 * (size_t)i, am.m + 4, am.v + 4);
    adam_moments_load<4>(ad, 5, 4 * (size_t)i, am.m + 7, am.v + 7);
  }
  if (visible && a.shs) {
    float gdir[3] = {0, 0, 0};
    sh_row_backward<DEG>(row, shx, shy, shz, gr3, want_sh_grad, gdir);
    const float dot = shx * gdir[0] + shy * gdir[1] + shz * gdir[2];
    gm[0] += (gdir[0] - shx * dot) * shinv; gm[1] += (gdir[1] - shy * dot) * shinv; gm[2] += (gdir[2] - shz * dot) * shinv;
  } else if (want_sh_grad && active) {
    sh_row_zero(row);
  }
  if (active) {
    if (g.means3D) { g.means3D[3 * (size_t)i] = gm[0]; g.means3D[3 * (size_t)i + 1] = gm[1]; g.means3D[3 * (size_t)i + 2] = gm[2]; }
    if (g.means2D) { g.means2D[3 * (size_t)i] = gm2[0]; g.means2D[3 * (size_t)i + 1] = gm2[1]; g.means2D[3 * (size_t)i + 2] = 0.0f; }
    if (g.colors_precomp) { g.colors_precomp[3 * (size_t)i] = grgb[0]; g.colors_precomp[3 * (size_t)i + 1] = grgb[1]; g.colors_precomp[3 * (size_t)i + 2] = grgb[2]; }
    if (g.opacities) g.opacities[i] = g_op;
    if (g.scales) { g.scales[2 * (size_t)i] = gs[0]; g.scales[2 * (size_t)i + 1] = gs[1]; }
    if (g.rotations) reinterpret_cast<float4 *>(g.rotations)[i] = make_float4(gq[0], gq[1], gq[2], gq[3]);
    if (g.cov3D_precomp) {
#pragma unroll
      for (int q = 0; q < 9; q++) g.cov3D_precomp[9 * (size_t)i + q] = gT[q];
    }
  }
  // Adam + the view's densification statistics where the gradient row is at hand (train_2dgs.py:189-199 over
  // gs2dgs/scene/gaussian_model.py:494-495: the statistic norms the whole means2D-gradient row, whose third component is 0)
  bool adam_go = false;
  if constexpr (SPLIT) {
    if (adam_on) {
      adam_go = !(ad.skip && *ad.skip != 0u);
      if (!adam_go && ad.skipped_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ad.skipped_counter, 1u);
      if (adam_go && active && !a.transmat) {
        float pin[11];
        if constexpr (LIN) {
#pragma unroll
          for (int q = 0; q < 11; q++) pin[q] = pre[q];
        }
        adam_leaf_pre<3>(ad, 0, const_cast<float *>(a.means3D), 3 * (size_t)i, gm, LIN ? pin : nullptr, am.m, am.v);
        adam_leaf_pre<1>(ad, 3, const_cast<float *>(a.opacities), (size_t)i, &g_op, LIN ? pin + 10 : nullptr, am.m + 3, am.v + 3);
        adam_leaf_pre<2>(ad, 4, const_cast<float *>(a.scales), 2 * (size_t)i, gs, LIN ? pin + 7 : nullptr, am.m + 4, am.v + 4);
        adam_leaf_pre<4>(ad, 5, const_cast<float *>(a.rotations), 4 * (size_t)i, gq, LIN ? pin + 3 : nullptr, am.m + 7, am.v + 7);
        if (ad.accum && visible) {
#pragma clang fp contract(off)
          const float gx = gm2[0], gy = gm2[1];
          ad.max_radii2D[i] = fmaxf(ad.max_radii2D[i], (float)(rad_bits & kRadiusMask));
          ad.accum[i] += sqrtf(gx * gx + gy * gy + 0.0f);
          ad.denom[i] += 1.0f;
        }
      }
    }
  }
  if (want_sh_grad) {
    if (!staged && active) sh_row_zero(row);
    __syncthreads();
    if (g.shs) {
      if (lin) unstage_sh_linear(s_sh, g.shs, g.shs_rest, i0);
      else unstage_sh_rows<SPLIT>(s_sh, g.shs, g.shs_rest, a.K, i0, nrows);
    }
    if constexpr (SPLIT) {
      if (adam_go && adam_sh) {
        if (lin) adam_sh_linear(ad, s_sh, const_cast<float *>(a.shs), const_cast<float *>(a.shs_rest), i0);
        else adam_sh_rows(ad, s_sh, const_cast<float *>(a.shs), const_cast<float *>(a.shs_rest), a.K, i0, nrows);
      }
    }
  }
}

template <int DEG, bool SPLIT>
__global__ void __launch_bounds__(256)
preprocess2d_backward_kernel(Pg2Args a, const Surfel *__restrict__ rec, const BinRec *__restrict__ bin,
                             const float *__restrict__ acc, ScorpGs3dGrads g, AdamEpi ad) {
  __shared__ __attribute__((aligned(16))) float s_sh[256 * kShStride];   // direct global->LDS loads land 16-byte words
  if constexpr (SPLIT && DEG == 3) {
    if (a.shs != nullptr && a.K == 16 && !a.transmat && a.N - (int)blockIdx.x * 256 >= 256) {
      preprocess2d_backward_body<DEG, SPLIT, true>(a, s_sh, rec, bin, acc, g, ad);
      return;
    }
  }
  preprocess2d_backward_body<DEG, SPLIT, false>(a, s_sh, rec, bin, acc, g, ad);
}

// Deterministic mode: the rows of one surfel are contiguous - partial[4 * pair_base[i] ... 4 * pair_base[i + 1]) - so the
// ordered per-surfel sum is a streaming read (gs3d_backward.hip has the 3-D twin).  Thirty-two lanes per surfel (lane = float
// of a row, twenty used) add the flagged rows in the fixed order pair, block - sixteen rows in flight per step.
__global__ void __launch_bounds__(256)
reduce_pair_rows2d_kernel(int N, const uint32_t *__restrict__ pair_base, uint32_t capacity, const uint8_t *__restrict__ row_flags,
                          const float *__restrict__ partial, float *__restrict__ acc) {
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5), col = threadIdx.x & 31;
  if (i >= N || col >= kAcc2Stride) return;
  const uint32_t r0 = min(pair_base[i], capacity) * 4u, r1 = min(pair_base[i + 1], capacity) * 4u;
  float sum = 0.0f;
  for (uint32_t r = r0; r < r1; r += 16) {
    uint32_t f[4];
#pragma unroll
    for (int p = 0; p < 4; p++) f[p] = r + 4 * p < r1 ? *reinterpret_cast<const uint32_t *>(row_flags + r + 4 * p) : 0u;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const bool on = (f[k >> 2] >> (8 * (k & 3))) & 0xFFu;
      v[k] = on ? partial[(size_t)(r + k) * kAcc2Stride + col] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) sum += v[k];   // (an absent row adds an exact zero)
  }
  acc[(size_t)i * kAcc2Stride + col] = sum;
}

Pg2Args make_args2(const ScorpGs3dInputs *in, const StateLayout &L) {
  Pg2Args a;
  a.N = in->num_gaussians; a.K = in->sh_coeffs; a.W = in->image_width; a.H = in->image_height;
  a.tiles_x = L.tiles_x; a.tiles_y = L.tiles_y; a.raw = in->raw_params; a.count_with_atomics = L.lds_binning ? 0 : 1;
  a.scale_mod = in->scale_modifier;
  a.view = in->viewmatrix; a.proj = in->projmatrix; a.campos = in->campos;
  a.means3D = in->means3D; a.shs = in->shs; a.shs_rest = in->shs_rest; a.colors_precomp = in->colors_precomp;
  a.opacities = in->opacities; a.scales = in->scales; a.rotations = in->rotations; a.transmat = in->cov3D_precomp;
  return a;
}

int validate2(const ScorpGs3dInputs *in) {
  if (!in) { set_error("inputs is NULL"); return SCORP_ERR_INVALID; }
  if (in->num_gaussians < 0 || in->image_width <= 0 || in->image_height <= 0) { set_error("bad sizes"); return SCORP_ERR_INVALID; }
  if (in->num_gaussians > 0) {
    if (!in->means3D || !in->opacities) { set_error("means3D / opacities is NULL"); return SCORP_ERR_INVALID; }
    if ((in->shs == nullptr) == (in->colors_precomp == nullptr)) { set_error("provide exactly one of shs / colors_precomp"); return SCORP_ERR_INVALID; }
    const bool sr = in->scales != nullptr && in->rotations != nullptr;
    if (sr == (in->cov3D_precomp != nullptr)) { set_error("provide exactly one of scales+rotations / precomputed transform"); return SCORP_ERR_INVALID; }
    if (in->shs && (in->sh_degree < 0 || in->sh_degree > 3 || in->sh_coeffs < (in->sh_degree + 1) * (in->sh_degree + 1))) {
      set_error("bad sh_degree / sh_coeffs"); return SCORP_ERR_INVALID;
    }
  }
  if (!in->bg || !in->viewmatrix || !in->projmatrix || !in->campos) { set_error("bg / matrices / campos is NULL"); return SCORP_ERR_INVALID; }
  if ((((uintptr_t)in->shs | (uintptr_t)in->shs_rest | (uintptr_t)in->rotations) & 15) != 0) {
    set_error("shs / shs_rest / rotations must be 16-byte aligned"); return SCORP_ERR_INVALID;
  }
  return SCORP_OK;
}

}  // namespace
}  // namespace scorp

using namespace scorp;

extern "C" size_t scorp_gs2d_state_bytes(int32_t N, int32_t W, int32_t H) { return StateLayout(N, W, H, true).total; }
extern "C" size_t scorp_gs2d_backward_scratch_bytes(int32_t N) {
  return align_up((size_t)(N > 0 ? N : 1) * kAcc2Stride * sizeof(float), 256);
}
extern "C" size_t scorp_gs2d_backward_scratch_bytes_ex(int32_t N, int32_t W, int32_t H, uint64_t capacity, uint32_t flags) {
  (void)W; (void)H;
  if (flags & SCORP_BACKWARD_DETERMINISTIC) return DetLayout(N, capacity, kAcc2Stride).total;
  return scorp_gs2d_backward_scratch_bytes(N);
}

extern "C" int scorp_gs2d_preprocess(const ScorpGs3dInputs *in, int32_t *out_radii, void *state, size_t state_bytes,
                                     scorp_stream_t stream_) {
  if (int e = validate2(in)) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const int N = in->num_gaussians;
  const StateLayout L(N, in->image_width, in->image_height, true);
  if (!state || state_bytes < L.total || ((uintptr_t)state & 255)) { set_error("state buffer NULL, misaligned or too small"); return SCORP_ERR_INVALID; }
  if (N > 0 && !out_radii) { set_error("out_radii is NULL"); return SCORP_ERR_INVALID; }
  char *base = (char *)state;
  uint32_t *tile_count = (uint32_t *)(base + L.tile_count);
  if (!L.lds_binning) SCORP_HIP_CHECK(hipMemsetAsync(tile_count, 0, ((size_t)L.tiles + 1) * 4, stream));
  if (N > 0) {
    ProfScope prof(kKPreprocess2d, stream);
    const Pg2Args a = make_args2(in, L);
    const dim3 grid((N + 255) / 256), block(256);
    const int deg = in->shs ? in->sh_degree : 0;
    Surfel *rec = (Surfel *)(base + L.rec);
    BinRec *bin = (BinRec *)(base + L.bin);
#define SCORP_L2(D, S) preprocess2d_kernel<D, S><<<grid, block, 0, stream>>>(a, rec, bin, (uint64_t *)(base + L.tile_mask), out_radii, tile_count)
    if (in->shs_rest) { switch (deg) { case 0: SCORP_L2(0, true); break; case 1: SCORP_L2(1, true); break; case 2: SCORP_L2(2, true); break; default: SCORP_L2(3, true); } }
    else { switch (deg) { case 0: SCORP_L2(0, false); break; case 1: SCORP_L2(1, false); break; case 2: SCORP_L2(2, false); break; default: SCORP_L2(3, false); } }
#undef SCORP_L2
    SCORP_KERNEL_CHECK("preprocess_2d", in->debug, stream);
  }
  return bin_count_and_scan(L, base, N, in->debug, stream);
}

static int render2d_impl(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity, float *out_color,
                         float *out_allmap, scorp_stream_t stream_, bool for_backward) {
  if (int e = validate2(in)) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const int N = in->num_gaussians, W = in->image_width, H = in->image_height;
  const StateLayout L(N, W, H, true);
  const PairLayout P(capacity);
  if (!state || ((uintptr_t)state & 255) || !pairs || ((uintptr_t)pairs & 255)) { set_error("state / pairs NULL or misaligned"); return SCORP_ERR_INVALID; }
  if (capacity > 0xFFFFFFFFull) { set_error("capacity above 2^32-1 pairs"); return SCORP_ERR_INVALID; }
  if (!out_color || !out_allmap) { set_error("output image pointer is NULL"); return SCORP_ERR_INVALID; }
  char *base = (char *)state, *pb = (char *)pairs;
  if (int e = bin_scatter_and_sort(L, P, base, pb, N, (uint32_t)capacity, in->debug, stream)) return e;
  {
    ProfScope prof(kKBlendForward2d, stream);
    auto bk = for_backward ? blend2d_forward_wave_kernel<true> : blend2d_forward_wave_kernel<false>;
    bk<<<(L.tiles + 7) / 8 * 32, 64, 0, stream>>>(
        (const uint32_t *)(base + L.tile_start), (const uint32_t *)(pb + P.list), (const Surfel *)(base + L.rec),
        (uint32_t)capacity, W, H, L.tiles_x, L.tiles, in->bg, out_color, out_allmap, (float *)(base + L.final_T),
        (uint32_t *)(base + L.n_contrib), (uint32_t *)(pb + P.hits));
  }
  SCORP_KERNEL_CHECK("blend_forward_2d", in->debug, stream);
  return SCORP_OK;
}

extern "C" int scorp_gs2d_render(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity,
                                 float *out_color, float *out_allmap, scorp_stream_t stream) {
  return render2d_impl(in, state, pairs, capacity, out_color, out_allmap, stream, true);
}

extern "C" int scorp_gs2d_render_image(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity,
                                       float *out_color, float *out_allmap, scorp_stream_t stream) {
  return render2d_impl(in, state, pairs, capacity, out_color, out_allmap, stream, false);
}

extern "C" int scorp_gs2d_backward(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                                   const float *dL_dcolor, const float *dL_dallmap, const ScorpGs3dGrads *grads,
                                   void *scratch, size_t scratch_bytes, scorp_stream_t stream_) {
  return scorp_gs2d_backward_ex(in, state, pairs, capacity, dL_dcolor, dL_dallmap, grads, scratch, scratch_bytes, 0u, stream_);
}

extern "C" int scorp_gs2d_backward_ex(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                                      const float *dL_dcolor, const float *dL_dallmap, const ScorpGs3dGrads *grads,
                                      void *scratch, size_t scratch_bytes, uint32_t flags, scorp_stream_t stream_) {
  return backward2d_impl(in, state, pairs, capacity, dL_dcolor, dL_dallmap, grads, scratch, scratch_bytes, flags, stream_, nullptr);
}

// `adam` (scorp_gs2d_train_view): the per-surfel kernel applies the optimizer step and the view's statistics itself
int scorp::backward2d_impl(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                           const float *dL_dcolor, const float *dL_dallmap, const ScorpGs3dGrads *grads, void *scratch,
                           size_t scratch_bytes, uint32_t flags, scorp_stream_t stream_, const AdamEpi *adam) {
  if (!in || !state || !pairs || !grads || !scratch || !dL_dcolor) { set_error("NULL argument to scorp_gs2d_backward"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const int N = in->num_gaussians, W = in->image_width, H = in->image_height;
  if (N <= 0) return SCORP_OK;
  const StateLayout L(N, W, H, true);
  const PairLayout P(capacity);
  const bool det = (flags & SCORP_BACKWARD_DETERMINISTIC) != 0, exact = (flags & SCORP_BACKWARD_EXACT_FP32) != 0;
  const size_t need = scorp_gs2d_backward_scratch_bytes_ex(N, W, H, capacity, flags);
  if (scratch_bytes < need || ((uintptr_t)scratch & 15)) {
    set_error("2D backward scratch too small or misaligned (%zu < %zu)", scratch_bytes, need);
    return SCORP_ERR_INVALID;
  }
  if (det && capacity * 4 > 0xFFFFFFFFull) { set_error("capacity too large for the deterministic backward"); return SCORP_ERR_INVALID; }
  const char *base = (const char *)state, *pb = (const char *)pairs;
  float *acc = (float *)scratch;
  float *partial = nullptr;
  uint8_t *row_flags = nullptr;
  uint32_t *pair_base = nullptr;
  const BinRec *bin_arr = (const BinRec *)(base + L.bin);
  const uint64_t *mask_arr = (const uint64_t *)(base + L.tile_mask);
  if (det) {
    const DetLayout DL(N, capacity, kAcc2Stride);
    char *p = (char *)scratch;
    partial = (float *)(p + DL.partial);
    row_flags = (uint8_t *)(p + DL.flags);
    pair_base = (uint32_t *)(p + DL.pair_base);
    SCORP_HIP_CHECK(hipMemsetAsync(row_flags, 0, (size_t)(capacity > 0 ? capacity : 1) * 4, stream));
    launch_pair_base(N, bin_arr, mask_arr, (uint32_t *)(p + DL.block_sums), pair_base, stream);
    SCORP_KERNEL_CHECK("pair_base", in->debug, stream);
  } else {
    SCORP_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)N * kAcc2Stride * sizeof(float), stream));
  }
  {
    ProfScope prof(kKBlendBackward2d, stream);
    const bool map = dL_dallmap != nullptr;
    auto wk = det ? (exact ? (map ? blend2d_backward_wave_kernel<true, true, true> : blend2d_backward_wave_kernel<false, true, true>)
                           : (map ? blend2d_backward_wave_kernel<true, false, true> : blend2d_backward_wave_kernel<false, false, true>))
                  : (exact ? (map ? blend2d_backward_wave_kernel<true, true, false> : blend2d_backward_wave_kernel<false, true, false>)
                           : (map ? blend2d_backward_wave_kernel<true, false, false> : blend2d_backward_wave_kernel<false, false, false>));
    wk<<<(L.tiles + 7) / 8 * 32, 64, 0, stream>>>(
        (const uint32_t *)(base + L.tile_start), (const uint32_t *)(pb + P.list), (const Surfel *)(base + L.rec),
        (uint32_t)capacity, W, H, L.tiles_x, L.tiles, in->bg, (const float *)(base + L.final_T),
        (const uint32_t *)(base + L.n_contrib), dL_dcolor, dL_dallmap, acc, (const uint32_t *)(pb + P.hits), partial, row_flags,
        pair_base, bin_arr, mask_arr);
  }
  SCORP_KERNEL_CHECK("blend_backward_2d", in->debug, stream);
  if (det) {
    reduce_pair_rows2d_kernel<<<(N + 7) / 8, 256, 0, stream>>>(N, pair_base, (uint32_t)capacity, row_flags, partial, acc);
    SCORP_KERNEL_CHECK("reduce_pair_rows_2d", in->debug, stream);
  }
  {
    ProfScope prof(kKPreprocessBackward2d, stream);
    const Pg2Args a = make_args2(in, L);
    const dim3 grid((N + 255) / 256), block(256);
    const int deg = in->shs ? in->sh_degree : 0;
    const ScorpGs3dGrads g = *grads;
    const Surfel *rec = (const Surfel *)(base + L.rec);
    const BinRec *bin = (const BinRec *)(base + L.bin);
    AdamEpi ad;
    memset(&ad, 0, sizeof(ad));
    if (adam && in->shs_rest) ad = *adam;   // (the fused step is defined for the training layout: dc / rest split leaves)
#define SCORP_B2(D, S) preprocess2d_backward_kernel<D, S><<<grid, block, 0, stream>>>(a, rec, bin, acc, g, ad)
    if (in->shs_rest) { switch (deg) { case 0: SCORP_B2(0, true); break; case 1: SCORP_B2(1, true); break; case 2: SCORP_B2(2, true); break; default: SCORP_B2(3, true); } }
    else { switch (deg) { case 0: SCORP_B2(0, false); break; case 1: SCORP_B2(1, false); break; case 2: SCORP_B2(2, false); break; default: SCORP_B2(3, false); } }
#undef SCORP_B2
  }
  SCORP_KERNEL_CHECK("preprocess_backward_2d", in->debug, stream);
  return SCORP_OK;
}

// xy[N,2], depth[N], T[N,9], normal_opacity[N,4], rgb[N,3], rect[N,4]; any may be NULL (stage-level parity tests)
extern "C" int scorp_gs2d_debug_geom(const void *state, int32_t N, int32_t W, int32_t H, float *T, float *xy, float *depth,
                                     float *normal_opacity, float *rgb, int32_t *rect, scorp_stream_t stream_) {
  if (!state) { set_error("state is NULL"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const StateLayout L(N, W, H, true);
  if (N <= 0) return SCORP_OK;
  Surfel *hrec = (Surfel *)malloc((size_t)N * sizeof(Surfel));
  BinRec *hbin = (BinRec *)malloc((size_t)N * sizeof(BinRec));
  if (!hrec || !hbin) { free(hrec); free(hbin); set_error("host allocation failed"); return SCORP_ERR_INVALID; }
  hipError_t e = hipMemcpyAsync(hrec, (const char *)state + L.rec, (size_t)N * sizeof(Surfel), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(hbin, (const char *)state + L.bin, (size_t)N * sizeof(BinRec), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { free(hrec); free(hbin); set_error("debug_geom copy failed: %s", hipGetErrorString(e)); return SCORP_ERR_HIP; }
  for (int i = 0; i < N; i++) {
    const bool vis = (hbin[i].radius & kRadiusMask) != 0;
    const Surfel z = {};
    const Surfel &s = vis ? hrec[i] : z;
    const float t9[9] = {s.r0.x, s.r0.y, s.r0.z, s.r0.w, s.r1.x, s.r1.y, s.r1.z, s.r1.w, s.r2.x};
    if (T) memcpy(T + 9 * (size_t)i, t9, sizeof(t9));
    if (xy) { xy[2 * i] = s.r2.y; xy[2 * i + 1] = s.r2.z; }
    if (depth) { uint32_t b = vis ? hbin[i].depth_bits : 0u; memcpy(depth + i, &b, 4); }
    if (normal_opacity) { normal_opacity[4 * i] = s.r3.x; normal_opacity[4 * i + 1] = s.r3.y; normal_opacity[4 * i + 2] = s.r3.z; normal_opacity[4 * i + 3] = s.r2.w; }
    if (rgb) { rgb[3 * i] = s.r3.w; rgb[3 * i + 1] = s.r4.x; rgb[3 * i + 2] = s.r4.y; }
    if (rect) { rect[4 * i] = vis ? hbin[i].x0 : 0; rect[4 * i + 1] = vis ? hbin[i].y0 : 0; rect[4 * i + 2] = vis ? hbin[i].x1 : 0; rect[4 * i + 3] = vis ? hbin[i].y1 : 0; }
  }
  free(hrec); free(hbin);
  return SCORP_OK;
}

extern "C" int scorp_gs2d_debug_tiles(const void *state, const void *pairs, uint64_t capacity, int32_t N, int32_t W,
                                      int32_t H, uint32_t *tile_start, uint32_t *point_list, scorp_stream_t stream_) {
  if (!state || !pairs) { set_error("state / pairs is NULL"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const StateLayout L(N, W, H, true);
  const PairLayout P(capacity);
  StateHeader h;
  SCORP_HIP_CHECK(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, stream));
  SCORP_HIP_CHECK(hipStreamSynchronize(stream));
  return copy_tile_lists_raster(L, P, state, pairs, capacity, h.num_pairs, tile_start, point_list, stream);
}
