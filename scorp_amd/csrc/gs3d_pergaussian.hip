// gs3d_pergaussian.hip — the two per-Gaussian (one thread per splat) kernels of the 3DGS path for gfx950:
//   preprocess          : project, cull, EWA covariance, conic, tile rectangle, SH -> RGB      (forward)
//   preprocess_backward : screen-space accumulators -> gradients of the call arguments         (backward)
// Both are HBM-streaming kernels (236 B in / 68 B out, resp. ~300 B in / 248 B out per splat).  The wide rows
// (spherical harmonics, 36..192 B per splat) go through LDS so that global accesses are coalesced 16-byte streams
// whatever the row length: a block stages its 256 rows with stride 49 floats (odd, so the later one-thread-per-row
// walks are bank-conflict free), and the backward overwrites each row in place with its gradient before streaming
// it back out.
//
// Two argument conventions share the kernels:
//   * the reference call site (gs3dgs/gaussian_renderer/__init__.py:101-109): activated opacities / scales /
//     unit quaternions and one shs[N,K,3] tensor;
//   * "raw" parameters (ScorpGs3dInputs.raw_params / shs_rest): the GaussianModel's own storage — logit opacity,
//     log scale, un-normalised quaternion, _features_dc[N,1,3] + _features_rest[N,K-1,3] — with the activations of
//     gs3dgs/scene/gaussian_model.py:126-146 applied in-kernel and differentiated in the backward, which removes the
//     torch.cat and five elementwise kernels (and their autograd mirrors) from every view.
// Arithmetic follows oracle/gs3d_oracle.c.  Contraction is off in the forward so discrete decisions round like the
// oracle.
#include <cstdlib>
#include "pergaussian.hpp"

namespace scorp {
namespace {

struct PgArgs {
  int N, K, W, H, tiles_x, tiles_y, raw, count_with_atomics;
  float tanfovx, tanfovy, scale_mod;
  const float *view, *proj, *campos;
  const float *means3D, *shs, *shs_rest, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
  uint8_t *visible;   // optional: radii > 0 as bytes (render()'s visibility_filter)
  int views;          // > 1: blockIdx.y = view; N, H, tiles_y are those of ONE view, the outputs those of the stacked image
};

// ---------------------------------------------------------------------------------------------------------
struct SplatGeom {
  bool vis;
  float tz, sx, sy, cA, cB, cC, op, kcut;
  uint64_t mask;
  int x0, y0, x1, y1, radius;
};

// Projection, EWA covariance -> conic, radius, tile rectangle and the exact tile mask of one Gaussian.
__device__ __forceinline__ SplatGeom splat_geometry(const PgArgs &a, const float *vm, const float *pm, int i, float px_,
                                                    float py_, float pz_, float4 q_raw, const float *sc_raw, float op_raw,
                                                    const float *cov3D_precomp) {
#pragma clang fp contract(off)
  bool vis = false;
  float tz = 0, sx = 0, sy = 0, cA = 0, cB = 0, cC = 0, op = 0, kcut = 0;
  uint64_t mask = 0;
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0, radius = 0;
  {
    // (all parameter loads left together at the top: one memory latency instead of means -> cull -> the rest; ~4 % of
    // the Gaussians turn out invisible and waste theirs)
    const float tx = vm[0] * px_ + vm[4] * py_ + vm[8] * pz_ + vm[12];
    const float ty = vm[1] * px_ + vm[5] * py_ + vm[9] * pz_ + vm[13];
    tz = __builtin_fmaf(vm[10], pz_, __builtin_fmaf(vm[6], py_, __builtin_fmaf(vm[2], px_, vm[14])));
    if (tz > kNearZ) {
      const float hx = pm[0] * px_ + pm[4] * py_ + pm[8] * pz_ + pm[12];
      const float hy = pm[1] * px_ + pm[5] * py_ + pm[9] * pz_ + pm[13];
      const float hw = pm[3] * px_ + pm[7] * py_ + pm[11] * pz_ + pm[15];
      const float pw = 1.0f / (hw + kWEps);
      const float ndcx = hx * pw, ndcy = hy * pw;
      float c6[6];
      if (cov3D_precomp) {
#pragma unroll
        for (int q = 0; q < 6; q++) c6[q] = cov3D_precomp[6 * (size_t)i + q];
      } else {
        float invn;
        const float4 q4 = act_quat(q_raw, a.raw, &invn);
        const float r = q4.x, x = q4.y, y = q4.z, z = q4.w;
        const float s0 = a.scale_mod * act_scale(sc_raw[0], a.raw), s1 = a.scale_mod * act_scale(sc_raw[1], a.raw),
                    s2 = a.scale_mod * act_scale(sc_raw[2], a.raw);
        float L[9];
        L[0] = (1 - 2 * (y * y + z * z)) * s0; L[1] = (2 * (x * y - r * z)) * s1;     L[2] = (2 * (x * z + r * y)) * s2;
        L[3] = (2 * (x * y + r * z)) * s0;     L[4] = (1 - 2 * (x * x + z * z)) * s1; L[5] = (2 * (y * z - r * x)) * s2;
        L[6] = (2 * (x * z - r * y)) * s0;     L[7] = (2 * (y * z + r * x)) * s1;     L[8] = (1 - 2 * (x * x + y * y)) * s2;
        c6[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
        c6[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
        c6[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
        c6[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
        c6[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
        c6[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
      }
      const float limx = kFovGuard * a.tanfovx, limy = kFovGuard * a.tanfovy;
      const float txc = fminf(limx, fmaxf(-limx, tx / tz)) * tz;
      const float tyc = fminf(limy, fmaxf(-limy, ty / tz)) * tz;
      const float fx = (float)a.W / (2 * a.tanfovx), fy = (float)a.H / (2 * a.tanfovy);
      const float J00 = fx / tz, J02 = -(fx * txc) / (tz * tz), J11 = fy / tz, J12 = -(fy * tyc) / (tz * tz);
      float M0[3], M1[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        M0[c] = J00 * vm[c * 4 + 0] + J02 * vm[c * 4 + 2];
        M1[c] = J11 * vm[c * 4 + 1] + J12 * vm[c * 4 + 2];
      }
      float s0v[3], s1v[3];
      s0v[0] = c6[0] * M0[0] + c6[1] * M0[1] + c6[2] * M0[2];
      s0v[1] = c6[1] * M0[0] + c6[3] * M0[1] + c6[4] * M0[2];
      s0v[2] = c6[2] * M0[0] + c6[4] * M0[1] + c6[5] * M0[2];
      s1v[0] = c6[0] * M1[0] + c6[1] * M1[1] + c6[2] * M1[2];
      s1v[1] = c6[1] * M1[0] + c6[3] * M1[1] + c6[4] * M1[2];
      s1v[2] = c6[2] * M1[0] + c6[4] * M1[1] + c6[5] * M1[2];
      const float ca = M0[0] * s0v[0] + M0[1] * s0v[1] + M0[2] * s0v[2] + kDilation;
      const float cb = M0[0] * s1v[0] + M0[1] * s1v[1] + M0[2] * s1v[2];
      const float cc = M1[0] * s1v[0] + M1[1] * s1v[1] + M1[2] * s1v[2] + kDilation;
      const float det = ca * cc - cb * cb;
      if (det != 0.0f) {
        const float det_inv = 1.0f / det;
        const float mid = 0.5f * (ca + cc);
        const float disc = sqrtf(fmaxf(kLambdaFloor, mid * mid - det));
        const float lam = fmaxf(mid + disc, mid - disc);
        radius = (int)ceilf(kRadiusSigma * sqrtf(lam));
        sx = ((ndcx + 1) * a.W - 1) * 0.5f; sy = ((ndcy + 1) * a.H - 1) * 0.5f;
        x0 = min(a.tiles_x, max(0, (int)((sx - radius) / kTile)));
        y0 = min(a.tiles_y, max(0, (int)((sy - radius) / kTile)));
        x1 = min(a.tiles_x, max(0, (int)((sx + radius + kTile - 1) / kTile)));
        y1 = min(a.tiles_y, max(0, (int)((sy + radius + kTile - 1) / kTile)));
        if ((x1 - x0) * (y1 - y0) > 0) {
          vis = true;
          cA = cc * det_inv; cB = -cb * det_inv; cC = ca * det_inv;
          op = act_opacity(op_raw, a.raw);
          // alpha >= 1/255 needs q <= 2 ln(255 o): tiles (here) and 8x8 pixel blocks (blend kernels) whose minimum q
          // is larger are skipped without changing a single output bit (common.hpp: conic_min_over_box).
          kcut = 1.01f * 2.0f * logf(fmaxf(255.0f * op, 1.0f)) + 0.02f;
          // det < 0 (an indefinite conic: fp32 cancellation on a huge or near-camera splat, or a cov3D_precomp that is not
          // positive semi-definite) passes the reference's `det == 0` cull and is then drawn wherever power <= 0.  The exact
          // culling below assumes a convex q, so such a splat keeps its whole rectangle, and kcut = +inf tells the blend
          // kernels both to skip their block test and to run the group with the `power > 0` guard (gs3d_forward.hip).
          if (!(det > 0.0f)) { kcut = __builtin_inff(); mask = kMaskAll; }
          else if (x1 - x0 <= 8 && y1 - y0 <= 8) {
            for (int ty = y0; ty < y1; ty++)
              for (int tx = x0; tx < x1; tx++)
                if (conic_min_over_box(sx, sy, cA, cB, cC, (float)(tx * kTile), (float)(tx * kTile + kTile - 1),
                                       (float)(ty * kTile), (float)(ty * kTile + kTile - 1)) <= kcut)
                  mask |= 1ull << ((ty - y0) * 8 + (tx - x0));
          } else {
            mask = kMaskAll;
          }
        }
      }
    }
  }
  SplatGeom r;
  r.vis = vis; r.tz = tz; r.sx = sx; r.sy = sy; r.cA = cA; r.cB = cB; r.cC = cC; r.op = op; r.kcut = kcut;
  r.mask = mask; r.x0 = x0; r.y0 = y0; r.x1 = x1; r.y1 = y1; r.radius = radius;
  return r;
}

// Writes one Gaussian's records: SplatRec (visible only), BinRec, tile mask, radius.
__device__ __forceinline__ void emit_splat(const PgArgs &a, int i, const SplatGeom &g, const float *rgb, int clamp_bits,
                                           SplatRec *__restrict__ rec, BinRec *__restrict__ bin,
                                           uint64_t *__restrict__ tile_mask, int32_t *__restrict__ radii,
                                           uint32_t *__restrict__ tile_count) {
  BinRec br;
  br.x0 = br.y0 = br.x1 = br.y1 = 0; br.depth_bits = 0; br.radius = 0;
  int radius_out = 0;
  if (g.vis) {
    float4 *dst = reinterpret_cast<float4 *>(rec + i);
    // the conic in the blend kernels' exponent form (SplatRec, common.hpp)
    dst[0] = make_float4(g.sx, g.sy, -kConicScale * g.cA, -2.0f * kConicScale * g.cB);
    dst[1] = make_float4(-kConicScale * g.cC, __builtin_amdgcn_logf(g.op), rgb[0], rgb[1]);
    dst[2] = make_float4(rgb[2], g.tz, kConicScale * g.kcut, g.op);
    br.x0 = (uint16_t)g.x0; br.y0 = (uint16_t)g.y0; br.x1 = (uint16_t)g.x1; br.y1 = (uint16_t)g.y1;
    br.depth_bits = __float_as_uint(g.tz);
    br.radius = g.radius | (clamp_bits << kClampShift);
    radius_out = g.radius;
    if (a.count_with_atomics)  // fallback binning for images with more tiles than an LDS histogram holds
      for_each_tile(g.x0, g.y0, g.x1, g.y1, g.mask, a.tiles_x, [&](int t) { atomicAdd(&tile_count[t], 1u); });
  }
  reinterpret_cast<uint4 *>(bin)[i] = *reinterpret_cast<const uint4 *>(&br);
  tile_mask[i] = g.mask;
  radii[i] = radius_out;
  if (a.visible) a.visible[i] = radius_out > 0 ? 1 : 0;
}

// LIN: a full block of the split K = 16 layout, whose SH rows go to LDS with direct global -> LDS loads.  It is a
// separate instantiation so that the compiler sees ONE straight path "parameter loads, 12 LDS loads per wave, maths":
// its wait for the parameters then is vmcnt(12) and the 48 KiB stream overlaps the projection maths (with the runtime
// flag the paths merged and the wait was vmcnt(0): the maths started only after the whole stream had landed).
template <int DEG, bool SPLIT, bool LIN>
__device__ __forceinline__ void preprocess_body(const PgArgs &a, float *s_sh, SplatRec *__restrict__ rec,
                                                BinRec *__restrict__ bin, uint64_t *__restrict__ tile_mask,
                                                int32_t *__restrict__ radii, uint32_t *__restrict__ tile_count) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = LIN || i < a.N;
  constexpr int NFL = 3 * (DEG + 1) * (DEG + 1);
  const size_t i0 = (size_t)blockIdx.x * 256;
  const int nrows_f = LIN ? 256 : min(256, a.N - (int)i0);
  constexpr bool lin = LIN;
  // The per-Gaussian parameters leave FIRST: vmcnt retires in order, so loads issued behind the 48 KiB SH stream would
  // only be usable once all of it has landed; issued ahead of it, the projection maths overlaps the stream.
  float px_ = 0, py_ = 0, pz_ = 0, op_raw = 0;
  float4 q_raw = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
  float sc_raw[3] = {0.0f, 0.0f, 0.0f};
  if constexpr (LIN) {   // (scales / rotations present: the caller checked)
    RawParams r;
    float p[11];
    raw_issue_params(r, a.means3D + 3 * (size_t)i, a.rotations + 4 * (size_t)i, a.scales + 3 * (size_t)i, a.opacities + i);
    // One view per launch: nobody reads these rows again before the backward, > 1 GB of traffic later - nontemporal (same box,
    // S3: 63 - 66 -> 57 us).  A stacked launch (a.views > 1) re-reads them once per view, from L2: default policy there.
    if (SCORP_NT_SH && a.views <= 1) stage_sh_linear_async<2>(s_sh, a.shs, a.shs_rest, i0);
    else stage_sh_linear_async<0>(s_sh, a.shs, a.shs_rest, i0);
    raw_take_params(r, p);
    px_ = p[0]; py_ = p[1]; pz_ = p[2];
    q_raw = make_float4(p[3], p[4], p[5], p[6]);
    sc_raw[0] = p[7]; sc_raw[1] = p[8]; sc_raw[2] = p[9];
    op_raw = p[10];
  } else if (active) {
    px_ = a.means3D[3 * (size_t)i]; py_ = a.means3D[3 * (size_t)i + 1]; pz_ = a.means3D[3 * (size_t)i + 2];
    if (!a.cov3D_precomp) {
      q_raw = reinterpret_cast<const float4 *>(a.rotations)[i];
      sc_raw[0] = a.scales[3 * (size_t)i]; sc_raw[1] = a.scales[3 * (size_t)i + 1]; sc_raw[2] = a.scales[3 * (size_t)i + 2];
    }
    op_raw = a.opacities[i];
  }
  const int view = a.views > 1 ? (int)blockIdx.y : 0;   // workgroup-uniform: the matrices stay scalar loads
  float vm[16], pm[16];
  {  // camera matrices through the scalar cache (constant address space): no VMEM slots, no vmcnt dependence
    const CFloat *cv = (const CFloat *)a.view + 16 * view, *cp = (const CFloat *)a.proj + 16 * view;
#pragma unroll
    for (int q = 0; q < 16; q++) { vm[q] = cv[q]; pm[q] = cp[q]; }
  }
  SplatGeom g;
  g.vis = false; g.mask = 0; g.radius = 0;
  if (active) g = splat_geometry(a, vm, pm, i, px_, py_, pz_, q_raw, sc_raw, op_raw, a.cov3D_precomp);
  if (view > 0) {   // into the view's band of the stacked image: only the TILE rectangle moves (integers; the tile mask
                    // is relative to it).  The record keeps the view's own pixel coordinates - adding view * H to sy
                    // would round its low bits away - and the blend kernel subtracts the band's first row instead.
    g.y0 += view * a.tiles_y; g.y1 += view * a.tiles_y;
  }
  const bool vis = active && g.vis;
  float rgb[3] = {0.0f, 0.0f, 0.0f};
  int clamp_bits = 0;
  if (DEG == 0 && a.shs) {
    // Degree 0 (the objects of the pose sweep and of the post-refinement, 45 stacked views per launch): the colour is three
    // floats of the Gaussian's own row - read directly.  Through the LDS rows this instantiation carried the 50 KB array of the
    // degree-3 one (three workgroups per CU for a kernel that is a chain of loads) and two workgroup barriers.
    if (vis) {
      const float *d = a.shs + (SPLIT ? (size_t)3 : (size_t)3 * a.K) * i;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        rgb[q] = SH_C0 * d[q] + 0.5f;
        if (rgb[q] < 0.0f) clamp_bits |= 1 << q;
        rgb[q] = fmaxf(rgb[q], 0.0f);
      }
    }
  } else if (a.shs) {
    if (lin) stage_sh_wait();
    if (__syncthreads_or(vis ? 1 : 0)) {  // a block with nothing visible never touches its SH rows
      if (!lin) {
        stage_sh_rows<NFL, SPLIT>(s_sh, a.shs, a.shs_rest, a.K, i0, nrows_f);
        __syncthreads();
      }
      if (vis) {
        const CFloat *cc = (const CFloat *)a.campos + 3 * view;
        const float dx = px_ - cc[0], dy = py_ - cc[1], dz = pz_ - cc[2];
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        sh_row_to_rgb<DEG>(sh_row(s_sh, threadIdx.x, lin), dx * inv, dy * inv, dz * inv, rgb);
#pragma unroll
        for (int q = 0; q < 3; q++) {
          if (rgb[q] < 0.0f) clamp_bits |= 1 << q;  // remembered for the backward (zero gradient where clamped)
          rgb[q] = fmaxf(rgb[q], 0.0f);
        }
      }
    }
  } else if (vis) {
#pragma unroll
    for (int q = 0; q < 3; q++) rgb[q] = a.colors_precomp[3 * (size_t)i + q];
  }
  if (!active) return;
  emit_splat(a, view * a.N + i, g, rgb, clamp_bits, rec, bin, tile_mask, radii, tile_count);
}

template <int DEG, bool SPLIT>
__global__ void __launch_bounds__(256)
preprocess_kernel(PgArgs a, SplatRec *__restrict__ rec, BinRec *__restrict__ bin, uint64_t *__restrict__ tile_mask,
                  int32_t *__restrict__ radii, uint32_t *__restrict__ tile_count) {
  __shared__ __attribute__((aligned(16))) float s_sh[DEG == 0 ? 4 : 256 * kShStride];   // direct global->LDS loads land 16-byte words
  if constexpr (SPLIT && DEG == 3) {
    if (a.shs != nullptr && a.K == 16 && !a.cov3D_precomp && a.N - (int)blockIdx.x * 256 >= 256) {
      preprocess_body<DEG, SPLIT, true>(a, s_sh, rec, bin, tile_mask, radii, tile_count);
      return;
    }
  }
  preprocess_body<DEG, SPLIT, false>(a, s_sh, rec, bin, tile_mask, radii, tile_count);
}

// ---------------------------------------------------------------------------------------------------------
// LIN as in the forward: full workgroups of the training layout (dc / rest split, K = 16, scales + rotations) fetch
// everything a Gaussian needs - radius word, accumulator row, parameters: 22 dwords - with untracked loads AHEAD of
// the SH stream and pick them up at vmcnt(12); otherwise the chain "radius -> visible? -> accumulators, parameters"
// would sit behind the 48 KiB stream (the compiler waits with vmcnt(0) while LDS-DMA loads are pending).
template <int DEG, bool SPLIT, bool LIN>
__device__ __forceinline__ void preprocess_backward_body(const PgArgs &a, float *s_sh, const BinRec *__restrict__ bin,
                                                         const float *__restrict__ acc, const ScorpGs3dGrads &g, const AdamEpi &ad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = LIN || i < a.N;
  const size_t i0 = (size_t)blockIdx.x * 256;
  const int nrows = LIN ? 256 : min(256, a.N - (int)i0);
  constexpr int NFL = 3 * (DEG + 1) * (DEG + 1);
  float pre[11], pre_acc[11];   // LIN: means 0-2, rotation 3-6, scale 7-9, opacity 10 | accumulators 0-9, radius word 10
  int32_t rad_bits = 0;
  if constexpr (LIN) {
    RawParams r, ra;
    raw_issue_params(r, a.means3D + 3 * (size_t)i, a.rotations + 4 * (size_t)i, a.scales + 3 * (size_t)i, a.opacities + i);
    static_assert(kAccStride >= 10, "accumulator row: 10 floats used here");
    const float *ap = acc + (size_t)i * kAccStride;
    raw_issue_params(ra, ap, ap + 3, ap + 7, reinterpret_cast<const float *>(&bin[i].radius));
    // nontemporal as in the forward (94 -> 90.5 us) - unless the optimizer step runs in the epilogue: the block then reads its SH
    // parameters a second time, from L2
    if (SCORP_NT_SH && !(SPLIT && ad.on != 0)) stage_sh_linear_async<2>(s_sh, a.shs, a.shs_rest, i0);
    else stage_sh_linear_async<0>(s_sh, a.shs, a.shs_rest, i0);
    raw_take_params(r, pre);
    raw_take_params(ra, pre_acc);   // (its vmcnt(12) is already satisfied)
    rad_bits = __float_as_int(pre_acc[10]);
  } else {
    rad_bits = active ? bin[i].radius : 0;
  }
  const bool visible = (rad_bits & kRadiusMask) != 0;
  // fused Adam (scorp_gs3d_train_view with `adam`): the SH gradient rows are wanted in LDS even when nobody asked for them
  // in global memory, and the geometry chain runs even when every gradient pointer is NULL
  const bool adam_on = SPLIT && ad.on != 0;
  const bool adam_sh = adam_on && (ad.m[1] != nullptr || ad.m[2] != nullptr);
  const bool adam_geom = adam_on && (ad.m[0] || ad.m[3] || ad.m[4] || ad.m[5] || ad.accum);
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
  float vm[16], pm[16];
#pragma unroll
  for (int q = 0; q < 16; q++) { vm[q] = ((const CFloat *)a.view)[q]; pm[q] = ((const CFloat *)a.proj)[q]; }   // scalar cache
  float gm[3] = {0, 0, 0}, gs[3] = {0, 0, 0}, gq[4] = {0, 0, 0, 0}, gc6[6] = {0, 0, 0, 0, 0, 0};
  float shx = 0, shy = 0, shz = 0, shinv = 0, gr3[3] = {0, 0, 0};
  float a_[10];
#pragma unroll
  for (int q = 0; q < 10; q++) a_[q] = 0.0f;
  float g_op = 0.0f;
  const ShRow row = sh_row(s_sh, threadIdx.x, lin);
  // colour-only calls (post_refine_gs.py:53-56: every geometry / opacity leaf frozen) skip the whole geometry chain
  const bool want_geom = g.means3D || g.means2D || g.opacities || g.scales || g.rotations || g.cov3D_precomp || adam_geom;
  if (visible && !want_geom) {
#pragma unroll
    for (int q = 6; q < 9; q++) a_[q] = LIN ? pre_acc[q] : acc[(size_t)i * kAccStride + q];
    if (a.shs) {
      const float p0 = LIN ? pre[0] : a.means3D[3 * (size_t)i], p1 = LIN ? pre[1] : a.means3D[3 * (size_t)i + 1],
                  p2 = LIN ? pre[2] : a.means3D[3 * (size_t)i + 2];
      const float d0 = p0 - ((const CFloat *)a.campos)[0], d1 = p1 - ((const CFloat *)a.campos)[1], d2_ = p2 - ((const CFloat *)a.campos)[2];
      shinv = 1.0f / sqrtf(d0 * d0 + d1 * d1 + d2_ * d2_);
      shx = d0 * shinv; shy = d1 * shinv; shz = d2_ * shinv;
#pragma unroll
      for (int ch = 0; ch < 3; ch++) gr3[ch] = ((rad_bits >> (kClampShift + ch)) & 1) ? 0.0f : a_[6 + ch];
    }
  }
  if (visible && want_geom) {
    float p0, p1, p2;
    if constexpr (LIN) {
#pragma unroll
      for (int q = 0; q < 10; q++) a_[q] = pre_acc[q];
      p0 = pre[0]; p1 = pre[1]; p2 = pre[2];
    } else {
      const float4 *ap = reinterpret_cast<const float4 *>(acc + (size_t)i * kAccStride);
      const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
      a_[0] = a0.x; a_[1] = a0.y; a_[2] = a0.z; a_[3] = a0.w; a_[4] = a1.x; a_[5] = a1.y; a_[6] = a1.z; a_[7] = a1.w;
      a_[8] = a2.x; a_[9] = a2.y;
      p0 = a.means3D[3 * (size_t)i]; p1 = a.means3D[3 * (size_t)i + 1]; p2 = a.means3D[3 * (size_t)i + 2];
    }
    const float tx = vm[0] * p0 + vm[4] * p1 + vm[8] * p2 + vm[12];
    const float ty = vm[1] * p0 + vm[5] * p1 + vm[9] * p2 + vm[13];
    const float tz = __builtin_fmaf(vm[10], p2, __builtin_fmaf(vm[6], p1, __builtin_fmaf(vm[2], p0, vm[14])));
    float c6[6];
    float R[9], sm[3] = {0, 0, 0}, sact[3] = {0, 0, 0}, inv_qn = 1.0f;
    float4 qn = make_float4(1, 0, 0, 0);
    if (!LIN && a.cov3D_precomp) {
#pragma unroll
      for (int q = 0; q < 6; q++) c6[q] = a.cov3D_precomp[6 * (size_t)i + q];
    } else {
      float4 q_in;
      float s_in[3];
      if constexpr (LIN) {
        q_in = make_float4(pre[3], pre[4], pre[5], pre[6]);
        s_in[0] = pre[7]; s_in[1] = pre[8]; s_in[2] = pre[9];
      } else {
        q_in = reinterpret_cast<const float4 *>(a.rotations)[i];
#pragma unroll
        for (int k = 0; k < 3; k++) s_in[k] = a.scales[3 * (size_t)i + k];
      }
      qn = act_quat(q_in, a.raw, &inv_qn);
      const float r = qn.x, x = qn.y, y = qn.z, z = qn.w;
      R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z);     R[2] = 2 * (x * z + r * y);
      R[3] = 2 * (x * y + r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
      R[6] = 2 * (x * z - r * y);     R[7] = 2 * (y * z + r * x);     R[8] = 1 - 2 * (x * x + y * y);
#pragma unroll
      for (int k = 0; k < 3; k++) { sact[k] = act_scale(s_in[k], a.raw); sm[k] = a.scale_mod * sact[k]; }
      float L[9];
#pragma unroll
      for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
        for (int k = 0; k < 3; k++) L[r_ * 3 + k] = R[r_ * 3 + k] * sm[k];
      c6[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
      c6[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
      c6[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
      c6[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
      c6[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
      c6[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
    }
    const float limx = kFovGuard * a.tanfovx, limy = kFovGuard * a.tanfovy;
    const float txtz = tx / tz, tytz = ty / tz;
    const bool clamp_x = (txtz < -limx) || (txtz > limx), clamp_y = (tytz < -limy) || (tytz > limy);
    const float txc = fminf(limx, fmaxf(-limx, txtz)) * tz, tyc = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float fx = (float)a.W / (2 * a.tanfovx), fy = (float)a.H / (2 * a.tanfovy);
    const float J00 = fx / tz, J02 = -(fx * txc) / (tz * tz), J11 = fy / tz, J12 = -(fy * tyc) / (tz * tz);
    float M0[3], M1[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      M0[c] = J00 * vm[c * 4 + 0] + J02 * vm[c * 4 + 2];
      M1[c] = J11 * vm[c * 4 + 1] + J12 * vm[c * 4 + 2];
    }
    float s0[3], s1[3];
    s0[0] = c6[0] * M0[0] + c6[1] * M0[1] + c6[2] * M0[2];
    s0[1] = c6[1] * M0[0] + c6[3] * M0[1] + c6[4] * M0[2];
    s0[2] = c6[2] * M0[0] + c6[4] * M0[1] + c6[5] * M0[2];
    s1[0] = c6[0] * M1[0] + c6[1] * M1[1] + c6[2] * M1[2];
    s1[1] = c6[1] * M1[0] + c6[3] * M1[1] + c6[4] * M1[2];
    s1[2] = c6[2] * M1[0] + c6[4] * M1[1] + c6[5] * M1[2];
    const float ca = M0[0] * s0[0] + M0[1] * s0[1] + M0[2] * s0[2] + kDilation;
    const float cb = M0[0] * s1[0] + M0[1] * s1[1] + M0[2] * s1[2];
    const float cc = M1[0] * s1[0] + M1[1] * s1[1] + M1[2] * s1[2] + kDilation;
    const float det = ca * cc - cb * cb, d2 = 1.0f / (det * det + kDet2Eps);
    const float gA = a_[2], gB = a_[3], gC = a_[4];
    const float ga = d2 * (-cc * cc * gA + cb * cc * gB - cb * cb * gC);
    const float gc = d2 * (-cb * cb * gA + ca * cb * gB - ca * ca * gC);
    const float gb = d2 * (2 * cb * cc * gA - (ca * cc + cb * cb) * gB + 2 * ca * cb * gC);
    const float h = 0.5f * gb;
    float F[9];
#pragma unroll
    for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
      for (int q = 0; q < 3; q++)
        F[r_ * 3 + q] = ga * M0[r_] * M0[q] + h * (M0[r_] * M1[q] + M1[r_] * M0[q]) + gc * M1[r_] * M1[q];
    gc6[0] = F[0]; gc6[1] = 2 * F[1]; gc6[2] = 2 * F[2]; gc6[3] = F[4]; gc6[4] = 2 * F[5]; gc6[5] = F[8];
    float gM0[3], gM1[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      gM0[q] = 2 * (ga * s0[q] + h * s1[q]);
      gM1[q] = 2 * (h * s0[q] + gc * s1[q]);
    }
    float gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      gJ00 += gM0[q] * vm[q * 4 + 0]; gJ02 += gM0[q] * vm[q * 4 + 2];
      gJ11 += gM1[q] * vm[q * 4 + 1]; gJ12 += gM1[q] * vm[q * 4 + 2];
    }
    const float tz2 = 1.0f / (tz * tz), tz3 = tz2 / tz;
    float gt[3];
    gt[0] = clamp_x ? 0.0f : -fx * tz2 * gJ02;
    gt[1] = clamp_y ? 0.0f : -fy * tz2 * gJ12;
    gt[2] = -fx * tz2 * gJ00 - fy * tz2 * gJ11 + 2 * fx * txc * tz3 * gJ02 + 2 * fy * tyc * tz3 * gJ12;
    gt[2] += a_[9];
#pragma unroll
    for (int q = 0; q < 3; q++) gm[q] += vm[q * 4 + 0] * gt[0] + vm[q * 4 + 1] * gt[1] + vm[q * 4 + 2] * gt[2];
    const float hx = pm[0] * p0 + pm[4] * p1 + pm[8] * p2 + pm[12];
    const float hy = pm[1] * p0 + pm[5] * p1 + pm[9] * p2 + pm[13];
    const float hw = pm[3] * p0 + pm[7] * p1 + pm[11] * p2 + pm[15];
    const float pw = 1.0f / (hw + kWEps);
#pragma unroll
    for (int q = 0; q < 3; q++)
      gm[q] += (pm[q * 4 + 0] * pw - pm[q * 4 + 3] * hx * pw * pw) * a_[0] +
               (pm[q * 4 + 1] * pw - pm[q * 4 + 3] * hy * pw * pw) * a_[1];
    g_op = a_[5];
    if (a.raw & 1) {
      const float o = act_opacity(LIN ? pre[10] : a.opacities[i], a.raw);
      g_op *= o * (1.0f - o);
    }
    if (a.shs) {   // the SH part itself runs after the rows have landed in LDS (below)
      const float d0 = p0 - ((const CFloat *)a.campos)[0], d1 = p1 - ((const CFloat *)a.campos)[1], d2_ = p2 - ((const CFloat *)a.campos)[2];
      shinv = 1.0f / sqrtf(d0 * d0 + d1 * d1 + d2_ * d2_);
      shx = d0 * shinv; shy = d1 * shinv; shz = d2_ * shinv;
#pragma unroll
      for (int ch = 0; ch < 3; ch++) gr3[ch] = ((rad_bits >> (kClampShift + ch)) & 1) ? 0.0f : a_[6 + ch];
    }
    if (!a.cov3D_precomp) {
      const float r = qn.x, x = qn.y, y = qn.z, z = qn.w;
      const float Gs[9] = {gc6[0], 0.5f * gc6[1], 0.5f * gc6[2], 0.5f * gc6[1], gc6[3], 0.5f * gc6[4],
                           0.5f * gc6[2], 0.5f * gc6[4], gc6[5]};
      float gL[9], gR[9];
#pragma unroll
      for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
          float s = 0;
#pragma unroll
          for (int m = 0; m < 3; m++) s += Gs[r_ * 3 + m] * R[m * 3 + k] * sm[k];
          gL[r_ * 3 + k] = 2 * s;
        }
#pragma unroll
      for (int k = 0; k < 3; k++) {
        gs[k] = a.scale_mod * (R[k] * gL[k] + R[3 + k] * gL[3 + k] + R[6 + k] * gL[6 + k]);
        if (a.raw & 2) gs[k] *= sact[k];  // d exp(v) / dv
#pragma unroll
        for (int r_ = 0; r_ < 3; r_++) gR[r_ * 3 + k] = gL[r_ * 3 + k] * sm[k];
      }
      gq[0] = 2 * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
      gq[1] = 2 * (y * gR[1] + z * gR[2] + y * gR[3] - 2 * x * gR[4] - r * gR[5] + z * gR[6] + r * gR[7] - 2 * x * gR[8]);
      gq[2] = 2 * (-2 * y * gR[0] + x * gR[1] + r * gR[2] + x * gR[3] + z * gR[5] - r * gR[6] + z * gR[7] - 2 * y * gR[8]);
      gq[3] = 2 * (-2 * z * gR[0] - r * gR[1] + x * gR[2] + r * gR[3] - 2 * z * gR[4] + y * gR[5] + x * gR[6] + y * gR[7]);
      if (a.raw & 4) {  // through q / |q|
        const float dotq = r * gq[0] + x * gq[1] + y * gq[2] + z * gq[3];
        gq[0] = (gq[0] - r * dotq) * inv_qn; gq[1] = (gq[1] - x * dotq) * inv_qn;
        gq[2] = (gq[2] - y * dotq) * inv_qn; gq[3] = (gq[3] - z * dotq) * inv_qn;
      }
    }
  }
  if (a.shs && lin) { stage_sh_wait(); __syncthreads(); }
  AdamGeomMoments am;   // asked for here, used in the epilogue: the SH phase in between hides the latency
  const bool adam_pre = SPLIT && adam_on && active;
  if (adam_pre) {
    adam_moments_load<3>(ad, 0, 3 * (size_t)i, am.m, am.v);
    adam_moments_load<1>(ad, 3, (size_t)i, am.m + 3, am.v + 3);
    adam_moments_load<3>(ad, 4, 3 * (size_t)i, am.m + 4, am.v + 4);
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
    if (g.means2D) { g.means2D[3 * (size_t)i] = a_[0]; g.means2D[3 * (size_t)i + 1] = a_[1]; g.means2D[3 * (size_t)i + 2] = 0.0f; }
    if (g.colors_precomp) { g.colors_precomp[3 * (size_t)i] = a_[6]; g.colors_precomp[3 * (size_t)i + 1] = a_[7]; g.colors_precomp[3 * (size_t)i + 2] = a_[8]; }
    if (g.opacities) g.opacities[i] = g_op;
    if (g.scales) { g.scales[3 * (size_t)i] = gs[0]; g.scales[3 * (size_t)i + 1] = gs[1]; g.scales[3 * (size_t)i + 2] = gs[2]; }
    if (g.rotations) reinterpret_cast<float4 *>(g.rotations)[i] = make_float4(gq[0], gq[1], gq[2], gq[3]);
    if (g.cov3D_precomp) {
#pragma unroll
      for (int q = 0; q < 6; q++) g.cov3D_precomp[6 * (size_t)i + q] = gc6[q];
    }
  }
  // The optimizer step of this Gaussian, here, where its whole gradient row is at hand (train_3dgs.py:191-193 over
  // gs3dgs/scene/gaussian_model.py:197-206, as FusedAdam / scorp_adam_step_guarded apply it - same adam_one, same bits), and
  // the view's share of the densification statistics (train_3dgs.py:180-181; scorp_densification_stats' arithmetic).
  // Saves, per Gaussian and iteration, the 248-byte gradient row's trip to HBM and back and a second read of the 236
  // bytes of parameters: 1652 -> 932 bytes of optimizer traffic.  Nothing moves if the view's overflow word is set.
  bool adam_go = false;
  if constexpr (SPLIT) {
    if (adam_on) {
      adam_go = !(ad.skip && *ad.skip != 0u);
      if (!adam_go && ad.skipped_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ad.skipped_counter, 1u);
      if (adam_go && active) {
        float pin[11];
        if constexpr (LIN) {
#pragma unroll
          for (int q = 0; q < 11; q++) pin[q] = pre[q];
        }
        adam_leaf_pre<3>(ad, 0, const_cast<float *>(a.means3D), 3 * (size_t)i, gm, LIN ? pin : nullptr, am.m, am.v);
        adam_leaf_pre<1>(ad, 3, const_cast<float *>(a.opacities), (size_t)i, &g_op, LIN ? pin + 10 : nullptr, am.m + 3, am.v + 3);
        adam_leaf_pre<3>(ad, 4, const_cast<float *>(a.scales), 3 * (size_t)i, gs, LIN ? pin + 7 : nullptr, am.m + 4, am.v + 4);
        adam_leaf_pre<4>(ad, 5, const_cast<float *>(a.rotations), 4 * (size_t)i, gq, LIN ? pin + 3 : nullptr, am.m + 7, am.v + 7);
        if (ad.accum && visible) {
#pragma clang fp contract(off)
          const float gx = a_[0], gy = a_[1];
          ad.max_radii2D[i] = fmaxf(ad.max_radii2D[i], (float)(rad_bits & kRadiusMask));
          ad.accum[i] += sqrtf(gx * gx + gy * gy + 0.0f);
          ad.denom[i] += 1.0f;
        }
      }
    }
  }
  if (want_sh_grad) {
    if (!staged && active) sh_row_zero(row);  // nothing visible in this block: rows were never staged
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
preprocess_backward_kernel(PgArgs a, const BinRec *__restrict__ bin, const float *__restrict__ acc,
                           ScorpGs3dGrads g, AdamEpi ad) {
  __shared__ __attribute__((aligned(16))) float s_sh[256 * kShStride];   // direct global->LDS loads land 16-byte words
  if constexpr (SPLIT && DEG == 3) {
    if (a.shs != nullptr && a.K == 16 && !a.cov3D_precomp && a.N - (int)blockIdx.x * 256 >= 256) {
      preprocess_backward_body<DEG, SPLIT, true>(a, s_sh, bin, acc, g, ad);
      return;
    }
  }
  preprocess_backward_body<DEG, SPLIT, false>(a, s_sh, bin, acc, g, ad);
}

PgArgs make_args(const ScorpGs3dInputs *in, const StateLayout &L) {
  PgArgs a;
  a.N = in->num_gaussians; a.K = in->sh_coeffs; a.W = in->image_width; a.H = in->image_height;
  a.tiles_x = L.tiles_x; a.tiles_y = L.tiles_y; a.raw = in->raw_params; a.count_with_atomics = L.lds_binning ? 0 : 1;
  a.tanfovx = in->tanfovx; a.tanfovy = in->tanfovy; a.scale_mod = in->scale_modifier;
  a.view = in->viewmatrix; a.proj = in->projmatrix; a.campos = in->campos;
  a.means3D = in->means3D; a.shs = in->shs; a.shs_rest = in->shs_rest; a.colors_precomp = in->colors_precomp;
  a.opacities = in->opacities; a.scales = in->scales; a.rotations = in->rotations; a.cov3D_precomp = in->cov3D_precomp;
  a.visible = nullptr;
  a.views = in->num_views > 1 ? in->num_views : 1;
  if (a.views > 1) a.tiles_y = L.tiles_y / a.views;   // (L describes the stacked image)
  return a;
}

}  // namespace

void launch_preprocess(const ScorpGs3dInputs *in, const StateLayout &L, SplatRec *rec, BinRec *bin, uint64_t *tile_mask,
                       int32_t *radii, uint32_t *tile_count, uint8_t *out_visible, hipStream_t stream) {
  PgArgs a = make_args(in, L);
  a.visible = out_visible;
  const dim3 grid((a.N + 255) / 256, a.views), block(256);
  const int deg = in->shs ? in->sh_degree : 0;
  const bool split = in->shs_rest != nullptr;
#define SCORP_LAUNCH_PRE(D, S) preprocess_kernel<D, S><<<grid, block, 0, stream>>>(a, rec, bin, tile_mask, radii, tile_count)
  if (split) {
    switch (deg) { case 0: SCORP_LAUNCH_PRE(0, true); break; case 1: SCORP_LAUNCH_PRE(1, true); break;
                   case 2: SCORP_LAUNCH_PRE(2, true); break; default: SCORP_LAUNCH_PRE(3, true); }
  } else {
    switch (deg) { case 0: SCORP_LAUNCH_PRE(0, false); break; case 1: SCORP_LAUNCH_PRE(1, false); break;
                   case 2: SCORP_LAUNCH_PRE(2, false); break; default: SCORP_LAUNCH_PRE(3, false); }
  }
#undef SCORP_LAUNCH_PRE
}

void launch_preprocess_backward(const ScorpGs3dInputs *in, const StateLayout &L, const BinRec *bin, const float *acc,
                                const ScorpGs3dGrads *grads, hipStream_t stream, const AdamEpi *adam) {
  const PgArgs a = make_args(in, L);
  const dim3 grid((a.N + 255) / 256), block(256);
  const int deg = in->shs ? in->sh_degree : 0;
  const bool split = in->shs_rest != nullptr;
  const ScorpGs3dGrads g = *grads;
  AdamEpi ad;
  memset(&ad, 0, sizeof(ad));
  if (adam && split) ad = *adam;   // (the fused step is defined for the training layout: dc / rest split leaves)
#define SCORP_LAUNCH_PB(D, S) preprocess_backward_kernel<D, S><<<grid, block, 0, stream>>>(a, bin, acc, g, ad)
  if (split) {
    switch (deg) { case 0: SCORP_LAUNCH_PB(0, true); break; case 1: SCORP_LAUNCH_PB(1, true); break;
                   case 2: SCORP_LAUNCH_PB(2, true); break; default: SCORP_LAUNCH_PB(3, true); }
  } else {
    switch (deg) { case 0: SCORP_LAUNCH_PB(0, false); break; case 1: SCORP_LAUNCH_PB(1, false); break;
                   case 2: SCORP_LAUNCH_PB(2, false); break; default: SCORP_LAUNCH_PB(3, false); }
  }
#undef SCORP_LAUNCH_PB
}

}  // namespace scorp
