// gs2d_maps.hip — the per-pixel tail of the 2DGS render() fused into one forward and one backward kernel.
//
// gs2dgs/gaussian_renderer/__init__.py:131-160 turns the rasterizer's allmap[7,H,W] into render_alpha, the world-space
// render_normal, the expected / median surface depth (nan_to_num'd, mixed by pipe.depth_ratio), and the pseudo surface
// normal of that depth map (gs2dgs/utils/point_utils.py:9-40: back-project with the camera rays, central differences,
// cross product, normalise, zero border, times alpha.detach()).  In PyTorch that is ~25 elementwise / indexing kernels
// forward and as many backward over 1.9 M pixels (1.2 ms per training step at 1600x1200); here it is 2 launches that
// read allmap once.  The backward is a gather: a pixel's depth enters the normals of its four 4-neighbours, so each
// thread re-derives those four centres' normal gradients instead of scattering with atomics.
#include "common.hpp"

namespace scorp {
namespace {

constexpr float kNormEps = 1e-12f;  // torch.nn.functional.normalize default eps

struct MapsDev {
  int W, H;
  float depth_ratio;
  const float *view;    // world_view_transform, 4x4 row-major as torch stores it (device)
  const float *rays_o;  // [3] (device)
};
struct MapsArgs {
  int W, H;
  float depth_ratio;
  float V[9];   // world_view_transform[:3,:3]
  float ro[3];
  __device__ explicit MapsArgs(const MapsDev &d) : W(d.W), H(d.H), depth_ratio(d.depth_ratio) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
#pragma unroll
      for (int i = 0; i < 3; i++) V[j * 3 + i] = d.view[j * 4 + i];
      ro[j] = d.rays_o[j];
    }
  }
};

// torch.nan_to_num(x, 0, 0): nan -> 0, +inf -> 0, -inf -> lowest finite
__device__ __forceinline__ float nan_to_num00(float x) {
  if (x != x) return 0.0f;
  if (x == __builtin_inff()) return 0.0f;
  if (x == -__builtin_inff()) return -3.402823466e+38f;
  return x;
}
__device__ __forceinline__ bool passes_grad(float x) { return x == x && fabsf(x) != __builtin_inff(); }

__device__ __forceinline__ float surf_depth_of(const float *__restrict__ allmap, size_t HW, size_t p, float r) {
  const float e = nan_to_num00(allmap[p] / allmap[HW + p]);
  const float m = nan_to_num00(allmap[5 * HW + p]);
  return e * (1.0f - r) + r * m;
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ V3 point_of(float d, const float *__restrict__ rays_d, size_t p, const float *ro) {
#pragma clang fp contract(off)
  return {d * rays_d[3 * p] + ro[0], d * rays_d[3 * p + 1] + ro[1], d * rays_d[3 * p + 2] + ro[2]};
}

__global__ void __launch_bounds__(256)
maps_forward_kernel(MapsDev dev, const float *__restrict__ allmap, const float *__restrict__ rays_d,
                    float *__restrict__ render_alpha, float *__restrict__ render_normal, float *__restrict__ render_dist,
                    float *__restrict__ surf_depth, float *__restrict__ surf_normal) {
  const MapsArgs a(dev);
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= a.W || y >= a.H) return;
  const size_t HW = (size_t)a.W * a.H, p = (size_t)y * a.W + x;
  const float al = allmap[HW + p];
  const float n0 = allmap[2 * HW + p], n1 = allmap[3 * HW + p], n2 = allmap[4 * HW + p];
  render_alpha[p] = al;
  render_dist[p] = allmap[6 * HW + p];
#pragma unroll
  for (int j = 0; j < 3; j++) render_normal[j * HW + p] = n0 * a.V[j * 3] + n1 * a.V[j * 3 + 1] + n2 * a.V[j * 3 + 2];
  surf_depth[p] = surf_depth_of(allmap, HW, p, a.depth_ratio);
  V3 n = {0.0f, 0.0f, 0.0f};
  if (x >= 1 && y >= 1 && x < a.W - 1 && y < a.H - 1) {
    const size_t pu = p - a.W, pd = p + a.W, pl = p - 1, pr = p + 1;
    const V3 dv = point_of(surf_depth_of(allmap, HW, pd, a.depth_ratio), rays_d, pd, a.ro) -
                  point_of(surf_depth_of(allmap, HW, pu, a.depth_ratio), rays_d, pu, a.ro);
    const V3 dh = point_of(surf_depth_of(allmap, HW, pr, a.depth_ratio), rays_d, pr, a.ro) -
                  point_of(surf_depth_of(allmap, HW, pl, a.depth_ratio), rays_d, pl, a.ro);
    const V3 c = cross3(dv, dh);
    const float inv = 1.0f / fmaxf(sqrtf(dot3(c, c)), kNormEps);
    n = {c.x * inv * al, c.y * inv * al, c.z * inv * al};
  }
  surf_normal[p] = n.x; surf_normal[HW + p] = n.y; surf_normal[2 * HW + p] = n.z;
}

// Where the backward takes its upstream gradients from.  kReg = false: the five gradient maps of an arbitrary loss
// (scorp_gs2d_maps_backward).  kReg = true: the two regularisers of train_2dgs.py:142-150 fused in —
// normal_loss = lambda_n * mean(1 - rend_normal . surf_normal), dist_loss = lambda_d * mean(render_dist) — whose
// gradient maps are scaled copies of the OTHER map (d/d rend_normal = -k surf_normal, d/d surf_normal = -k rend_normal,
// d/d dist = k_d) and are re-derived from allmap on the fly; the surface depth is recomputed instead of stored.
struct MapsGrads {
  const float *sd, *g_alpha, *g_rn, *g_dist, *g_sd, *g_sn;   // kReg = false
  float kn, kd;                                               // kReg = true: lambda_n * g0 / HW, lambda_d * g1 / HW
};

template <bool kReg>
__device__ __forceinline__ float depth_at(const MapsArgs &a, const MapsGrads &G, const float *__restrict__ allmap, size_t HW, size_t q) {
  return kReg ? surf_depth_of(allmap, HW, q, a.depth_ratio) : G.sd[q];
}

// world-space rendered normal at pixel q (what render() returns as render_normal)
__device__ __forceinline__ V3 world_normal(const MapsArgs &a, const float *__restrict__ allmap, size_t HW, size_t q) {
  const float n0 = allmap[2 * HW + q], n1 = allmap[3 * HW + q], n2 = allmap[4 * HW + q];
  return {n0 * a.V[0] + n1 * a.V[1] + n2 * a.V[2], n0 * a.V[3] + n1 * a.V[4] + n2 * a.V[5], n0 * a.V[6] + n1 * a.V[7] + n2 * a.V[8]};
}

// the two difference vectors of the normal centred at pixel c (must be interior), their cross product and its length
template <bool kReg>
__device__ __forceinline__ void centre_frame(const MapsArgs &a, const MapsGrads &G, const float *__restrict__ rays_d,
                                             const float *__restrict__ allmap, size_t HW, size_t c, V3 &dv, V3 &dh, V3 &cr,
                                             float &len) {
  const size_t pu = c - a.W, pd = c + a.W, pl = c - 1, pr = c + 1;
  dv = point_of(depth_at<kReg>(a, G, allmap, HW, pd), rays_d, pd, a.ro) - point_of(depth_at<kReg>(a, G, allmap, HW, pu), rays_d, pu, a.ro);
  dh = point_of(depth_at<kReg>(a, G, allmap, HW, pr), rays_d, pr, a.ro) - point_of(depth_at<kReg>(a, G, allmap, HW, pl), rays_d, pl, a.ro);
  cr = cross3(dv, dh);
  len = sqrtf(dot3(cr, cr));
}

// Gradient of the loss with respect to the two difference vectors dv, dh of a normal, given the gradient G with respect to the
// (alpha-weighted) unit normal: through normalize(dv x dh) and the cross product.
__device__ __forceinline__ void frame_grads(const V3 dv, const V3 dh, const V3 cr, float len, const V3 G, V3 &g_dv, V3 &g_dh) {
  V3 gc;
  if (len > kNormEps) {
    const float inv = 1.0f / len;
    const V3 n = {cr.x * inv, cr.y * inv, cr.z * inv};
    const float d = dot3(n, G);
    gc = {(G.x - n.x * d) * inv, (G.y - n.y * d) * inv, (G.z - n.z * d) * inv};
  } else {
    gc = {G.x / kNormEps, G.y / kNormEps, G.z / kNormEps};
  }
  g_dv = cross3(dh, gc);   // d(dv x dh)/d dv
  g_dh = cross3(gc, dv);   // d(dv x dh)/d dh
}
// The maps' backward, with the shared work shared (until round 6 one thread per pixel did all of it: 56 us at 1600x1200, now
// 40): a pixel's gradient needs the frames of its four neighbours, a frame needs the surface points of ITS four neighbours - one thread per pixel
// evaluated twenty surface depths (a division and two nan_to_num each) and five frames, all but one of them also
// evaluated by its neighbours.  Here a workgroup's 64 x 4 pixels stage, through LDS, the surface points of the tile
// grown by two pixels (68 x 8, once each) and the frame gradients of the tile grown by one (66 x 6, once each):
// 2.1 depths and 1.5 frames per pixel.  Same expressions in the same order per value.
constexpr int kRT_W = 64, kRT_H = 4, kRT_PW = kRT_W + 4, kRT_PH = kRT_H + 4, kRT_CW = kRT_W + 2, kRT_CH = kRT_H + 2;
// (tiles of 8 rows: the same 40 us; of 16: 50)
template <bool kReg>
__global__ void __launch_bounds__(256)
maps_backward_tiled_kernel(MapsDev dev, const float *__restrict__ allmap, const float *__restrict__ rays_d, MapsGrads Gr,
                           const float *__restrict__ g_out2, float lambda_normal, float lambda_dist, float *__restrict__ g_allmap) {
  const MapsArgs a(dev);
  __shared__ float sP[3][kRT_PW * kRT_PH];        // surface points
  __shared__ float sG[kReg ? 9 : 6][kRT_CW * kRT_CH];   // per centre: g_dv (3), g_dh (3); kReg: -k alpha normalize(dv x dh) (3)
  const int x0 = blockIdx.x * kRT_W, y0 = blockIdx.y * kRT_H;
  const size_t HW = (size_t)a.W * a.H;
  if (kReg) {
    const float inv_hw = 1.0f / (float)HW;
    Gr.kn = lambda_normal * (g_out2 ? g_out2[0] : 1.0f) * inv_hw;
    Gr.kd = lambda_dist * (g_out2 ? g_out2[1] : 1.0f) * inv_hw;
  }
  const bool want_frames = kReg || Gr.g_sn;
  if (want_frames) {
    for (int i = threadIdx.x; i < kRT_PW * kRT_PH; i += 256) {
      const int gx = x0 - 2 + i % kRT_PW, gy = y0 - 2 + i / kRT_PW;
      if (gx >= 0 && gy >= 0 && gx < a.W && gy < a.H) {
        const size_t q = (size_t)gy * a.W + gx;
        const V3 P = point_of(depth_at<kReg>(a, Gr, allmap, HW, q), rays_d, q, a.ro);
        sP[0][i] = P.x; sP[1][i] = P.y; sP[2][i] = P.z;
      }
    }
    __syncthreads();
    auto pt = [&](int i) { return V3{sP[0][i], sP[1][i], sP[2][i]}; };
    for (int i = threadIdx.x; i < kRT_CW * kRT_CH; i += 256) {
      const int lx = i % kRT_CW, ly = i / kRT_CW, cx = x0 - 1 + lx, cy = y0 - 1 + ly;
      V3 g_dv = {0.0f, 0.0f, 0.0f}, g_dh = g_dv, grn = g_dv;
      if (cx >= 1 && cy >= 1 && cx < a.W - 1 && cy < a.H - 1) {
        const int pc = (ly + 1) * kRT_PW + (lx + 1);   // the centre in the point tile
        const V3 dv = pt(pc + kRT_PW) - pt(pc - kRT_PW), dh = pt(pc + 1) - pt(pc - 1);
        const V3 cr = cross3(dv, dh);
        const float len = sqrtf(dot3(cr, cr));
        const size_t c = (size_t)cy * a.W + cx;
        const float al = allmap[HW + c];
        V3 G;
        if (kReg) {
          const V3 rn = world_normal(a, allmap, HW, c);
          G = {-Gr.kn * rn.x * al, -Gr.kn * rn.y * al, -Gr.kn * rn.z * al};
          const float sc = -Gr.kn * al / fmaxf(len, kNormEps);
          grn = {cr.x * sc, cr.y * sc, cr.z * sc};
        } else {
          G = {Gr.g_sn[c] * al, Gr.g_sn[HW + c] * al, Gr.g_sn[2 * HW + c] * al};
        }
        frame_grads(dv, dh, cr, len, G, g_dv, g_dh);
      }
      sG[0][i] = g_dv.x; sG[1][i] = g_dv.y; sG[2][i] = g_dv.z; sG[3][i] = g_dh.x; sG[4][i] = g_dh.y; sG[5][i] = g_dh.z;
      if constexpr (kReg) { sG[6][i] = grn.x; sG[7][i] = grn.y; sG[8][i] = grn.z; }
    }
    __syncthreads();
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, x = x0 + tx, y = y0 + ty;
  if (x >= a.W || y >= a.H) return;
  const size_t p = (size_t)y * a.W + x;
  const bool xin = x >= 1 && x < a.W - 1, yin = y >= 1 && y < a.H - 1;
  const int ci = (ty + 1) * kRT_CW + (tx + 1);   // this pixel in the centre tile
  float gd = (!kReg && Gr.g_sd) ? Gr.g_sd[p] : 0.0f;
  if (want_frames) {
    V3 gp = {0.0f, 0.0f, 0.0f};
    if (xin && y >= 2) { const int j = ci - kRT_CW; gp.x += sG[0][j]; gp.y += sG[1][j]; gp.z += sG[2][j]; }         // the lower end of dv there
    if (xin && y < a.H - 2) { const int j = ci + kRT_CW; gp.x -= sG[0][j]; gp.y -= sG[1][j]; gp.z -= sG[2][j]; }
    if (yin && x >= 2) { const int j = ci - 1; gp.x += sG[3][j]; gp.y += sG[4][j]; gp.z += sG[5][j]; }
    if (yin && x < a.W - 2) { const int j = ci + 1; gp.x -= sG[3][j]; gp.y -= sG[4][j]; gp.z -= sG[5][j]; }
    gd += gp.x * rays_d[3 * p] + gp.y * rays_d[3 * p + 1] + gp.z * rays_d[3 * p + 2];
  }
  const float a0 = allmap[p], al = allmap[HW + p], med = allmap[5 * HW + p];
  float g0 = 0.0f, g1 = (!kReg && Gr.g_alpha) ? Gr.g_alpha[p] : 0.0f;
  const float ge = gd * (1.0f - a.depth_ratio);
  if (al != 0.0f && passes_grad(a0 / al)) { g0 = ge / al; g1 -= ge * a0 / (al * al); }
  g_allmap[p] = g0;
  g_allmap[HW + p] = g1;
  V3 grn = {0.0f, 0.0f, 0.0f};   // gradient w.r.t. the world-space rendered normal at this pixel
  if (kReg) {
    if (xin && yin) grn = {sG[kReg ? 6 : 0][ci], sG[kReg ? 7 : 0][ci], sG[kReg ? 8 : 0][ci]};
  } else if (Gr.g_rn) {
    grn = {Gr.g_rn[p], Gr.g_rn[HW + p], Gr.g_rn[2 * HW + p]};
  }
#pragma unroll
  for (int i = 0; i < 3; i++) g_allmap[(2 + i) * HW + p] = grn.x * a.V[i] + grn.y * a.V[3 + i] + grn.z * a.V[6 + i];
  g_allmap[5 * HW + p] = passes_grad(med) ? gd * a.depth_ratio : 0.0f;
  g_allmap[6 * HW + p] = kReg ? Gr.kd : (Gr.g_dist ? Gr.g_dist[p] : 0.0f);
}

// ---- the regularisers' forward: per-block partial sums of (1 - rend_normal . surf_normal) and render_dist ----
__global__ void __launch_bounds__(256)
reg_forward_kernel(MapsDev dev, const float *__restrict__ allmap, const float *__restrict__ rays_d, float *__restrict__ partials) {
  const MapsArgs a(dev);
  __shared__ float s_red[2][4];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  float t = 0.0f, dist = 0.0f;
  if (x < a.W && y < a.H) {
    const size_t HW = (size_t)a.W * a.H, p = (size_t)y * a.W + x;
    dist = allmap[6 * HW + p];
    float dotn = 0.0f;
    if (x >= 1 && y >= 1 && x < a.W - 1 && y < a.H - 1) {
      MapsGrads none = {};
      V3 dv, dh, cr;
      float len;
      centre_frame<true>(a, none, rays_d, allmap, HW, p, dv, dh, cr, len);
      const V3 rn = world_normal(a, allmap, HW, p);
      dotn = dot3(rn, cr) * (allmap[HW + p] / fmaxf(len, kNormEps));
    }
    t = 1.0f - dotn;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { t += __shfl_xor(t, off, 64); dist += __shfl_xor(dist, off, 64); }
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = t; s_red[1][threadIdx.x >> 6] = dist; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int b = blockIdx.y * gridDim.x + blockIdx.x;
    partials[2 * b] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    partials[2 * b + 1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
  }
}

__global__ void __launch_bounds__(256)
reg_finalize_kernel(const float *__restrict__ partials, int nblocks, double inv_hw, float lambda_normal, float lambda_dist,
                    float *__restrict__ out2) {
  __shared__ double s_a[256], s_b[256];
  double sa = 0, sb = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { sa += partials[2 * i]; sb += partials[2 * i + 1]; }
  s_a[threadIdx.x] = sa; s_b[threadIdx.x] = sb;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) { s_a[threadIdx.x] += s_a[threadIdx.x + off]; s_b[threadIdx.x] += s_b[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = (float)(lambda_normal * s_a[0] * inv_hw);
    out2[1] = (float)(lambda_dist * s_b[0] * inv_hw);
  }
}

int fill_args(MapsDev &a, int W, int H, const float *viewmatrix, const float *rays_o, float depth_ratio) {
  if (W <= 0 || H <= 0) { set_error("bad image size %dx%d", W, H); return SCORP_ERR_INVALID; }
  if (!viewmatrix || !rays_o) { set_error("viewmatrix / rays_o is NULL"); return SCORP_ERR_INVALID; }
  a.W = W; a.H = H; a.depth_ratio = depth_ratio; a.view = viewmatrix; a.rays_o = rays_o;
  return SCORP_OK;
}

}  // namespace
}  // namespace scorp

using namespace scorp;

extern "C" int scorp_gs2d_maps_forward(int32_t W, int32_t H, const float *allmap, const float *viewmatrix,
                                       const float *rays_d, const float *rays_o, float depth_ratio,
                                       float *render_alpha, float *render_normal, float *render_dist, float *surf_depth,
                                       float *surf_normal, scorp_stream_t stream_) {
  MapsDev a;
  if (int e = fill_args(a, W, H, viewmatrix, rays_o, depth_ratio)) return e;
  if (!allmap || !rays_d || !render_alpha || !render_normal || !render_dist || !surf_depth || !surf_normal) {
    set_error("NULL map pointer in scorp_gs2d_maps_forward"); return SCORP_ERR_INVALID;
  }
  hipStream_t stream = (hipStream_t)stream_;
  {
    ProfScope prof(kKMapsForward2d, stream);
    maps_forward_kernel<<<dim3((W + 63) / 64, (H + 3) / 4), 256, 0, stream>>>(a, allmap, rays_d, render_alpha, render_normal,
                                                                             render_dist, surf_depth, surf_normal);
  }
  SCORP_KERNEL_CHECK("surfel_maps_forward", 0, stream);
  return SCORP_OK;
}

extern "C" int scorp_gs2d_maps_backward(int32_t W, int32_t H, const float *allmap, const float *viewmatrix,
                                        const float *rays_d, const float *rays_o, float depth_ratio,
                                        const float *surf_depth, const float *g_render_alpha, const float *g_render_normal,
                                        const float *g_render_dist, const float *g_surf_depth, const float *g_surf_normal,
                                        float *g_allmap, scorp_stream_t stream_) {
  MapsDev a;
  if (int e = fill_args(a, W, H, viewmatrix, rays_o, depth_ratio)) return e;
  if (!allmap || !rays_d || !surf_depth || !g_allmap) { set_error("NULL map pointer in scorp_gs2d_maps_backward"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  {
    ProfScope prof(kKMapsBackward2d, stream);
    MapsGrads Gr = {};
    Gr.sd = surf_depth; Gr.g_alpha = g_render_alpha; Gr.g_rn = g_render_normal; Gr.g_dist = g_render_dist;
    Gr.g_sd = g_surf_depth; Gr.g_sn = g_surf_normal;
    maps_backward_tiled_kernel<false><<<dim3((W + kRT_W - 1) / kRT_W, (H + kRT_H - 1) / kRT_H), 256, 0, stream>>>(
        a, allmap, rays_d, Gr, nullptr, 0.0f, 0.0f, g_allmap);
  }
  SCORP_KERNEL_CHECK("surfel_maps_backward", 0, stream);
  return SCORP_OK;
}

static inline int reg_blocks(int W, int H) { return ((W + 63) / 64) * ((H + 3) / 4); }

extern "C" size_t scorp_gs2d_regularizers_workspace_bytes(int32_t W, int32_t H) {
  return align_up((size_t)reg_blocks(W > 0 ? W : 1, H > 0 ? H : 1) * 8, 256);
}

extern "C" int scorp_gs2d_regularizers_forward(int32_t W, int32_t H, const float *allmap, const float *viewmatrix,
                                               const float *rays_d, const float *rays_o, float depth_ratio,
                                               float lambda_normal, float lambda_dist, float *out2, void *workspace,
                                               size_t workspace_bytes, scorp_stream_t stream_) {
  MapsDev a;
  if (int e = fill_args(a, W, H, viewmatrix, rays_o, depth_ratio)) return e;
  if (!allmap || !rays_d || !out2 || !workspace) { set_error("NULL pointer in scorp_gs2d_regularizers_forward"); return SCORP_ERR_INVALID; }
  if (workspace_bytes < scorp_gs2d_regularizers_workspace_bytes(W, H)) { set_error("regulariser workspace too small"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  {
    ProfScope prof(kKMapsForward2d, stream);
    reg_forward_kernel<<<dim3((W + 63) / 64, (H + 3) / 4), 256, 0, stream>>>(a, allmap, rays_d, (float *)workspace);
  }
  SCORP_KERNEL_CHECK("surfel_regularizers_forward", 0, stream);
  reg_finalize_kernel<<<1, 256, 0, stream>>>((const float *)workspace, reg_blocks(W, H), 1.0 / ((double)W * H), lambda_normal,
                                             lambda_dist, out2);
  SCORP_KERNEL_CHECK("surfel_regularizers_finalize", 0, stream);
  return SCORP_OK;
}

extern "C" int scorp_gs2d_regularizers_backward(int32_t W, int32_t H, const float *allmap, const float *viewmatrix,
                                                const float *rays_d, const float *rays_o, float depth_ratio,
                                                float lambda_normal, float lambda_dist, const float *g_out2,
                                                float *g_allmap, scorp_stream_t stream_) {
  MapsDev a;
  if (int e = fill_args(a, W, H, viewmatrix, rays_o, depth_ratio)) return e;
  if (!allmap || !rays_d || !g_allmap) { set_error("NULL pointer in scorp_gs2d_regularizers_backward"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  {
    ProfScope prof(kKMapsBackward2d, stream);
    MapsGrads Gr = {};
    maps_backward_tiled_kernel<true><<<dim3((W + kRT_W - 1) / kRT_W, (H + kRT_H - 1) / kRT_H), 256, 0, stream>>>(
        a, allmap, rays_d, Gr, g_out2, lambda_normal, lambda_dist, g_allmap);
  }
  SCORP_KERNEL_CHECK("surfel_regularizers_backward", 0, stream);
  return SCORP_OK;
}
