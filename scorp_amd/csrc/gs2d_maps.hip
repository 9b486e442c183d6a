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

// Gradient of the loss with respect to the two difference vectors of the normal centred at pixel c (must be interior).
__device__ __forceinline__ void centre_grads(const MapsArgs &a, const float *__restrict__ sd, const float *__restrict__ rays_d,
                                             const float *__restrict__ allmap, const float *__restrict__ g_sn, size_t HW,
                                             size_t c, V3 &g_dv, V3 &g_dh) {
  const size_t pu = c - a.W, pd = c + a.W, pl = c - 1, pr = c + 1;
  const V3 dv = point_of(sd[pd], rays_d, pd, a.ro) - point_of(sd[pu], rays_d, pu, a.ro);
  const V3 dh = point_of(sd[pr], rays_d, pr, a.ro) - point_of(sd[pl], rays_d, pl, a.ro);
  const V3 cr = cross3(dv, dh);
  const float len = sqrtf(dot3(cr, cr));
  const float al = allmap[HW + c];
  const V3 G = {g_sn[c] * al, g_sn[HW + c] * al, g_sn[2 * HW + c] * al};
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

__global__ void __launch_bounds__(256)
maps_backward_kernel(MapsDev dev, const float *__restrict__ allmap, const float *__restrict__ rays_d,
                     const float *__restrict__ sd, const float *__restrict__ g_alpha, const float *__restrict__ g_rn,
                     const float *__restrict__ g_dist, const float *__restrict__ g_sd, const float *__restrict__ g_sn,
                     float *__restrict__ g_allmap) {
  const MapsArgs a(dev);
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= a.W || y >= a.H) return;
  const size_t HW = (size_t)a.W * a.H, p = (size_t)y * a.W + x;
  float gd = g_sd ? g_sd[p] : 0.0f;
  if (g_sn) {
    V3 gp = {0.0f, 0.0f, 0.0f}, u, v;
    const bool xin = x >= 1 && x < a.W - 1, yin = y >= 1 && y < a.H - 1;
    if (xin && y >= 2) { centre_grads(a, sd, rays_d, allmap, g_sn, HW, p - a.W, u, v); gp.x += u.x; gp.y += u.y; gp.z += u.z; }       // this pixel is the lower end of dv there
    if (xin && y < a.H - 2) { centre_grads(a, sd, rays_d, allmap, g_sn, HW, p + a.W, u, v); gp.x -= u.x; gp.y -= u.y; gp.z -= u.z; }
    if (yin && x >= 2) { centre_grads(a, sd, rays_d, allmap, g_sn, HW, p - 1, u, v); gp.x += v.x; gp.y += v.y; gp.z += v.z; }
    if (yin && x < a.W - 2) { centre_grads(a, sd, rays_d, allmap, g_sn, HW, p + 1, u, v); gp.x -= v.x; gp.y -= v.y; gp.z -= v.z; }
    gd += gp.x * rays_d[3 * p] + gp.y * rays_d[3 * p + 1] + gp.z * rays_d[3 * p + 2];
  }
  const float a0 = allmap[p], al = allmap[HW + p], med = allmap[5 * HW + p];
  float g0 = 0.0f, g1 = g_alpha ? g_alpha[p] : 0.0f;
  const float ge = gd * (1.0f - a.depth_ratio);
  // where the forward's a0/alpha was nan/inf (empty pixels) nan_to_num stops the gradient; PyTorch then still divides
  // 0 by alpha = 0 and hands the rasterizer NaN at pixels it never reads — here those entries are plain zeros
  if (al != 0.0f && passes_grad(a0 / al)) { g0 = ge / al; g1 -= ge * a0 / (al * al); }
  g_allmap[p] = g0;
  g_allmap[HW + p] = g1;
#pragma unroll
  for (int i = 0; i < 3; i++)
    g_allmap[(2 + i) * HW + p] = g_rn ? g_rn[p] * a.V[i] + g_rn[HW + p] * a.V[3 + i] + g_rn[2 * HW + p] * a.V[6 + i] : 0.0f;
  g_allmap[5 * HW + p] = passes_grad(med) ? gd * a.depth_ratio : 0.0f;
  g_allmap[6 * HW + p] = g_dist ? g_dist[p] : 0.0f;
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
    maps_backward_kernel<<<dim3((W + 63) / 64, (H + 3) / 4), 256, 0, stream>>>(
        a, allmap, rays_d, surf_depth, g_render_alpha, g_render_normal, g_render_dist, g_surf_depth, g_surf_normal, g_allmap);
  }
  SCORP_KERNEL_CHECK("surfel_maps_backward", 0, stream);
  return SCORP_OK;
}
