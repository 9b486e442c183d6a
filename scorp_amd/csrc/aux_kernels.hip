// aux_kernels.hip — the two small hot-path helpers that are not rasterization:
//   scorp_knn_dist2 : replaces simple_knn._C.distCUDA2 (gs3dgs/scene/gaussian_model.py:22,177) — mean squared
//                     distance of every point to its 3 nearest neighbours, used once to initialise the scales;
//   scorp_adam_step : the per-iteration Adam update of all parameter groups in ONE launch (replaces the six-group
//                     torch.optim.Adam(eps=1e-15) step of gaussian_model.py:197-206 / train_3dgs.py:191-193);
//   scorp_gs3d_render_tail : what render() does to the rasterizer's outputs (gaussian_renderer/__init__.py:113-120:
//                     render_depth = nan_to_num(depth / alpha, 0, 0), visibility_filter = radii > 0) in one launch —
//                     three 5-microsecond torch kernels per view otherwise.
#include "common.hpp"

namespace scorp {
namespace {

// ---- exact 3-NN by tiled brute force ------------------------------------------------------------------------------
// N is 1e5..4e5 at initialisation (dataset_readers.py:321,378) and the call happens once per run, so the O(N^2)
// form is the robust choice: every thread owns a query, the block sweeps all points through LDS 1024 at a time
// (broadcast reads), 3 smallest squared distances kept in registers.  ~8 VALU per pair: 360k points = 1.3e11 pairs
// : 34.9 ms measured on an MI355X (round 3, once per run at create_from_pcd).  A point is its own nearest neighbour by index, not by distance, so duplicates count with distance 0.
constexpr int kKnnTile = 1024;
__global__ void __launch_bounds__(256)
knn_dist2_kernel(int N, const float *__restrict__ xyz, float *__restrict__ out) {
  __shared__ float s_x[kKnnTile], s_y[kKnnTile], s_z[kKnnTile];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = i < N;
  float qx = 0, qy = 0, qz = 0;
  if (active) { qx = xyz[3 * (size_t)i]; qy = xyz[3 * (size_t)i + 1]; qz = xyz[3 * (size_t)i + 2]; }
  float b0 = 3.4e38f, b1 = 3.4e38f, b2 = 3.4e38f;  // b0 <= b1 <= b2
  for (int base = 0; base < N; base += kKnnTile) {
    __syncthreads();
    for (int k = threadIdx.x; k < kKnnTile; k += 256) {
      const int j = base + k;
      if (j < N) { s_x[k] = xyz[3 * (size_t)j]; s_y[k] = xyz[3 * (size_t)j + 1]; s_z[k] = xyz[3 * (size_t)j + 2]; }
    }
    __syncthreads();
    const int cnt = min(kKnnTile, N - base);
    for (int k = 0; k < cnt; k++) {
      const float dx = s_x[k] - qx, dy = s_y[k] - qy, dz = s_z[k] - qz;
      float d = dx * dx + dy * dy + dz * dz;
      d = (base + k == i) ? 3.4e38f : d;
      // insert into the sorted triple
      const float n2 = fminf(b2, fmaxf(b1, d));
      const float n1 = fminf(b1, fmaxf(b0, d));
      b0 = fminf(b0, d); b1 = n1; b2 = n2;
    }
  }
  if (active) {
    // with fewer than 4 points some slots stay "infinite": average what exists, as a k-d tree query would
    float s = 0.0f; int c = 0;
    if (b0 < 3.0e38f) { s += b0; c++; }
    if (b1 < 3.0e38f) { s += b1; c++; }
    if (b2 < 3.0e38f) { s += b2; c++; }
    out[i] = c ? s / 3.0f : 0.0f;
  }
}

// ---- multi-tensor Adam -------------------------------------------------------------------------------------------
struct AdamPack {
  ScorpAdamTensor t[SCORP_ADAM_MAX_TENSORS];
  uint32_t first_block[SCORP_ADAM_MAX_TENSORS + 1];  // block range of each tensor
  int n;
};
constexpr int kAdamPerBlock = 256 * 4 * 4;  // 4 float4 per thread

// (adam_one: common.hpp, shared with the fused epilogue of the per-Gaussian backward)

__global__ void __launch_bounds__(256)
adam_kernel(AdamPack pk, float omb1, float beta2, float omb2, float eps, float bc1, float inv_sqrt_bc2,
            const uint32_t *__restrict__ skip_if_nonzero, uint32_t *__restrict__ skipped_counter) {
  if (skip_if_nonzero && *skip_if_nonzero != 0u) {   // the step is conditional on a device word (see the entry point)
    if (skipped_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped_counter, 1u);
    return;
  }
  int ti = 0;
#pragma unroll
  for (int k = 1; k < SCORP_ADAM_MAX_TENSORS; k++)
    if (k < pk.n && blockIdx.x >= pk.first_block[k]) ti = k;
  const ScorpAdamTensor T = pk.t[ti];
  const size_t base = (size_t)(blockIdx.x - pk.first_block[ti]) * kAdamPerBlock;
  const float step_size = T.lr / bc1;
  const bool vec = (((uintptr_t)T.param | (uintptr_t)T.grad | (uintptr_t)T.exp_avg | (uintptr_t)T.exp_avg_sq) & 15) == 0;
  // Four 16-byte quads per thread, all of their loads asked for before the first is computed on (left in a loop the
  // compiler may not move a round's loads above the previous round's stores), and NONTEMPORAL: moments and gradients are
  // touched once per step, the parameters' next reader is a whole view later (pergaussian.hpp, adam_ld4 / adam_st4).
  size_t e[4];
  bool quad[4], any[4];
  float4 p[4], m[4], v[4], g[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    e[r] = base + ((size_t)r * 256 + threadIdx.x) * 4;
    any[r] = e[r] < T.numel;
    quad[r] = vec && e[r] + 4 <= T.numel;
    if (quad[r]) {
      p[r] = adam_ld4(reinterpret_cast<const float4 *>(T.param + e[r])); m[r] = adam_ld4(reinterpret_cast<const float4 *>(T.exp_avg + e[r]));
      v[r] = adam_ld4(reinterpret_cast<const float4 *>(T.exp_avg_sq + e[r])); g[r] = adam_ld4(reinterpret_cast<const float4 *>(T.grad + e[r]));
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (quad[r]) {
      adam_one(p[r].x, g[r].x, m[r].x, v[r].x, omb1, beta2, omb2, step_size, inv_sqrt_bc2, eps);
      adam_one(p[r].y, g[r].y, m[r].y, v[r].y, omb1, beta2, omb2, step_size, inv_sqrt_bc2, eps);
      adam_one(p[r].z, g[r].z, m[r].z, v[r].z, omb1, beta2, omb2, step_size, inv_sqrt_bc2, eps);
      adam_one(p[r].w, g[r].w, m[r].w, v[r].w, omb1, beta2, omb2, step_size, inv_sqrt_bc2, eps);
      adam_st4(reinterpret_cast<float4 *>(T.param + e[r]), p[r]);
      adam_st4(reinterpret_cast<float4 *>(T.exp_avg + e[r]), m[r]);
      adam_st4(reinterpret_cast<float4 *>(T.exp_avg_sq + e[r]), v[r]);
    } else if (any[r]) {
      for (size_t k = e[r]; k < min(e[r] + 4, (size_t)T.numel); k++) {
        float p1 = T.param[k], m1 = T.exp_avg[k], v1 = T.exp_avg_sq[k];
        adam_one(p1, T.grad[k], m1, v1, omb1, beta2, omb2, step_size, inv_sqrt_bc2, eps);
        T.param[k] = p1; T.exp_avg[k] = m1; T.exp_avg_sq[k] = v1;
      }
    }
  }
}

}  // namespace
}  // namespace scorp

using namespace scorp;

// ---- densify / prune compaction: every parameter tensor and both Adam moments of a model re-indexed in ONE launch ----
namespace scorp {
namespace {
struct RowPack { ScorpRowTensor t[SCORP_ROWS_MAX_TENSORS]; int n; };
// blockIdx.y = tensor; a thread moves one float: out row j <- row (src_index[j] & 0x7fffffff), zeros for a fresh row
// (bit 31 of the index) of a tensor that asks for it (the Adam moments of cloned / split Gaussians start at zero,
// gaussian_model.py:452-470 `densification_postfix`)
__global__ void __launch_bounds__(256)
gather_rows_kernel(RowPack pk, const int32_t *__restrict__ src_index, uint64_t n_out) {
  const ScorpRowTensor T = pk.t[blockIdx.y];
  const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_out * T.row_floats) return;
  const uint64_t j = e / T.row_floats;
  const uint32_t c = (uint32_t)(e - j * T.row_floats);
  const uint32_t s = (uint32_t)src_index[j];
  const bool fresh = (s >> 31) != 0;
  T.dst[e] = (fresh && T.zero_if_fresh) ? 0.0f : T.src[(uint64_t)(s & 0x7FFFFFFFu) * T.row_floats + c];
}
}  // namespace
}  // namespace scorp

namespace scorp {
namespace {
// One pass over a GaussianModel for the align loop's rigid / scale updates (utils/gaussians.py:12-108: translate, scale,
// rotate incl. the SH coefficients): thread i owns Gaussian i,
//   xyz      <- ((xyz - c) R^T) * s + c + t
//   rotation <- q (x) normalize(rotation)            (Hamilton product, q = the quaternion of R)
//   scaling  <- scaling + log(s)                     (log-space scales; `dims` of them: 3, or 2 for surfels)
//   rest[l]  <- D_l rest[l]  for the SH bands l = 1..3 present (real Wigner-D blocks, row-major 3x3, 5x5, 7x7), per channel
// instead of ~20 elementwise / einsum torch launches over the model.  params (device, 113 floats):
//   R[9] c[3] t[3] s[3] q[4] D1[9] D2[25] D3[49] | flags[8] (floats: 1 = rotate SH).
struct TransformParams { float R[9], c[3], t[3], s[3], q[4], D1[9], D2[25], D3[49]; };
__global__ void __launch_bounds__(256)
transform_gaussians_kernel(int N, int k_rest, int dims, float *__restrict__ xyz, float *__restrict__ rot, float *__restrict__ scaling,
                           float *__restrict__ rest, const TransformParams *__restrict__ pp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const TransformParams &P = *pp;   // uniform address: scalar loads
  {
    const float x = xyz[3 * (size_t)i] - P.c[0], y = xyz[3 * (size_t)i + 1] - P.c[1], z = xyz[3 * (size_t)i + 2] - P.c[2];
    xyz[3 * (size_t)i]     = (P.R[0] * x + P.R[1] * y + P.R[2] * z) * P.s[0] + P.c[0] + P.t[0];
    xyz[3 * (size_t)i + 1] = (P.R[3] * x + P.R[4] * y + P.R[5] * z) * P.s[1] + P.c[1] + P.t[1];
    xyz[3 * (size_t)i + 2] = (P.R[6] * x + P.R[7] * y + P.R[8] * z) * P.s[2] + P.c[2] + P.t[2];
  }
  if (rot) {   // (NULL: no rotation part - translate / scale leave the quaternions untouched, as utils/gaussians.py:12-40 does)
    float4 b = reinterpret_cast<const float4 *>(rot)[i];
    const float inv = 1.0f / sqrtf(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w);
    b.x *= inv; b.y *= inv; b.z *= inv; b.w *= inv;
    const float aw = P.q[0], ax = P.q[1], ay = P.q[2], az = P.q[3];
    reinterpret_cast<float4 *>(rot)[i] = make_float4(aw * b.x - ax * b.y - ay * b.z - az * b.w, aw * b.y + ax * b.x + ay * b.w - az * b.z,
                                                      aw * b.z - ax * b.w + ay * b.x + az * b.y, aw * b.w + ax * b.z - ay * b.y + az * b.x);
  }
  if (scaling)   // (NULL: no scale part - nothing is written into a tensor the caller may share with another model)
    for (int d = 0; d < dims; d++) scaling[(size_t)dims * i + d] += logf(P.s[d]);
  if (!rest) return;
  // SH bands: rest[i][j][ch], j = 0 .. k_rest-1 (coefficient 1 + j of the full set)
  float *r = rest + (size_t)i * k_rest * 3;
  if (k_rest >= 3) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      const float v0 = r[0 * 3 + ch], v1 = r[1 * 3 + ch], v2 = r[2 * 3 + ch];
#pragma unroll
      for (int a = 0; a < 3; a++) r[a * 3 + ch] = P.D1[a * 3] * v0 + P.D1[a * 3 + 1] * v1 + P.D1[a * 3 + 2] * v2;
    }
  }
  if (k_rest >= 8) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      float v[5];
#pragma unroll
      for (int b = 0; b < 5; b++) v[b] = r[(3 + b) * 3 + ch];
#pragma unroll
      for (int a = 0; a < 5; a++) {
        float acc = 0.0f;
#pragma unroll
        for (int b = 0; b < 5; b++) acc += P.D2[a * 5 + b] * v[b];
        r[(3 + a) * 3 + ch] = acc;
      }
    }
  }
  if (k_rest >= 15) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      float v[7];
#pragma unroll
      for (int b = 0; b < 7; b++) v[b] = r[(8 + b) * 3 + ch];
#pragma unroll
      for (int a = 0; a < 7; a++) {
        float acc = 0.0f;
#pragma unroll
        for (int b = 0; b < 7; b++) acc += P.D3[a * 7 + b] * v[b];
        r[(8 + a) * 3 + ch] = acc;
      }
    }
  }
}
}  // namespace
}  // namespace scorp

extern "C" int scorp_gaussians_transform(float *xyz, float *rotation, float *scaling, float *features_rest, int32_t N,
                                         int32_t rest_coeffs, int32_t scale_dims, const float *params, scorp_stream_t stream_) {
  if (N < 0 || (N > 0 && (!xyz || !params)) || rest_coeffs < 0 || scale_dims < 1 || scale_dims > 3 ||
      (((uintptr_t)rotation | (uintptr_t)params) & 15)) {
    set_error("bad arguments to scorp_gaussians_transform"); return SCORP_ERR_INVALID;
  }
  if (N == 0) return SCORP_OK;
  hipStream_t stream = (hipStream_t)stream_;
  transform_gaussians_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, rest_coeffs, scale_dims, xyz, rotation, scaling, rest_coeffs > 0 ? features_rest : nullptr,
                                                                  reinterpret_cast<const scorp::TransformParams *>(params));
  SCORP_KERNEL_CHECK("transform_gaussians", 0, stream);
  return SCORP_OK;
}

extern "C" int scorp_gather_rows(const ScorpRowTensor *tensors, int32_t n, const int32_t *src_index, uint64_t num_out_rows,
                                 scorp_stream_t stream_) {
  if (n < 0 || n > SCORP_ROWS_MAX_TENSORS || (n > 0 && !tensors) || (num_out_rows > 0 && !src_index)) {
    set_error("bad arguments to scorp_gather_rows (n=%d)", n); return SCORP_ERR_INVALID;
  }
  if (n == 0 || num_out_rows == 0) return SCORP_OK;
  RowPack pk;
  pk.n = n;
  uint64_t most = 0;
  for (int k = 0; k < n; k++) {
    if (!tensors[k].src || !tensors[k].dst || tensors[k].row_floats == 0) { set_error("scorp_gather_rows: bad tensor %d", k); return SCORP_ERR_INVALID; }
    pk.t[k] = tensors[k];
    most = most > tensors[k].row_floats ? most : tensors[k].row_floats;
  }
  const uint64_t blocks = (num_out_rows * most + 255) / 256;
  if (blocks > 0x7FFFFFFFull) { set_error("scorp_gather_rows: too many rows"); return SCORP_ERR_INVALID; }
  gather_rows_kernel<<<dim3((unsigned)blocks, (unsigned)n), 256, 0, (hipStream_t)stream_>>>(pk, src_index, num_out_rows);
  SCORP_KERNEL_CHECK("gather_rows", 0, (hipStream_t)stream_);
  return SCORP_OK;
}

extern "C" int scorp_knn_dist2(const float *xyz, int32_t N, float *out, scorp_stream_t stream_) {
  if (N < 0 || (N > 0 && (!xyz || !out))) { set_error("bad arguments to scorp_knn_dist2"); return SCORP_ERR_INVALID; }
  if (N == 0) return SCORP_OK;
  hipStream_t stream = (hipStream_t)stream_;
  {
    ProfScope prof(kKKnn, stream);
    knn_dist2_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, xyz, out);
  }
  SCORP_KERNEL_CHECK("knn_dist2", 0, stream);
  return SCORP_OK;
}

namespace scorp {
namespace {
// train_3dgs.py:180-181 for one view: the three masked updates as one pass over the Gaussians
__global__ void __launch_bounds__(256)
densification_stats_kernel(int N, const int32_t *__restrict__ radii, const uint8_t *__restrict__ visible,
                           const float *__restrict__ grad, int stride, int norm3, const uint32_t *__restrict__ skip_if_nonzero,
                           float *__restrict__ max_radii2D, float *__restrict__ accum, float *__restrict__ denom) {
#pragma clang fp contract(off)
  if (skip_if_nonzero && *skip_if_nonzero) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N || !visible[i]) return;
  const float gx = grad[(size_t)i * stride], gy = grad[(size_t)i * stride + 1];
  // the 2DGS model takes the norm over the whole row (gs2dgs/scene/gaussian_model.py:494-495), the 3DGS one over x, y
  const float gz = norm3 ? grad[(size_t)i * stride + 2] : 0.0f;
  max_radii2D[i] = fmaxf(max_radii2D[i], (float)radii[i]);
  accum[i] += sqrtf(gx * gx + gy * gy + gz * gz);
  denom[i] += 1.0f;
}
}  // namespace
}  // namespace scorp

extern "C" int scorp_densification_stats(int32_t N, const int32_t *radii, const uint8_t *visible, const float *grad_means2D,
                                         int32_t grad_stride, const uint32_t *skip_if_nonzero, float *max_radii2D,
                                         float *xyz_gradient_accum, float *denom, scorp_stream_t stream_) {
  return scorp_densification_stats_ex(N, radii, visible, grad_means2D, grad_stride, 2, skip_if_nonzero, max_radii2D,
                                      xyz_gradient_accum, denom, stream_);
}

extern "C" int scorp_densification_stats_ex(int32_t N, const int32_t *radii, const uint8_t *visible, const float *grad_means2D,
                                            int32_t grad_stride, int32_t norm_components, const uint32_t *skip_if_nonzero,
                                            float *max_radii2D, float *xyz_gradient_accum, float *denom, scorp_stream_t stream_) {
  if (N < 0 || (norm_components != 2 && norm_components != 3) || grad_stride < norm_components) {
    set_error("scorp_densification_stats: N=%d, grad_stride=%d, norm_components=%d", N, grad_stride, norm_components);
    return SCORP_ERR_INVALID;
  }
  if (N == 0) return SCORP_OK;
  if (!radii || !visible || !grad_means2D || !max_radii2D || !xyz_gradient_accum || !denom) {
    set_error("scorp_densification_stats: NULL pointer"); return SCORP_ERR_INVALID;
  }
  hipStream_t stream = (hipStream_t)stream_;
  densification_stats_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, radii, visible, grad_means2D, grad_stride, norm_components == 3,
                                                                 skip_if_nonzero, max_radii2D, xyz_gradient_accum, denom);
  SCORP_KERNEL_CHECK("densification_stats", 0, stream);
  return SCORP_OK;
}

extern "C" int scorp_adam_step(const ScorpAdamTensor *tensors, int32_t n, double beta1, double beta2, double eps,
                               int32_t step, scorp_stream_t stream_) {
  return scorp_adam_step_guarded(tensors, n, beta1, beta2, eps, step, nullptr, stream_);
}

extern "C" int scorp_adam_step_guarded(const ScorpAdamTensor *tensors, int32_t n, double beta1, double beta2, double eps,
                                       int32_t step, const uint32_t *skip_if_nonzero, scorp_stream_t stream_) {
  return scorp_adam_step_guarded_ex(tensors, n, beta1, beta2, eps, step, skip_if_nonzero, nullptr, stream_);
}

extern "C" int scorp_adam_step_guarded_ex(const ScorpAdamTensor *tensors, int32_t n, double beta1, double beta2, double eps,
                                          int32_t step, const uint32_t *skip_if_nonzero, uint32_t *skipped_counter,
                                          scorp_stream_t stream_) {
  if (n < 0 || n > SCORP_ADAM_MAX_TENSORS || (n > 0 && !tensors) || step < 1) {
    set_error("bad arguments to scorp_adam_step (n=%d, step=%d)", n, step); return SCORP_ERR_INVALID;
  }
  hipStream_t stream = (hipStream_t)stream_;
  AdamPack pk;
  pk.n = 0;
  uint32_t blocks = 0;
  for (int k = 0; k < n; k++) {
    if (tensors[k].numel == 0) continue;
    if (!tensors[k].param || !tensors[k].grad || !tensors[k].exp_avg || !tensors[k].exp_avg_sq) {
      set_error("scorp_adam_step: NULL pointer in tensor %d", k); return SCORP_ERR_INVALID;
    }
    pk.t[pk.n] = tensors[k];
    pk.first_block[pk.n] = blocks;
    blocks += (uint32_t)((tensors[k].numel + kAdamPerBlock - 1) / kAdamPerBlock);
    pk.n++;
  }
  pk.first_block[pk.n] = blocks;
  if (blocks == 0) return SCORP_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  {
    ProfScope prof(kKAdam, stream);
    adam_kernel<<<blocks, 256, 0, stream>>>(pk, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                                            (float)bc1, (float)(1.0 / sqrt(bc2)), skip_if_nonzero, skipped_counter);
  }
  SCORP_KERNEL_CHECK("adam", 0, stream);
  return SCORP_OK;
}

namespace scorp {
namespace {
__global__ void __launch_bounds__(256)
render_tail_kernel(const float *__restrict__ depth, const float *__restrict__ alpha, size_t HW, const int32_t *__restrict__ radii,
                   int N, float *__restrict__ out_depth, uint8_t *__restrict__ out_visible) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < HW) out_depth[i] = nan_to_num00(depth[i] / alpha[i]);
  if (i < (size_t)N) out_visible[i] = radii[i] > 0 ? 1 : 0;
}
__global__ void __launch_bounds__(256)
render_tail_backward_kernel(const float *__restrict__ g_out, const float *__restrict__ depth, const float *__restrict__ alpha,
                            size_t HW, float *__restrict__ g_depth, float *__restrict__ g_alpha) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const float d = depth[i], a = alpha[i], q = d / a;
  // nan_to_num passes the gradient only where its input was finite; PyTorch then still forms 0 / 0 = NaN at the
  // empty pixels (alpha = 0), which the rasterizer never reads: plain zeros here
  const bool pass = a != 0.0f && q == q && fabsf(q) != __builtin_inff();
  const float g = pass ? g_out[i] : 0.0f;
  g_depth[i] = pass ? g / a : 0.0f;
  g_alpha[i] = pass ? -g * d / (a * a) : 0.0f;
}
// One pose hypothesis' disagreement with a target view, from the rasterizer's raw outputs: sum over the pixels of
// |alpha - alpha*| + |nan_to_num(depth / alpha) - depth*| (the normalisation of render_tail_kernel, not stored), scaled
// and added to one device float: a grid-stride pass, wave and workgroup sums, one float atomic per workgroup.
__global__ void __launch_bounds__(256)
pose_score_kernel(const float *__restrict__ depth, const float *__restrict__ alpha, const float *__restrict__ tgt_depth,
                  const float *__restrict__ tgt_alpha, size_t HW, float scale, float *__restrict__ acc, bool vec4) {
  __shared__ float s_part[4];
  float sum = 0.0f;
  const size_t stride = (size_t)gridDim.x * 256;
  auto term = [](float d, float a, float td, float ta) { return fabsf(a - ta) + fabsf(nan_to_num00(d / a) - td); };
  if (vec4) {   // all four maps 16-byte aligned: one 16-byte load per map and thread, all in flight together
    const size_t Q = HW / 4;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < Q; q += stride) {
      const float4 d = reinterpret_cast<const float4 *>(depth)[q], a = reinterpret_cast<const float4 *>(alpha)[q];
      const float4 td = reinterpret_cast<const float4 *>(tgt_depth)[q], ta = reinterpret_cast<const float4 *>(tgt_alpha)[q];
      sum += (term(d.x, a.x, td.x, ta.x) + term(d.y, a.y, td.y, ta.y)) + (term(d.z, a.z, td.z, ta.z) + term(d.w, a.w, td.w, ta.w));
    }
    for (size_t i = 4 * Q + (size_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += stride) sum += term(depth[i], alpha[i], tgt_depth[i], tgt_alpha[i]);
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += stride) sum += term(depth[i], alpha[i], tgt_depth[i], tgt_alpha[i]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, scale * (s_part[0] + s_part[1] + s_part[2] + s_part[3]));
}
}  // namespace
}  // namespace scorp

extern "C" int scorp_gs3d_pose_score_accumulate(const float *depth, const float *alpha, const float *tgt_depth,
                                                const float *tgt_alpha, int64_t HW, float scale, float *acc,
                                                scorp_stream_t stream_) {
  if (HW < 0 || !acc || (HW > 0 && (!depth || !alpha || !tgt_depth || !tgt_alpha))) {
    set_error("bad argument to scorp_gs3d_pose_score_accumulate"); return SCORP_ERR_INVALID;
  }
  if (HW == 0) return SCORP_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t want = ((size_t)HW + 1023) / 1024;   // 4 pixels per thread
  const bool vec4 = (((uintptr_t)depth | (uintptr_t)alpha | (uintptr_t)tgt_depth | (uintptr_t)tgt_alpha) & 15) == 0;
  pose_score_kernel<<<(unsigned)(want < 2048 ? want : 2048), 256, 0, stream>>>(depth, alpha, tgt_depth, tgt_alpha, (size_t)HW,
                                                                               scale, acc, vec4);
  SCORP_KERNEL_CHECK("pose_score", 0, stream);
  return SCORP_OK;
}

extern "C" int scorp_gs3d_render_tail(const float *depth, const float *alpha, int64_t HW, const int32_t *radii, int32_t N,
                                      float *out_depth, uint8_t *out_visible, scorp_stream_t stream_) {
  if (HW < 0 || N < 0 || (HW > 0 && (!depth || !alpha || !out_depth)) || (N > 0 && (!radii || !out_visible))) {
    set_error("bad argument to scorp_gs3d_render_tail"); return SCORP_ERR_INVALID;
  }
  const size_t n = (size_t)HW > (size_t)N ? (size_t)HW : (size_t)N;
  if (n == 0) return SCORP_OK;
  hipStream_t stream = (hipStream_t)stream_;
  render_tail_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(depth, alpha, (size_t)HW, radii, N, out_depth, out_visible);
  SCORP_KERNEL_CHECK("render_tail", 0, stream);
  return SCORP_OK;
}

extern "C" int scorp_gs3d_render_tail_backward(const float *g_out, const float *depth, const float *alpha, int64_t HW,
                                               float *g_depth, float *g_alpha, scorp_stream_t stream_) {
  if (HW < 0 || (HW > 0 && (!g_out || !depth || !alpha || !g_depth || !g_alpha))) {
    set_error("bad argument to scorp_gs3d_render_tail_backward"); return SCORP_ERR_INVALID;
  }
  if (HW == 0) return SCORP_OK;
  hipStream_t stream = (hipStream_t)stream_;
  render_tail_backward_kernel<<<(unsigned)(((size_t)HW + 255) / 256), 256, 0, stream>>>(g_out, depth, alpha, (size_t)HW, g_depth, g_alpha);
  SCORP_KERNEL_CHECK("render_tail_backward", 0, stream);
  return SCORP_OK;
}
