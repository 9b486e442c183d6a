// loss.hip — fused photometric loss of the training step for gfx950:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean SSIM(x, y)),   optionally with x, y multiplied by a mask.
//
// Replaces, for the training harnesses, the torch formulation at train_3dgs.py:106-107 / post_refine_gs.py:103-111
// built from gs3dgs/utils/loss_utils.py:17-73 (five depthwise 11x11 convolutions forward, their autograd
// mirrors backward).  Same definition: Gaussian window sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2,
// mean over all C*H*W elements.  The 2-D window is the outer product of the 1-D one, so it is applied separably.
//
// Forward : one pass over 32x32 tiles (+5 halo) per channel: the five windowed moments, the SSIM value, |x-y|,
//           per-block partial sums, and the three derivative maps d ssim/d{mu1, E[x^2], E[xy]}.
// Backward: one pass: windowed sums of the three maps (the window is symmetric), combined with x and y.
#include "common.hpp"

namespace scorp {
namespace {

constexpr int kLT = 32;            // output tile edge
constexpr int kLH = 5;             // halo
constexpr int kLP = kLT + 2 * kLH; // patch edge (42)
constexpr int kLPS = kLP + 3;      // patch row stride 45: the 4-wide horizontal work items hit 32 distinct banks
constexpr int kLHS = kLT + 1;      // row stride of the horizontal-pass results (33: conflict-free writes)
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct Window { float w[11]; };

// Window exactly as the reference builds it (loss_utils.py:23-26): python-double exp rounded to fp32, fp32 sum.
Window make_window() {
  Window win;
  float s = 0.0f;
  for (int i = 0; i < 11; i++) { win.w[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += win.w[i]; }
  for (int i = 0; i < 11; i++) win.w[i] /= s;
  return win;
}

__device__ __forceinline__ float block_sum_256(float v, float *s_red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ void __launch_bounds__(256)
ssim_l1_forward_kernel(const float *__restrict__ img, const float *__restrict__ gt, const float *__restrict__ mask,
                       int H, int W, Window win, float *__restrict__ dmaps, float *__restrict__ partials) {
  __shared__ float s_x[kLP][kLPS], s_y[kLP][kLPS];
  __shared__ float s_h[5][kLP][kLHS];
  __shared__ float s_red[4];
  const int ch = blockIdx.z;
  const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
  const size_t HW = (size_t)H * W;
  const float *xp = img + ch * HW, *yp = gt + ch * HW;
  for (int i = threadIdx.x; i < kLP * kLP; i += 256) {
    const int r = i / kLP, c = i % kLP;
    const int gy = y0 + r - kLH, gx = x0 + c - kLH;
    float xv = 0.0f, yv = 0.0f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const size_t p = (size_t)gy * W + gx;
      const float m = mask ? mask[p] : 1.0f;
      xv = xp[p] * m; yv = yp[p] * m;
    }
    s_x[r][c] = xv; s_y[r][c] = yv;
  }
  __syncthreads();
  // horizontal pass: each work item = one patch row x 4 adjacent output columns
  for (int it = threadIdx.x; it < kLP * (kLT / 4); it += 256) {
    const int r = it / (kLT / 4), c0 = (it % (kLT / 4)) * 4;
    float xs[14], ys[14];
#pragma unroll
    for (int k = 0; k < 14; k++) { xs[k] = s_x[r][c0 + k]; ys[k] = s_y[r][c0 + k]; }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = win.w[k], xv = xs[o + k], yv = ys[o + k];
        m1 += w * xv; m2 += w * yv; e11 += w * xv * xv; e22 += w * yv * yv; e12 += w * xv * yv;
      }
      s_h[0][r][c0 + o] = m1; s_h[1][r][c0 + o] = m2; s_h[2][r][c0 + o] = e11; s_h[3][r][c0 + o] = e22; s_h[4][r][c0 + o] = e12;
    }
  }
  __syncthreads();
  // vertical pass: thread = one column x 4 adjacent rows
  const int c = threadIdx.x % kLT, r0 = (threadIdx.x / kLT) * 4;
  float l1_sum = 0.0f, ssim_sum = 0.0f;
  float col[5][14];
#pragma unroll
  for (int q = 0; q < 5; q++)
#pragma unroll
    for (int k = 0; k < 14; k++) col[q][k] = s_h[q][r0 + k][c];
#pragma unroll
  for (int o = 0; o < 4; o++) {
    float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float w = win.w[k];
      m1 += w * col[0][o + k]; m2 += w * col[1][o + k]; e11 += w * col[2][o + k]; e22 += w * col[3][o + k]; e12 += w * col[4][o + k];
    }
    const int gy = y0 + r0 + o, gx = x0 + c;
    if (gy < H && gx < W) {
      const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
      const float s1 = e11 - m1s, s2 = e22 - m2s, s12 = e12 - m12;
      const float A1 = 2 * m12 + kC1, A2 = 2 * s12 + kC2, B1 = m1s + m2s + kC1, B2 = s1 + s2 + kC2;
      const float inv = 1.0f / (B1 * B2);
      const float ssim = A1 * A2 * inv;
      ssim_sum += ssim;
      const float xv = s_x[r0 + o + kLH][c + kLH], yv = s_y[r0 + o + kLH][c + kLH];
      l1_sum += fabsf(xv - yv);
      if (dmaps) {
        const size_t p = (size_t)gy * W + gx;
        const size_t CHW = (size_t)gridDim.z * HW;
        dmaps[ch * HW + p] = (2 * m2 * (A2 - A1) - 2 * m1 * ssim * (B2 - B1)) * inv;  // d ssim / d mu1
        dmaps[CHW + ch * HW + p] = -ssim / B2;                                        // d ssim / d E[x^2]
        dmaps[2 * CHW + ch * HW + p] = 2 * A1 * inv;                                   // d ssim / d E[xy]
      }
    }
  }
  const float bl1 = block_sum_256(l1_sum, s_red);
  const float bss = block_sum_256(ssim_sum, s_red);
  if (threadIdx.x == 0) {
    const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partials[2 * b] = bl1;
    partials[2 * b + 1] = bss;
  }
}

__global__ void __launch_bounds__(256)
loss_finalize_kernel(const float *__restrict__ partials, int nblocks, double n_elems, float lambda, float *__restrict__ out) {
  __shared__ double s_a[256], s_b[256];
  double a = 0, b = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) { s_a[threadIdx.x] += s_a[threadIdx.x + off]; s_b[threadIdx.x] += s_b[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double l1 = s_a[0] / n_elems, ss = s_b[0] / n_elems;
    out[0] = (float)((1.0 - lambda) * l1 + lambda * (1.0 - ss));
    out[1] = (float)l1;
    out[2] = (float)ss;
  }
}

__global__ void __launch_bounds__(256)
ssim_l1_backward_kernel(const float *__restrict__ img, const float *__restrict__ gt, const float *__restrict__ mask,
                        const float *__restrict__ dmaps, int H, int W, Window win, float lambda, float inv_n,
                        const float *__restrict__ grad_out, float *__restrict__ grad_img) {
  __shared__ float s_m[3][kLP][kLPS];
  __shared__ float s_h[3][kLP][kLHS];
  const int ch = blockIdx.z;
  const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
  const size_t HW = (size_t)H * W, CHW = (size_t)gridDim.z * HW;
  for (int i = threadIdx.x; i < kLP * kLP; i += 256) {
    const int r = i / kLP, c = i % kLP;
    const int gy = y0 + r - kLH, gx = x0 + c - kLH;
    float a = 0, b = 0, d = 0;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const size_t p = ch * HW + (size_t)gy * W + gx;
      a = dmaps[p]; b = dmaps[CHW + p]; d = dmaps[2 * CHW + p];
    }
    s_m[0][r][c] = a; s_m[1][r][c] = b; s_m[2][r][c] = d;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < kLP * (kLT / 4); it += 256) {
    const int r = it / (kLT / 4), c0 = (it % (kLT / 4)) * 4;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      float v[14];
#pragma unroll
      for (int k = 0; k < 14; k++) v[k] = s_m[q][r][c0 + k];
#pragma unroll
      for (int o = 0; o < 4; o++) {
        float s = 0;
#pragma unroll
        for (int k = 0; k < 11; k++) s += win.w[k] * v[o + k];
        s_h[q][r][c0 + o] = s;
      }
    }
  }
  __syncthreads();
  const int c = threadIdx.x % kLT, r0 = (threadIdx.x / kLT) * 4;
  const float go = grad_out ? grad_out[0] : 1.0f;
  float col[3][14];
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int k = 0; k < 14; k++) col[q][k] = s_h[q][r0 + k][c];
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int gy = y0 + r0 + o, gx = x0 + c;
    if (gy < H && gx < W) {
      float ca = 0, cb = 0, cd = 0;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = win.w[k];
        ca += w * col[0][o + k]; cb += w * col[1][o + k]; cd += w * col[2][o + k];
      }
      const size_t p = (size_t)gy * W + gx;
      const float m = mask ? mask[p] : 1.0f;
      const float xv = img[ch * HW + p] * m, yv = gt[ch * HW + p] * m;
      const float diff = xv - yv;
      const float sgn = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
      const float g = (1.0f - lambda) * sgn - lambda * (ca + 2.0f * xv * cb + yv * cd);
      grad_img[ch * HW + p] = go * inv_n * m * g;
    }
  }
}

}  // namespace
}  // namespace scorp

using namespace scorp;

static inline int loss_blocks(int C, int H, int W) { return ((W + kLT - 1) / kLT) * ((H + kLT - 1) / kLT) * C; }

// workspace = derivative maps [3][C][H][W] followed by per-block partial sums [blocks][2]
extern "C" size_t scorp_loss_workspace_bytes(int32_t C, int32_t H, int32_t W) {
  return align_up((size_t)3 * C * H * W * 4, 256) + align_up((size_t)loss_blocks(C, H, W) * 8, 256);
}

extern "C" int scorp_loss_l1_ssim_forward(const float *img, const float *gt, const float *mask, int32_t C, int32_t H,
                                          int32_t W, float lambda_dssim, float *out_loss3, void *workspace,
                                          size_t workspace_bytes, int32_t need_backward, scorp_stream_t stream_) {
  if (!img || !gt || !out_loss3 || !workspace) { set_error("NULL argument to scorp_loss_l1_ssim_forward"); return SCORP_ERR_INVALID; }
  if (C <= 0 || H <= 0 || W <= 0) { set_error("bad image shape"); return SCORP_ERR_INVALID; }
  if (workspace_bytes < scorp_loss_workspace_bytes(C, H, W) || ((uintptr_t)workspace & 15)) {
    set_error("loss workspace too small or misaligned"); return SCORP_ERR_INVALID;
  }
  hipStream_t stream = (hipStream_t)stream_;
  float *dmaps = (float *)workspace;
  float *partials = (float *)((char *)workspace + align_up((size_t)3 * C * H * W * 4, 256));
  const dim3 grid((W + kLT - 1) / kLT, (H + kLT - 1) / kLT, C);
  const Window win = make_window();
  {
    ProfScope prof(kKLossForward, stream);
    ssim_l1_forward_kernel<<<grid, 256, 0, stream>>>(img, gt, mask, H, W, win, need_backward ? dmaps : nullptr, partials);
  }
  SCORP_KERNEL_CHECK("ssim_l1_forward", 0, stream);
  loss_finalize_kernel<<<1, 256, 0, stream>>>(partials, loss_blocks(C, H, W), (double)C * H * W, lambda_dssim, out_loss3);
  SCORP_KERNEL_CHECK("loss_finalize", 0, stream);
  return SCORP_OK;
}

extern "C" int scorp_loss_l1_ssim_backward(const float *img, const float *gt, const float *mask, int32_t C, int32_t H,
                                           int32_t W, float lambda_dssim, const void *workspace, const float *grad_out,
                                           float *grad_img, scorp_stream_t stream_) {
  if (!img || !gt || !workspace || !grad_img) { set_error("NULL argument to scorp_loss_l1_ssim_backward"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((W + kLT - 1) / kLT, (H + kLT - 1) / kLT, C);
  const Window win = make_window();
  {
    ProfScope prof(kKLossBackward, stream);
    ssim_l1_backward_kernel<<<grid, 256, 0, stream>>>(img, gt, mask, (const float *)workspace, H, W, win, lambda_dssim,
                                                      (float)(1.0 / ((double)C * H * W)), grad_out, grad_img);
  }
  SCORP_KERNEL_CHECK("ssim_l1_backward", 0, stream);
  return SCORP_OK;
}
