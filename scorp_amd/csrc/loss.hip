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
constexpr int kHO = 6;             // output columns per horizontal work item
constexpr int kHG = (kLT + kHO - 1) / kHO;  // groups per patch row (6): kLP * kHG = 252 items <= 256 threads
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

// Workgroup -> (channel, tile) with XCD locality: the hardware deals consecutive workgroup ids round-robin to the 8
// XCDs, each with its own L2.  Giving XCD x the x-th contiguous eighth of the (channel, tile-row, tile-column) order
// keeps neighbouring tiles — which share their 5-pixel halos — in one L2 instead of refetching them over the fabric.
struct LossTile { int ch, x0, y0, logical; bool valid; };
__device__ __forceinline__ LossTile loss_tile(int gx, int gy, int C) {
  const int total = gx * gy * C, per = (total + 7) / 8;
  const int b = blockIdx.x, logical = (b & 7) * per + (b >> 3);
  LossTile t;
  t.valid = (b >> 3) < per && logical < total;
  t.logical = logical;
  t.ch = logical / (gx * gy);
  const int rem = logical - t.ch * gx * gy;
  t.y0 = (rem / gx) * kLT; t.x0 = (rem % gx) * kLT;
  return t;
}

__global__ void __launch_bounds__(256)
ssim_l1_forward_kernel(const float *__restrict__ img, const float *__restrict__ gt, const float *__restrict__ mask,
                       int C, int H, int W, Window win, float *__restrict__ dmaps, float *__restrict__ partials) {
  __shared__ float s_x[kLP][kLPS], s_y[kLP][kLPS];
  __shared__ float s_h[4][kLP][kLHS];   // mu1, mu2, E[x^2] + E[y^2] (only the sum enters SSIM and d/dx), E[xy]
  __shared__ float s_red[4];
  const LossTile lt = loss_tile((W + kLT - 1) / kLT, (H + kLT - 1) / kLT, C);
  if (!lt.valid) return;
  const int ch = lt.ch, x0 = lt.x0, y0 = lt.y0;
  const size_t HW = (size_t)H * W;
  const float *xp = img + ch * HW, *yp = gt + ch * HW;
  {
    // The patch: 1764 elements = 7 per thread.  All of a thread's loads leave together, from clamped (always valid)
    // addresses, and out-of-image elements are zeroed by a select afterwards: as a loop with the bounds test around the
    // loads this was seven dependent round trips to memory per workgroup - most of its life.
    constexpr int kPer = (kLP * kLP + 255) / 256;
    float xv[kPer], yv[kPer], mv[kPer];
    bool in[kPer];
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int i = min(threadIdx.x + 256 * j, kLP * kLP - 1);
      const int r = i / kLP, c = i - r * kLP;
      const int gy = y0 + r - kLH, gx = x0 + c - kLH;
      in[j] = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const size_t p = (size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
      xv[j] = xp[p]; yv[j] = yp[p];
      mv[j] = mask ? mask[p] : 1.0f;
    }
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int i = threadIdx.x + 256 * j;
      if (i < kLP * kLP) {
        const int r = i / kLP, c = i - r * kLP;
        s_x[r][c] = in[j] ? xv[j] * mv[j] : 0.0f;
        s_y[r][c] = in[j] ? yv[j] * mv[j] : 0.0f;
      }
    }
  }
  __syncthreads();
  // horizontal pass: one work item = one patch row x 6 adjacent output columns, 42 x 6 = 252 items = one pass of the
  // workgroup; the products x^2, y^2, xy are formed once per loaded element, not once per tap
  {
    const int it = threadIdx.x;
    const int r = it / kHG, c0 = (it % kHG) * kHO;
    if (it < kLP * kHG) {
      float xs[kHO + 10], ys[kHO + 10], ss[kHO + 10], xy[kHO + 10];
#pragma unroll
      for (int k = 0; k < kHO + 10; k++) {
        const bool in = c0 + k < kLP;   // the last group of a row covers columns 30..35 of 32
        xs[k] = in ? s_x[r][c0 + k] : 0.0f; ys[k] = in ? s_y[r][c0 + k] : 0.0f;
        ss[k] = xs[k] * xs[k] + ys[k] * ys[k]; xy[k] = xs[k] * ys[k];
      }
#pragma unroll
      for (int o = 0; o < kHO; o++) {
        float m1 = 0, m2 = 0, ess = 0, e12 = 0;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          const float w = win.w[k];
          m1 += w * xs[o + k]; m2 += w * ys[o + k]; ess += w * ss[o + k]; e12 += w * xy[o + k];
        }
        if (c0 + o < kLT) { s_h[0][r][c0 + o] = m1; s_h[1][r][c0 + o] = m2; s_h[2][r][c0 + o] = ess; s_h[3][r][c0 + o] = e12; }
      }
    }
  }
  __syncthreads();
  // vertical pass: thread = one column x 4 adjacent rows
  const int c = threadIdx.x % kLT, r0 = (threadIdx.x / kLT) * 4;
  float l1_sum = 0.0f, ssim_sum = 0.0f;
  float col[4][14];
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int k = 0; k < 14; k++) col[q][k] = s_h[q][r0 + k][c];
#pragma unroll
  for (int o = 0; o < 4; o++) {
    float m1 = 0, m2 = 0, ess = 0, e12 = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float w = win.w[k];
      m1 += w * col[0][o + k]; m2 += w * col[1][o + k]; ess += w * col[2][o + k]; e12 += w * col[3][o + k];
    }
    const int gy = y0 + r0 + o, gx = x0 + c;
    if (gy < H && gx < W) {
      const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
      const float s12 = e12 - m12;
      const float A1 = 2 * m12 + kC1, A2 = 2 * s12 + kC2, B1 = m1s + m2s + kC1, B2 = (ess - m1s - m2s) + kC2;
      const float inv = 1.0f / (B1 * B2);
      const float ssim = A1 * A2 * inv;
      ssim_sum += ssim;
      const float xv = s_x[r0 + o + kLH][c + kLH], yv = s_y[r0 + o + kLH][c + kLH];
      l1_sum += fabsf(xv - yv);
      if (dmaps) {
        const size_t p = (size_t)gy * W + gx;
        const size_t CHW = (size_t)C * HW;
        dmaps[ch * HW + p] = (2 * m2 * (A2 - A1) - 2 * m1 * ssim * (B2 - B1)) * inv;  // d ssim / d mu1
        dmaps[CHW + ch * HW + p] = -ssim / B2;                                        // d ssim / d E[x^2]
        dmaps[2 * CHW + ch * HW + p] = 2 * A1 * inv;                                   // d ssim / d E[xy]
      }
    }
  }
  const float bl1 = block_sum_256(l1_sum, s_red);
  const float bss = block_sum_256(ssim_sum, s_red);
  if (threadIdx.x == 0) {
    const int b = lt.logical;
    partials[2 * b] = bl1;
    partials[2 * b + 1] = bss;
  }
}

// 1024 threads, float2 loads four at a time in flight, then a fixed-order tree in double: a launch-latency-sized kernel
__global__ void __launch_bounds__(1024)
loss_finalize_kernel(const float *__restrict__ partials, int nblocks, double n_elems, float lambda, float *__restrict__ out) {
  __shared__ double s_a[1024], s_b[1024];
  const float2 *p2 = reinterpret_cast<const float2 *>(partials);
  double a = 0, b = 0;
  for (int i0 = threadIdx.x; i0 < nblocks; i0 += 4 * 1024) {
    float2 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int i = i0 + 1024 * j; v[j] = i < nblocks ? p2[i] : make_float2(0.0f, 0.0f); }
#pragma unroll
    for (int j = 0; j < 4; j++) { a += v[j].x; b += v[j].y; }
  }
  s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
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
                        const float *__restrict__ dmaps, int C, int H, int W, Window win, float lambda, float inv_n,
                        const float *__restrict__ grad_out, float *__restrict__ grad_img) {
  // The three derivative maps go through the same two LDS buffers one after the other (13 KB per workgroup instead
  // of 39 KB: occupancy, not arithmetic, limits this kernel); each thread accumulates its four output pixels.
  __shared__ float s_m[kLP][kLPS];
  __shared__ float s_h[kLP][kLHS];
  const LossTile lt = loss_tile((W + kLT - 1) / kLT, (H + kLT - 1) / kLT, C);
  if (!lt.valid) return;
  const int ch = lt.ch, x0 = lt.x0, y0 = lt.y0;
  const size_t HW = (size_t)H * W, CHW = (size_t)C * HW;
  const int c = threadIdx.x % kLT, r0 = (threadIdx.x / kLT) * 4;
  const float go = grad_out ? grad_out[0] : 1.0f;
  float xv[4], yv[4], mk[4], acc[4];
  // the thread's own four pixels: all loads issued together from clamped addresses, out-of-image ones zeroed by a
  // select (with the bounds test around the loads each pixel was two dependent round trips to memory)
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int gy = y0 + r0 + o, gx = x0 + c;
    const size_t p = (size_t)min(gy, H - 1) * W + min(gx, W - 1);
    mk[o] = mask ? mask[p] : 1.0f;
    xv[o] = img[ch * HW + p]; yv[o] = gt[ch * HW + p];
  }
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const bool in = y0 + r0 + o < H && x0 + c < W;
    mk[o] = in ? mk[o] : 0.0f;
    xv[o] *= mk[o]; yv[o] *= mk[o];
    acc[o] = 0.0f;
  }
  // the patch of map q+1 is fetched into registers while map q is convolved (7 pixels per thread)
  constexpr int kPerThread = (kLP * kLP + 255) / 256;
  float pre[kPerThread];
  auto fetch = [&](int q) {
    const float *map = dmaps + q * CHW + ch * HW;
#pragma unroll
    for (int j = 0; j < kPerThread; j++) {   // clamped, unconditional loads; the select comes when the value is stored
      const int i = min(threadIdx.x + 256 * j, kLP * kLP - 1);
      const int r = i / kLP, cc = i - r * kLP;
      const int gy = y0 + r - kLH, gx = x0 + cc - kLH;
      pre[j] = map[(size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];
    }
  };
  bool pre_in[kPerThread];   // the same for all three maps
#pragma unroll
  for (int j = 0; j < kPerThread; j++) {
    const int i = min(threadIdx.x + 256 * j, kLP * kLP - 1);
    const int r = i / kLP, cc = i - r * kLP;
    const int gy = y0 + r - kLH, gx = x0 + cc - kLH;
    pre_in[j] = gy >= 0 && gy < H && gx >= 0 && gx < W;
  }
  fetch(0);
#pragma unroll 1
  for (int q = 0; q < 3; q++) {
#pragma unroll
    for (int j = 0; j < kPerThread; j++) {
      const int i = threadIdx.x + 256 * j;
      if (i < kLP * kLP) s_m[i / kLP][i % kLP] = pre_in[j] ? pre[j] : 0.0f;
    }
    if (q < 2) fetch(q + 1);
    __syncthreads();
    {  // horizontal pass: 42 rows x 6 groups of 6 output columns = 252 work items, one pass
      const int it = threadIdx.x;
      const int r = it / kHG, c0 = (it % kHG) * kHO;
      if (it < kLP * kHG) {
        float v[kHO + 10];
#pragma unroll
        for (int k = 0; k < kHO + 10; k++) v[k] = c0 + k < kLP ? s_m[r][c0 + k] : 0.0f;
#pragma unroll
        for (int o = 0; o < kHO; o++) {
          float sacc = 0;
#pragma unroll
          for (int k = 0; k < 11; k++) sacc += win.w[k] * v[o + k];
          if (c0 + o < kLT) s_h[r][c0 + o] = sacc;
        }
      }
    }
    __syncthreads();
    float col[14];
#pragma unroll
    for (int k = 0; k < 14; k++) col[k] = s_h[r0 + k][c];
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float cv = 0;
#pragma unroll
      for (int k = 0; k < 11; k++) cv += win.w[k] * col[o + k];
      acc[o] += q == 0 ? cv : (q == 1 ? 2.0f * xv[o] * cv : yv[o] * cv);   // ca + 2 x cb + y cd
    }
  }
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int gy = y0 + r0 + o, gx = x0 + c;
    if (gy < H && gx < W) {
      const float diff = xv[o] - yv[o];
      const float sgn = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
      const float g = (1.0f - lambda) * sgn - lambda * acc[o];
      grad_img[ch * HW + (size_t)gy * W + gx] = go * inv_n * mk[o] * g;
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
  const int grid = (loss_blocks(C, H, W) + 7) / 8 * 8;
  const Window win = make_window();
  {
    ProfScope prof(kKLossForward, stream);
    ssim_l1_forward_kernel<<<grid, 256, 0, stream>>>(img, gt, mask, C, H, W, win, need_backward ? dmaps : nullptr, partials);
  }
  SCORP_KERNEL_CHECK("ssim_l1_forward", 0, stream);
  loss_finalize_kernel<<<1, 1024, 0, stream>>>(partials, loss_blocks(C, H, W), (double)C * H * W, lambda_dssim, out_loss3);
  SCORP_KERNEL_CHECK("loss_finalize", 0, stream);
  return SCORP_OK;
}

extern "C" int scorp_loss_l1_ssim_backward(const float *img, const float *gt, const float *mask, int32_t C, int32_t H,
                                           int32_t W, float lambda_dssim, const void *workspace, const float *grad_out,
                                           float *grad_img, scorp_stream_t stream_) {
  if (!img || !gt || !workspace || !grad_img) { set_error("NULL argument to scorp_loss_l1_ssim_backward"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const int grid = (loss_blocks(C, H, W) + 7) / 8 * 8;
  const Window win = make_window();
  {
    ProfScope prof(kKLossBackward, stream);
    ssim_l1_backward_kernel<<<grid, 256, 0, stream>>>(img, gt, mask, (const float *)workspace, C, H, W, win, lambda_dssim,
                                                      (float)(1.0 / ((double)C * H * W)), grad_out, grad_img);
  }
  SCORP_KERNEL_CHECK("ssim_l1_backward", 0, stream);
  return SCORP_OK;
}
