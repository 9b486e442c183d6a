// loss.hip — fused photometric loss of the training step for gfx950:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean SSIM(x, y)),   optionally with x, y multiplied by a mask.
//
// Replaces, for the training harnesses, the torch formulation at train_3dgs.py:106-107 / post_refine_gs.py:103-111
// built from gs3dgs/utils/loss_utils.py:17-73 (five depthwise 11x11 convolutions forward, their autograd
// mirrors backward).  Same definition: Gaussian window sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2,
// mean over all C*H*W elements.  The 2-D window is the outer product of the 1-D one, so it is applied separably.
//
// Forward : one pass, a wave per 64-column strip streaming down the image: the five windowed moments, the SSIM
//           value, |x-y|, per-wave partial sums, and the three derivative maps d ssim/d{mu1, E[x^2], E[xy]}.
// Backward: one pass over 32x32 tiles (+5 halo): windowed sums of the three maps (the window is symmetric),
//           combined with x and y.
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

// ---------------------------------------------------------------------------------------------------------
// Forward: ONE WAVE per strip of 64 columns x kStripRows rows of one channel, no workgroup barriers.  The wave
// walks its strip top to bottom one image row at a time: the row (64 + 10 halo columns) is loaded two rows ahead,
// (x, y, x^2 + y^2, xy) go through a per-wave LDS row buffer from which every lane reads its 11 taps (horizontal
// pass), and the vertical pass lives in registers: the last 11 rows' horizontal sums form a ring that is indexed
// statically because the row loop is unrolled by 11.  (The first form, 32x32 tiles per workgroup with the two passes
// separated by barriers as in the backward below, spent most of a workgroup's life waiting and needed 36 KB of LDS:
// 54 us against 49.)
// ---------------------------------------------------------------------------------------------------------
#ifndef SCORP_LOSS_STRIP_ITERS
#define SCORP_LOSS_STRIP_ITERS 3
#endif
constexpr int kStripIters = SCORP_LOSS_STRIP_ITERS;     // unrolled-by-11 groups of rows per strip
constexpr int kStripRows = 11 * kStripIters - 10;       // output rows per strip: 23 (3 975 waves = 3.9 per SIMD; rocprof, same box:
                                                        // 2 iterations 50.6 us, 3: 42.0, 4: 44.5, 5: 45.8, 6: 53.1 - the kernel
                                                        // is short of waves, not of arithmetic)
constexpr int kRowBuf = 80;                             // floats per row-buffer array (74 used)

struct StripJob { int ch, cx0, ry0; bool valid; };
__device__ __forceinline__ StripJob strip_job(int C, int H, int W) {
  const int sx = (W + 63) / 64, sy = (H + kStripRows - 1) / kStripRows;
  // readfirstlane: the wave index is uniform, but derived from threadIdx the compiler would carry the whole job
  // (channel, strip origin, every row base address) in vector registers
  const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  StripJob j;
  j.valid = wave < C * sx * sy;
  j.ch = wave / (sx * sy);
  const int rem = wave - j.ch * sx * sy;
  j.ry0 = (rem / sx) * kStripRows; j.cx0 = (rem % sx) * 64;
  return j;
}
static inline int strip_waves(int C, int H, int W) { return C * ((W + 63) / 64) * ((H + kStripRows - 1) / kStripRows); }

__global__ void __launch_bounds__(256)
ssim_l1_forward_strip_kernel(const float *__restrict__ img, const float *__restrict__ gt, const float *__restrict__ mask,
                             int C, int H, int W, Window win, float *__restrict__ dmaps, float *__restrict__ partials) {
  // [wave][column] of (x, y, x^2 + y^2, xy): a tap is ONE 16-byte LDS read.  (Four arrays [map][column], two ds_read2_b32 per
  // tap, until late in round 6: same box 44.6 -> 40.2 us - the horizontal pass was paying the LDS pipe per instruction.)
  __shared__ __attribute__((aligned(16))) float4 s_row4[4][kRowBuf];
  const StripJob job = strip_job(C, H, W);
  if (!job.valid) return;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float4 *buf4 = s_row4[wv];
  const int ch = job.ch, cx0 = job.cx0, ry0 = job.ry0;
  const size_t HW = (size_t)H * W, CHW = (size_t)C * HW;
  const float *xp = img + ch * HW, *yp = gt + ch * HW;
  const int gx = cx0 + lane;                               // this lane's output column
  const int ca = cx0 - kLH + lane, cb = cx0 + 64 - kLH + lane;   // columns of its one / two (lane < 10) patch elements
  const bool cin_a = ca >= 0 && ca < W, cin_b = lane < 2 * kLH && cb < W;
  const int ca_c = min(max(ca, 0), W - 1), cb_c = min(cb, W - 1);
  struct RowRegs { float xa, ya, ma, xb, yb, mb; };
  // clamped, unconditional loads (zeroed by a select later); uniform row base + 32-bit lane offset, so that the loads
  // take the scalar-base addressing form instead of a 64-bit address computation per load
  const uint32_t oa = (uint32_t)ca_c, ob = (uint32_t)cb_c;
  auto fetch = [&](int r) {
    RowRegs v;
    const size_t ro = (size_t)min(max(r, 0), H - 1) * W;
    const float *xr = xp + ro, *yr = yp + ro;
    v.xa = xr[oa]; v.ya = yr[oa]; v.xb = xr[ob]; v.yb = yr[ob];
    v.ma = v.mb = 1.0f;
    if (mask) { const float *mr = mask + ro; v.ma = mr[oa]; v.mb = mr[ob]; }
    return v;
  };
  const int rbeg = ry0 - kLH;
  if (mask) {
    // A strip whose whole support (its rows +- 5, its columns +- 5) lies where the mask is zero - most of the image when the
    // mask is one object's silhouette (post_refine_gs.py:103-111) - has x = y = 0 everywhere: SSIM is the constant below,
    // |x - y| = 0, and the derivative maps are constants too.  One pass over the mask decides it; the values are formed by
    // the same instructions as in the loop below (hardware reciprocals of operands the compiler cannot fold), the sums in
    // the same order: same bits as the long way round.  (Pixels of a NaN / infinite image under a zero mask come out clean
    // here and NaN there.)
    // (unconditional loads from clamped rows and columns, ALL of the support's rows in flight together - one memory round
    // trip; the registers are free here.  A row clamped into the image is one of the support's own rows, a clamped column is
    // masked by cin_a / cin_b)
    bool any = false;
    {
      constexpr int kSupport = kStripRows + 2 * kLH;
      float ma[kSupport], mb[kSupport];
#pragma unroll
      for (int u = 0; u < kSupport; u++) {
        const float *mr = mask + (size_t)min(max(rbeg + u, 0), H - 1) * W;
        ma[u] = mr[oa]; mb[u] = mr[ob];
      }
#pragma unroll
      for (int u = 0; u < kSupport; u++) any |= (cin_a & (ma[u] != 0.0f)) | (cin_b & (mb[u] != 0.0f));
    }
    if (__ballot(any) == 0) {
      float c1 = kC1, c2 = kC2, zero = 0.0f;
      asm volatile("" : "+v"(c1), "+v"(c2), "+v"(zero));
      const float m1 = zero, m2 = zero, ess = zero, e12 = zero;
      const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
      const float s12 = e12 - m12;
      const float A1 = 2 * m12 + c1, A2 = 2 * s12 + c2, B1 = m1s + m2s + c1, B2 = (ess - m1s - m2s) + c2;
      const float rB2 = __builtin_amdgcn_rcpf(B2);
      const float inv = __builtin_amdgcn_rcpf(B1) * rB2;
      const float ssim = A1 * A2 * inv;
      float ssim_sum = 0.0f;
      for (int ro = ry0; ro < min(ry0 + kStripRows, H); ro++) {
        if (gx < W) {
          ssim_sum += ssim;
          if (dmaps) {
            float *d0 = dmaps + ch * HW + (size_t)ro * W;
            d0[(uint32_t)gx] = (2 * m2 * (A2 - A1) - 2 * m1 * ssim * (B2 - B1)) * inv;
            (d0 + CHW)[(uint32_t)gx] = -ssim * rB2;
            (d0 + 2 * CHW)[(uint32_t)gx] = 2 * A1 * inv;
          }
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) ssim_sum += __shfl_xor(ssim_sum, off, 64);
      if (lane == 0) {
        const int wave = blockIdx.x * 4 + wv;
        partials[2 * wave] = 0.0f;
        partials[2 * wave + 1] = ssim_sum;
      }
      return;
    }
  }
  float hist[4][11], cx[11], cy[11];                       // rings, slot = row mod 11 (static: the loop is unrolled by 11)
#pragma unroll
  for (int k = 0; k < 11; k++) { hist[0][k] = hist[1][k] = hist[2][k] = hist[3][k] = 0.0f; cx[k] = cy[k] = 0.0f; }
  float l1_sum = 0.0f, ssim_sum = 0.0f;
  RowRegs n0 = fetch(rbeg), n1 = fetch(rbeg + 1);
#pragma unroll 1
  for (int it = 0; it < kStripIters; it++) {
#pragma unroll
    for (int u = 0; u < 11; u++) {
      const int r = rbeg + it * 11 + u;
      const RowRegs cur = n0;
      n0 = n1;
      n1 = fetch(r + 2);
      const bool rin = r >= 0 && r < H;
      {
        const float m = (rin && cin_a) ? cur.ma : 0.0f;
        const float xv = cur.xa * m, yv = cur.ya * m;
        buf4[lane] = make_float4(xv, yv, xv * xv + yv * yv, xv * yv);
        if (lane < 2 * kLH) {
          const float m2 = (rin && cin_b) ? cur.mb : 0.0f;
          const float x2 = cur.xb * m2, y2 = cur.yb * m2;
          buf4[64 + lane] = make_float4(x2, y2, x2 * x2 + y2 * y2, x2 * y2);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float h0 = 0, h1 = 0, h2 = 0, h3 = 0;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = win.w[k];
        const float4 b = buf4[lane + k];
        h0 += w * b.x; h1 += w * b.y; h2 += w * b.z; h3 += w * b.w;
        if (k == kLH) { cx[u] = b.x; cy[u] = b.y; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();                      // the next row's writes come after these reads
      hist[0][u] = h0; hist[1][u] = h1; hist[2][u] = h2; hist[3][u] = h3;
      // output row ro = r - 5: its window is rows ro-5 .. ro+5 = r-10 .. r, ring slots (u + 1 + k) mod 11
      const int ro = r - kLH;
      if (ro >= ry0 && ro < H && ro < ry0 + kStripRows) {   // wave-uniform
        float m1 = 0, m2 = 0, ess = 0, e12 = 0;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          const float w = win.w[k];
          const int sl = (u + 1 + k) % 11;
          m1 += w * hist[0][sl]; m2 += w * hist[1][sl]; ess += w * hist[2][sl]; e12 += w * hist[3][sl];
        }
        if (gx < W) {
          const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
          const float s12 = e12 - m12;
          const float A1 = 2 * m12 + kC1, A2 = 2 * s12 + kC2, B1 = m1s + m2s + kC1, B2 = (ess - m1s - m2s) + kC2;
          // B1 >= C1, B2 >= ~C2 > 0: hardware reciprocals (1 ulp) instead of two IEEE division sequences
          const float rB2 = __builtin_amdgcn_rcpf(B2);
          const float inv = __builtin_amdgcn_rcpf(B1) * rB2;
          const float ssim = A1 * A2 * inv;
          ssim_sum += ssim;
          const int sc = (u + 11 - kLH) % 11;               // the centre row's x, y
          l1_sum += fabsf(cx[sc] - cy[sc]);
          if (dmaps) {
            float *d0 = dmaps + ch * HW + (size_t)ro * W;    // uniform row base + lane offset
            d0[(uint32_t)gx] = (2 * m2 * (A2 - A1) - 2 * m1 * ssim * (B2 - B1)) * inv;  // d ssim / d mu1
            (d0 + CHW)[(uint32_t)gx] = -ssim * rB2;                                      // d ssim / d E[x^2]
            (d0 + 2 * CHW)[(uint32_t)gx] = 2 * A1 * inv;                                  // d ssim / d E[xy]
          }
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { l1_sum += __shfl_xor(l1_sum, off, 64); ssim_sum += __shfl_xor(ssim_sum, off, 64); }
  if (lane == 0) {
    const int wave = blockIdx.x * 4 + wv;
    partials[2 * wave] = l1_sum;
    partials[2 * wave + 1] = ssim_sum;
  }
}

// The loss values from the forward's per-wave partial sums: float2 loads four at a time in flight, then a fixed-order
// tree in double (kThreads threads of one workgroup; s_a / s_b: kThreads doubles each).
template <int kThreads>
__device__ __forceinline__ void finalize_sums(const float *__restrict__ partials, int nblocks, double n_elems, float lambda,
                                              float *__restrict__ out, double *s_a, double *s_b) {
  const float2 *p2 = reinterpret_cast<const float2 *>(partials);
  double a = 0, b = 0;
  for (int i0 = threadIdx.x; i0 < nblocks; i0 += 4 * kThreads) {
    float2 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int i = i0 + kThreads * j; v[j] = i < nblocks ? p2[i] : make_float2(0.0f, 0.0f); }
#pragma unroll
    for (int j = 0; j < 4; j++) { a += v[j].x; b += v[j].y; }
  }
  s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
  __syncthreads();
  for (int off = kThreads / 2; off >= 1; off >>= 1) {
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

// a launch-latency-sized kernel (4.8 us on the stream): the forward-only callers'.  The one-call views let the loss
// backward's extra workgroup do this instead (ssim_l1_backward_kernel, `fin`)
__global__ void __launch_bounds__(1024)
loss_finalize_kernel(const float *__restrict__ partials, int nblocks, double n_elems, float lambda, float *__restrict__ out) {
  __shared__ double s_a[1024], s_b[1024];
  finalize_sums<1024>(partials, nblocks, n_elems, lambda, out, s_a, s_b);
}

struct LossFinalize { const float *partials; int nparts; double n_elems; float *out; };   // out == NULL: nothing to do

// kMasked: a separate instantiation, so that the unmasked one (every training view) keeps its 74 registers and six waves
// per SIMD - with the mask's early exit as a run-time branch the allocator took 84 and the kernel lost a wave.
template <bool kMasked>
__global__ void __launch_bounds__(256)
ssim_l1_backward_kernel(const float *__restrict__ img, const float *__restrict__ gt, const float *__restrict__ mask,
                        const float *__restrict__ dmaps, int C, int H, int W, Window win, float lambda, float inv_n,
                        const float *__restrict__ grad_out, float *__restrict__ grad_img, LossFinalize fin) {
  // The three derivative maps go through the same two LDS buffers one after the other (13 KB per workgroup instead
  // of 39 KB: occupancy, not arithmetic, limits this kernel); each thread accumulates its four output pixels.
  __shared__ __attribute__((aligned(16))) float s_m[kLP][kLPS];
  __shared__ __attribute__((aligned(16))) float s_h[kLP][kLHS];
  if (fin.out && blockIdx.x == gridDim.x - 1) {   // the workgroup appended for the loss values (the partial sums are complete:
                                                  // the forward ran before this launch on the stream)
    static_assert(sizeof(s_m) >= 256 * sizeof(double) && sizeof(s_h) >= 256 * sizeof(double), "the tree's scratch fits the tile buffers");
    finalize_sums<256>(fin.partials, fin.nparts, fin.n_elems, lambda, fin.out, reinterpret_cast<double *>(&s_m[0][0]),
                       reinterpret_cast<double *>(&s_h[0][0]));
    return;
  }
  const LossTile lt = loss_tile((W + kLT - 1) / kLT, (H + kLT - 1) / kLT, C);
  if (!lt.valid) return;
  const int ch = lt.ch, x0 = lt.x0, y0 = lt.y0;
  const size_t HW = (size_t)H * W, CHW = (size_t)C * HW;
  const int c = threadIdx.x % kLT, r0 = (threadIdx.x / kLT) * 4;
  const float go = grad_out ? grad_out[0] : 1.0f;
  float xv[4], yv[4], mk[4], acc[4];
  if constexpr (kMasked) {
    // The mask first (one extra round trip for masked calls only): a tile it zeroes entirely - most tiles when the mask is
    // one object's silhouette - has a zero gradient whatever the maps hold, and leaves before anything else is fetched.
#pragma unroll
    for (int o = 0; o < 4; o++) {
      const int gy = y0 + r0 + o, gx = x0 + c;
      const float m = mask[(size_t)min(gy, H - 1) * W + min(gx, W - 1)];
      mk[o] = (gy < H && gx < W) ? m : 0.0f;
    }
    const bool any = (mk[0] != 0.0f) | (mk[1] != 0.0f) | (mk[2] != 0.0f) | (mk[3] != 0.0f);
    int *s_any = reinterpret_cast<int *>(&s_h[0][0]);   // (the tile buffers are idle; every writer writes the same 1)
    if (threadIdx.x == 0) *s_any = 0;
    __syncthreads();
    if (any) *s_any = 1;
    __syncthreads();
    const bool some = *s_any != 0;
    __syncthreads();                                    // (before the buffer is a tile buffer again)
    if (!some) {   // (workgroup-uniform)
#pragma unroll
      for (int o = 0; o < 4; o++) {
        const int gy = y0 + r0 + o, gx = x0 + c;
        if (gy < H && gx < W) grad_img[ch * HW + (size_t)gy * W + gx] = 0.0f;
      }
      return;
    }
  } else {
#pragma unroll
    for (int o = 0; o < 4; o++) mk[o] = (y0 + r0 + o < H && x0 + c < W) ? 1.0f : 0.0f;
  }
  // the thread's own four pixels: all loads issued together from clamped addresses, out-of-image ones zeroed by the mask
  // (with the bounds test around the loads each pixel was two dependent round trips to memory)
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int gy = y0 + r0 + o, gx = x0 + c;
    const size_t p = (size_t)min(gy, H - 1) * W + min(gx, W - 1);
    xv[o] = img[ch * HW + p]; yv[o] = gt[ch * HW + p];
  }
#pragma unroll
  for (int o = 0; o < 4; o++) {
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
  return align_up((size_t)3 * C * H * W * 4, 256) + align_up((size_t)strip_waves(C, H, W) * 8, 256);
}

namespace scorp {
// finalize = false: the loss values are left to loss_backward_impl's extra workgroup (out_loss3 there) - the one-call
// views, which always run both; nothing else may read out_loss3 in between.
int loss_forward_impl(const float *img, const float *gt, const float *mask, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                      float *out_loss3, void *workspace, size_t workspace_bytes, int32_t need_backward, bool finalize,
                      hipStream_t stream) {
  if (!img || !gt || !out_loss3 || !workspace) { set_error("NULL argument to scorp_loss_l1_ssim_forward"); return SCORP_ERR_INVALID; }
  if (C <= 0 || H <= 0 || W <= 0) { set_error("bad image shape"); return SCORP_ERR_INVALID; }
  if (workspace_bytes < scorp_loss_workspace_bytes(C, H, W) || ((uintptr_t)workspace & 15)) {
    set_error("loss workspace too small or misaligned"); return SCORP_ERR_INVALID;
  }
  float *dmaps = (float *)workspace;
  float *partials = (float *)((char *)workspace + align_up((size_t)3 * C * H * W * 4, 256));
  const Window win = make_window();
  const int nparts = strip_waves(C, H, W);
  {
    ProfScope prof(kKLossForward, stream);
    ssim_l1_forward_strip_kernel<<<(nparts + 3) / 4, 256, 0, stream>>>(img, gt, mask, C, H, W, win, need_backward ? dmaps : nullptr, partials);
  }
  SCORP_KERNEL_CHECK("ssim_l1_forward", 0, stream);
  if (finalize) {
    loss_finalize_kernel<<<1, 1024, 0, stream>>>(partials, nparts, (double)C * H * W, lambda_dssim, out_loss3);
    SCORP_KERNEL_CHECK("loss_finalize", 0, stream);
  }
  return SCORP_OK;
}

// out_loss3 != NULL: also writes the loss values from the forward's partial sums (loss_forward_impl with finalize = false)
int loss_backward_impl(const float *img, const float *gt, const float *mask, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                       const void *workspace, const float *grad_out, float *grad_img, float *out_loss3, hipStream_t stream) {
  if (!img || !gt || !workspace || !grad_img) { set_error("NULL argument to scorp_loss_l1_ssim_backward"); return SCORP_ERR_INVALID; }
  const int grid = (loss_blocks(C, H, W) + 7) / 8 * 8;
  const Window win = make_window();
  LossFinalize fin;
  fin.partials = (const float *)((const char *)workspace + align_up((size_t)3 * C * H * W * 4, 256));
  fin.nparts = strip_waves(C, H, W);
  fin.n_elems = (double)C * H * W;
  fin.out = out_loss3;
  {
    ProfScope prof(kKLossBackward, stream);
    if (mask)
      ssim_l1_backward_kernel<true><<<grid + (out_loss3 ? 1 : 0), 256, 0, stream>>>(img, gt, mask, (const float *)workspace, C, H, W, win,
                                                                           lambda_dssim, (float)(1.0 / ((double)C * H * W)), grad_out,
                                                                           grad_img, fin);
    else
      ssim_l1_backward_kernel<false><<<grid + (out_loss3 ? 1 : 0), 256, 0, stream>>>(img, gt, mask, (const float *)workspace, C, H, W, win,
                                                                           lambda_dssim, (float)(1.0 / ((double)C * H * W)), grad_out,
                                                                           grad_img, fin);
  }
  SCORP_KERNEL_CHECK("ssim_l1_backward", 0, stream);
  return SCORP_OK;
}
}  // namespace scorp

extern "C" int scorp_loss_l1_ssim_forward(const float *img, const float *gt, const float *mask, int32_t C, int32_t H,
                                          int32_t W, float lambda_dssim, float *out_loss3, void *workspace,
                                          size_t workspace_bytes, int32_t need_backward, scorp_stream_t stream_) {
  return loss_forward_impl(img, gt, mask, C, H, W, lambda_dssim, out_loss3, workspace, workspace_bytes, need_backward, true,
                           (hipStream_t)stream_);
}

extern "C" int scorp_loss_l1_ssim_backward(const float *img, const float *gt, const float *mask, int32_t C, int32_t H,
                                           int32_t W, float lambda_dssim, const void *workspace, const float *grad_out,
                                           float *grad_img, scorp_stream_t stream_) {
  return loss_backward_impl(img, gt, mask, C, H, W, lambda_dssim, workspace, grad_out, grad_img, nullptr, (hipStream_t)stream_);
}
