// pergaussian.hpp — device helpers shared by the per-Gaussian kernels of the 3DGS and 2DGS paths:
// SH constants and evaluation from an LDS-staged row, coalesced staging of SH rows through LDS (see
// gs3d_pergaussian.hip for the rationale), and the activations of the "raw parameter" convention.
#pragma once
#include "common.hpp"

namespace scorp {
namespace {

constexpr int kShStride = 49;
constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
static __device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
static __device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

// ---- coalesced staging of the SH rows of one block (rows i0 .. i0+nrows) into LDS, columns [0, NFL) ----
template <int NFL, bool SPLIT>
__device__ __forceinline__ void stage_sh_rows(float *__restrict__ lds, const float *__restrict__ shs,
                                              const float *__restrict__ shs_rest, int K, size_t i0, int nrows) {
  const int tid = threadIdx.x;
  if constexpr (!SPLIT) {
    const int K3 = K * 3;
    if constexpr (NFL % 4 == 0) {
      if ((K & 3) == 0) {
        constexpr int Q = NFL / 4;
        for (int e = tid; e < nrows * Q; e += 256) {
          const int r = e / Q, q = e % Q;
          const float4 v = *reinterpret_cast<const float4 *>(shs + (i0 + r) * K3 + 4 * q);
          float *d = lds + r * kShStride + 4 * q;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        return;
      }
    }
    for (int e = tid; e < nrows * NFL; e += 256) {
      const int r = e / NFL, c = e % NFL;
      lds[r * kShStride + c] = shs[(i0 + r) * K3 + c];
    }
  } else {
    for (int e = tid; e < nrows * 3; e += 256) {
      const int r = e / 3, c = e % 3;
      lds[r * kShStride + c] = shs[(i0 + r) * 3 + c];
    }
    if constexpr (NFL > 3) {
      constexpr int NR = NFL - 3;
      const int R3 = (K - 1) * 3;
      if (NR == 45 && R3 == 45 && nrows == 256) {  // whole rows of a full block: one contiguous 16-byte-aligned stream
        const float4 *src = reinterpret_cast<const float4 *>(shs_rest + i0 * 45);
        for (int e4 = tid; e4 < 256 * 45 / 4; e4 += 256) {
          const float4 v = src[e4];
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int f = 4 * e4 + j;
            lds[(f / 45) * kShStride + 3 + f % 45] = vv[j];
          }
        }
      } else {
        for (int e = tid; e < nrows * NR; e += 256) {
          const int r = e / NR, c = e % NR;
          lds[r * kShStride + 3 + c] = shs_rest[(i0 + r) * R3 + c];
        }
      }
    }
  }
}

// ---- the reverse: stream the (gradient) rows back out; columns >= 48 of very wide rows are zero-filled ----
template <bool SPLIT>
__device__ __forceinline__ void unstage_sh_rows(const float *__restrict__ lds, float *__restrict__ g_shs,
                                                float *__restrict__ g_rest, int K, size_t i0, int nrows) {
  const int tid = threadIdx.x;
  if constexpr (!SPLIT) {
    const int K3 = K * 3;
    if (K == 16) {
      for (int e = tid; e < nrows * 12; e += 256) {
        const int r = e / 12, q = e % 12;
        const float *s = lds + r * kShStride + 4 * q;
        *reinterpret_cast<float4 *>(g_shs + (i0 + r) * 48 + 4 * q) = make_float4(s[0], s[1], s[2], s[3]);
      }
    } else {
      for (int e = tid; e < nrows * K3; e += 256) {
        const int r = e / K3, c = e % K3;
        g_shs[(i0 + r) * K3 + c] = c < 48 ? lds[r * kShStride + c] : 0.0f;
      }
    }
  } else {
    for (int e = tid; e < nrows * 3; e += 256) {
      const int r = e / 3, c = e % 3;
      g_shs[(i0 + r) * 3 + c] = lds[r * kShStride + c];
    }
    const int R3 = (K - 1) * 3;
    if (R3 == 45 && nrows == 256) {
      float4 *dst = reinterpret_cast<float4 *>(g_rest + i0 * 45);
      for (int e4 = tid; e4 < 256 * 45 / 4; e4 += 256) {
        float vv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int f = 4 * e4 + j;
          vv[j] = lds[(f / 45) * kShStride + 3 + f % 45];
        }
        dst[e4] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      }
    } else if (R3 > 0) {
      for (int e = tid; e < nrows * R3; e += 256) {
        const int r = e / R3, c = e % R3;
        g_rest[(i0 + r) * R3 + c] = c < 45 ? lds[r * kShStride + 3 + c] : 0.0f;
      }
    }
  }
}

// A staged SH row as two bases so that the same code serves both LDS layouts:
//   padded  : one row of 48 floats at stride 49                      -> p0 = pr = row
//   linear  : dc rows (3 floats) and rest rows (45 floats) staged as plain copies of the global arrays; both strides are
//             odd, so one-thread-per-row walks stay bank-conflict free  -> p0 = dc row, pr = rest row - 3
struct ShRow { float *p0; float *pr; };
constexpr int kShLinearRest = 256 * 3;  // float offset of the rest rows in the linear layout
__device__ __forceinline__ ShRow sh_row(float *lds, int t, bool linear) {
  ShRow r;
  if (linear) { r.p0 = lds + 3 * t; r.pr = lds + kShLinearRest + 45 * t - 3; }
  else { r.p0 = lds + t * kShStride; r.pr = r.p0; }
  return r;
}
__device__ __forceinline__ void sh_row_zero(ShRow r) {
#pragma unroll
  for (int c = 0; c < 3; c++) r.p0[c] = 0.0f;
#pragma unroll
  for (int c = 3; c < 48; c++) r.pr[c] = 0.0f;
}
// Linear staging of a FULL block (256 rows) of the split layout with K = 16: two plain 16-byte streams, no index math,
// issued as direct global -> LDS loads (global_load_lds_dwordx4: each lane's 16 bytes land at
// M0 base + lane * 16, no VGPR round trip).  Fire-and-forget: the caller does its other work, then
// stage_sh_wait() + a workgroup barrier before anyone reads the rows.  48 wave-chunks of 1 KiB, 12 per wave.
#ifndef SCORP_NT_SH
#define SCORP_NT_SH 1   // nontemporal SH staging where the rows have no second reader nearby (0: A/B builds)
#endif
// AUX: the loads' cache-policy bits (0: default; 2: nontemporal - rows that nobody reads again before > 1 GB of other traffic)
template <int AUX = 0>
__device__ __forceinline__ void stage_sh_linear_async(float *__restrict__ lds, const float *__restrict__ dc,
                                                      const float *__restrict__ rest, size_t i0) {
  typedef const __attribute__((address_space(1))) void *GPtr;
  typedef __attribute__((address_space(3))) void *LPtr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float4 *d4 = reinterpret_cast<const float4 *>(dc + i0 * 3), *r4 = reinterpret_cast<const float4 *>(rest + i0 * 45);
  float4 *l4 = reinterpret_cast<float4 *>(lds), *lr4 = reinterpret_cast<float4 *>(lds + kShLinearRest);
#pragma unroll
  for (int q = 0; q < 12; q++) {
    const int c = wave + 4 * q;   // wave-uniform chunk id: 0..2 dc, 3..47 rest
    if (c < 3) __builtin_amdgcn_global_load_lds((GPtr)(d4 + c * 64 + lane), (LPtr)(l4 + c * 64), 16, 0, AUX);
    else __builtin_amdgcn_global_load_lds((GPtr)(r4 + (c - 3) * 64 + lane), (LPtr)(lr4 + (c - 3) * 64), 16, 0, AUX);
  }
}
__device__ __forceinline__ void stage_sh_wait() { __builtin_amdgcn_s_waitcnt(0); }

// Loads the compiler does not track.  Its wait-count pass treats ordinary loads and LDS-DMA loads sharing vmcnt as
// possibly out of order and puts vmcnt(0) in front of the first use of ANY loaded value, i.e. behind the whole SH stream.
// Loads do retire in order among themselves, so the per-Gaussian parameters (means 3, rotation 4, scale 3, opacity 1
// floats) fetched with raw_issue_params BEFORE stage_sh_linear_async are complete at vmcnt(12) (12 LDS loads per wave
// follow them).  Between issue and raw_take_params NOTHING may touch the destination registers: the compiler believes
// asm outputs ready at once and would happily copy them, so they are used exactly once, as plain inputs of the asm
// that waits and then moves them into fresh registers.  (No stores may be pending either: stores can overtake loads.)
struct RawParams { float v[11]; };
// NS = floats in the third group (3: 3DGS scales / accumulators 7-9; 2: surfel scales, slot 9 then repeats slot 8 so that
// nothing past the end of the array is read and the count of loads stays 11)
template <int NS = 3>
__device__ __forceinline__ void raw_issue_params(RawParams &r, const float *means, const float *rot, const float *scl,
                                                 const float *opa) {
  asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %3, off offset:4\n\tglobal_load_dword %2, %3, off offset:8"
               : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]) : "v"(means) : "memory");
  asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:4\n\tglobal_load_dword %2, %4, off offset:8\n\t"
               "global_load_dword %3, %4, off offset:12"
               : "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]) : "v"(rot) : "memory");
  if constexpr (NS == 3)
    asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %3, off offset:4\n\tglobal_load_dword %2, %3, off offset:8"
                 : "=&v"(r.v[7]), "=&v"(r.v[8]), "=&v"(r.v[9]) : "v"(scl) : "memory");
  else
    asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %3, off offset:4\n\tglobal_load_dword %2, %3, off offset:4"
                 : "=&v"(r.v[7]), "=&v"(r.v[8]), "=&v"(r.v[9]) : "v"(scl) : "memory");
  asm volatile("global_load_dword %0, %1, off" : "=&v"(r.v[10]) : "v"(opa) : "memory");
}
__device__ __forceinline__ void raw_take_params(const RawParams &r, float *p) {
  asm volatile("s_waitcnt vmcnt(12)\n\t"
               "v_mov_b32 %0, %11\n\tv_mov_b32 %1, %12\n\tv_mov_b32 %2, %13\n\tv_mov_b32 %3, %14\n\t"
               "v_mov_b32 %4, %15\n\tv_mov_b32 %5, %16\n\tv_mov_b32 %6, %17\n\tv_mov_b32 %7, %18\n\t"
               "v_mov_b32 %8, %19\n\tv_mov_b32 %9, %20\n\tv_mov_b32 %10, %21"
               : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]),
                 "=&v"(p[8]), "=&v"(p[9]), "=&v"(p[10])
               : "v"(r.v[0]), "v"(r.v[1]), "v"(r.v[2]), "v"(r.v[3]), "v"(r.v[4]), "v"(r.v[5]), "v"(r.v[6]), "v"(r.v[7]),
                 "v"(r.v[8]), "v"(r.v[9]), "v"(r.v[10])
               : "memory");
}

__device__ __forceinline__ void unstage_sh_linear(const float *__restrict__ lds, float *__restrict__ g_dc,
                                                  float *__restrict__ g_rest, size_t i0) {
  float4 *d4 = reinterpret_cast<float4 *>(g_dc + i0 * 3), *r4 = reinterpret_cast<float4 *>(g_rest + i0 * 45);
  const float4 *l4 = reinterpret_cast<const float4 *>(lds), *lr4 = reinterpret_cast<const float4 *>(lds + kShLinearRest);
  // (plain stores: nontemporal ones measured 95.7 against 94 us for the kernel)
  for (int e = threadIdx.x; e < 256 * 3 / 4; e += 256) d4[e] = l4[e];
  for (int e = threadIdx.x; e < 256 * 45 / 4; e += 256) r4[e] = lr4[e];
}

// SH -> RGB + 0.5 from a staged row (gs3dgs/utils/sh_utils.py:57-112 restated for [K,3] rows)
template <int DEG>
__device__ __forceinline__ void sh_row_to_rgb(ShRow row, float x, float y, float z, float *rgb) {
  const float *sh0 = row.p0, *sh = row.pr;  // sh0[c]: degree-0 coefficient; sh[3k + c]: coefficient k >= 1
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float r = SH_C0 * sh0[c];
    if constexpr (DEG > 0) {
      r = r - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
      if constexpr (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        r = r + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] + SH_C2[2] * (2 * zz - xx - yy) * sh[18 + c] +
            SH_C2[3] * xz * sh[21 + c] + SH_C2[4] * (xx - yy) * sh[24 + c];
        if constexpr (DEG > 2) {
          r = r + SH_C3[0] * y * (3 * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
              SH_C3[2] * y * (4 * zz - xx - yy) * sh[33 + c] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[36 + c] +
              SH_C3[4] * x * (4 * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
              SH_C3[6] * x * (xx - 3 * yy) * sh[45 + c];
        }
      }
    }
    rgb[c] = r + 0.5f;
  }
}


// Backward of sh_row_to_rgb for one staged row: gr[ch] = dL/d(colour ch) after the clamp mask.  Accumulates
// dL/d(unit direction) into gdir and, if write_grad, overwrites the row IN PLACE with dL/d(coefficients)
// (each channel's coefficients are consumed before they are overwritten).
template <int DEG>
__device__ __forceinline__ void sh_row_backward(ShRow srow, float x, float y, float z, const float *gr3,
                                                bool write_grad, float *gdir) {
        float *row = srow.pr;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        float basis[16];
        basis[0] = SH_C0;
        if constexpr (DEG > 0) { basis[1] = -SH_C1 * y; basis[2] = SH_C1 * z; basis[3] = -SH_C1 * x; }
        if constexpr (DEG > 1) {
          basis[4] = SH_C2[0] * xy; basis[5] = SH_C2[1] * yz; basis[6] = SH_C2[2] * (2 * zz - xx - yy);
          basis[7] = SH_C2[3] * xz; basis[8] = SH_C2[4] * (xx - yy);
        }
        if constexpr (DEG > 2) {
          basis[9] = SH_C3[0] * y * (3 * xx - yy); basis[10] = SH_C3[1] * xy * z;
          basis[11] = SH_C3[2] * y * (4 * zz - xx - yy); basis[12] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
          basis[13] = SH_C3[4] * x * (4 * zz - xx - yy); basis[14] = SH_C3[5] * z * (xx - yy);
          basis[15] = SH_C3[6] * x * (xx - 3 * yy);
        }
  #pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          const float gr = gr3[ch];
          float rx = 0, ry = 0, rz = 0;
          if constexpr (DEG > 0) {
            const float s1_ = row[3 + ch], s2_ = row[6 + ch], s3_ = row[9 + ch];
            rx = -SH_C1 * s3_; ry = -SH_C1 * s1_; rz = SH_C1 * s2_;
          }
          if constexpr (DEG > 1) {
            const float s4 = row[12 + ch], s5 = row[15 + ch], s6 = row[18 + ch], s7 = row[21 + ch], s8 = row[24 + ch];
            rx += SH_C2[0] * y * s4 + SH_C2[2] * 2 * -x * s6 + SH_C2[3] * z * s7 + SH_C2[4] * 2 * x * s8;
            ry += SH_C2[0] * x * s4 + SH_C2[1] * z * s5 + SH_C2[2] * 2 * -y * s6 + SH_C2[4] * 2 * -y * s8;
            rz += SH_C2[1] * y * s5 + SH_C2[2] * 4 * z * s6 + SH_C2[3] * x * s7;
          }
          if constexpr (DEG > 2) {
            const float s9 = row[27 + ch], s10 = row[30 + ch], s11 = row[33 + ch], s12 = row[36 + ch], s13 = row[39 + ch],
                        s14 = row[42 + ch], s15 = row[45 + ch];
            rx += SH_C3[0] * s9 * 6 * xy + SH_C3[1] * s10 * yz + SH_C3[2] * s11 * -2 * xy + SH_C3[3] * s12 * -6 * xz +
                  SH_C3[4] * s13 * (-3 * xx + 4 * zz - yy) + SH_C3[5] * s14 * 2 * xz + SH_C3[6] * s15 * 3 * (xx - yy);
            ry += SH_C3[0] * s9 * 3 * (xx - yy) + SH_C3[1] * s10 * xz + SH_C3[2] * s11 * (-3 * yy + 4 * zz - xx) +
                  SH_C3[3] * s12 * -6 * yz + SH_C3[4] * s13 * -2 * xy + SH_C3[5] * s14 * -2 * yz + SH_C3[6] * s15 * -6 * xy;
            rz += SH_C3[1] * s10 * xy + SH_C3[2] * s11 * 8 * yz + SH_C3[3] * s12 * 3 * (2 * zz - xx - yy) +
                  SH_C3[4] * s13 * 8 * xz + SH_C3[5] * s14 * (xx - yy);
          }
          gdir[0] += rx * gr; gdir[1] += ry * gr; gdir[2] += rz * gr;
          // the coefficients of this channel are consumed: overwrite them in place with their gradients
          if (write_grad) {
  #pragma unroll
            for (int k = 1; k < 16; k++) row[3 * k + ch] = k < (DEG + 1) * (DEG + 1) ? basis[k] * gr : 0.0f;
            srow.p0[ch] = basis[0] * gr;
          }
        }
}

// ---- the optimizer step inside the per-Gaussian backward kernels (ScorpFusedAdam; 3-D and 2-D) ----
// n consecutive floats of one leaf: Adam with the gradient in registers.  `p_in`: the parameter values the thread already
// holds (NULL: read them).
template <int n>
__device__ __forceinline__ void adam_leaf(const AdamEpi &ad, int k, float *__restrict__ param, size_t at, const float *gr,
                                          const float *p_in) {
  if (!ad.m[k]) return;
  float *pp = param + at, *pm_ = ad.m[k] + at, *pv = ad.v[k] + at;
  float pvals[n], m[n], v[n];
#pragma unroll
  for (int q = 0; q < n; q++) { pvals[q] = p_in ? p_in[q] : pp[q]; m[q] = adam_ld1(pm_ + q); v[q] = adam_ld1(pv + q); }
#pragma unroll
  for (int q = 0; q < n; q++) adam_one(pvals[q], gr[q], m[q], v[q], ad.omb1, ad.beta2, ad.omb2, ad.step_size[k], ad.inv_sqrt_bc2, ad.eps);
#pragma unroll
  for (int q = 0; q < n; q++) { adam_st1(pp + q, pvals[q]); adam_st1(pm_ + q, m[q]); adam_st1(pv + q, v[q]); }
}
// The same with the two moments asked for EARLY (before the SH phase of the kernel, which hides their latency; left to
// adam_leaf the four leaves are four dependent rounds of load - compute - store behind one another, because the compiler
// may not move a leaf's loads above the previous leaf's stores).  Slots: means 0-2, opacity 3, scales 4-6, rotation 7-10.
struct AdamGeomMoments { float m[11], v[11]; };
template <int n>
__device__ __forceinline__ void adam_moments_load(const AdamEpi &ad, int k, size_t at, float *m, float *v) {
  if (!ad.m[k]) return;
#pragma unroll
  for (int q = 0; q < n; q++) { m[q] = adam_ld1(ad.m[k] + at + q); v[q] = adam_ld1(ad.v[k] + at + q); }
}
template <int n>
__device__ __forceinline__ void adam_leaf_pre(const AdamEpi &ad, int k, float *__restrict__ param, size_t at, const float *gr,
                                              const float *p_in, float *m, float *v) {
  if (!ad.m[k]) return;
  float *pp = param + at, *pm_ = ad.m[k] + at, *pv = ad.v[k] + at;
  float pvals[n];
#pragma unroll
  for (int q = 0; q < n; q++) pvals[q] = p_in ? p_in[q] : pp[q];
#pragma unroll
  for (int q = 0; q < n; q++) adam_one(pvals[q], gr[q], m[q], v[q], ad.omb1, ad.beta2, ad.omb2, ad.step_size[k], ad.inv_sqrt_bc2, ad.eps);
#pragma unroll
  for (int q = 0; q < n; q++) { adam_st1(pp + q, pvals[q]); adam_st1(pm_ + q, m[q]); adam_st1(pv + q, v[q]); }
}
__device__ __forceinline__ void adam_one4(const AdamEpi &ad, int k, float4 &p, const float4 g, float4 &m, float4 &v) {
  adam_one(p.x, g.x, m.x, v.x, ad.omb1, ad.beta2, ad.omb2, ad.step_size[k], ad.inv_sqrt_bc2, ad.eps);
  adam_one(p.y, g.y, m.y, v.y, ad.omb1, ad.beta2, ad.omb2, ad.step_size[k], ad.inv_sqrt_bc2, ad.eps);
  adam_one(p.z, g.z, m.z, v.z, ad.omb1, ad.beta2, ad.omb2, ad.step_size[k], ad.inv_sqrt_bc2, ad.eps);
  adam_one(p.w, g.w, m.w, v.w, ad.omb1, ad.beta2, ad.omb2, ad.step_size[k], ad.inv_sqrt_bc2, ad.eps);
}
// The SH leaves of a block, gradient rows in LDS.  Linear layout (a full block of the training layout): two 16-byte streams
// per array; the parameters are read again from global memory - the block streamed those 46 KB into LDS microseconds ago
// (L2) and has since overwritten them there with their gradients.
// The streams are software-pipelined by hand: a thread holds FOUR (p, m, v) triples, the two it is computing on and the two
// it has asked for next.  Left as a plain loop the compiler keeps one triple per thread - load, ~180 instructions of IEEE
// sqrt and division, store, and only then the next load: with twelve waves per CU that is ~27 KB in flight per CU, 5.3 TB/s
// for the kernel; see profiles/r06_fused_adam_step.txt for what the deeper pipeline buys.
template <int N4>   // float4 elements of the block's slice of one leaf
__device__ __forceinline__ void adam_stream4(const AdamEpi &ad, int k, const float4 *__restrict__ g4, float4 *__restrict__ P,
                                             float4 *__restrict__ M, float4 *__restrict__ V) {
  static_assert(N4 >= 256, "every thread has a first element");
  int e = threadIdx.x;
  float4 p0 = adam_ld4(P + e), m0 = adam_ld4(M + e), v0 = adam_ld4(V + e), p1, m1, v1;
  if (e + 256 < N4) { p1 = adam_ld4(P + e + 256); m1 = adam_ld4(M + e + 256); v1 = adam_ld4(V + e + 256); }
  for (; e < N4; e += 512) {
    float4 p2, m2, v2, p3, m3, v3;
    if (e + 512 < N4) { p2 = adam_ld4(P + e + 512); m2 = adam_ld4(M + e + 512); v2 = adam_ld4(V + e + 512); }
    if (e + 768 < N4) { p3 = adam_ld4(P + e + 768); m3 = adam_ld4(M + e + 768); v3 = adam_ld4(V + e + 768); }
    adam_one4(ad, k, p0, g4[e], m0, v0);
    adam_st4(P + e, p0); adam_st4(M + e, m0); adam_st4(V + e, v0);
    if (e + 256 < N4) {
      adam_one4(ad, k, p1, g4[e + 256], m1, v1);
      adam_st4(P + e + 256, p1); adam_st4(M + e + 256, m1); adam_st4(V + e + 256, v1);
    }
    p0 = p2; m0 = m2; v0 = v2; p1 = p3; m1 = m3; v1 = v3;
  }
}
__device__ __forceinline__ void adam_sh_linear(const AdamEpi &ad, const float *__restrict__ lds, float *__restrict__ dc,
                                               float *__restrict__ rest, size_t i0) {
  const float4 *l4 = reinterpret_cast<const float4 *>(lds), *lr4 = reinterpret_cast<const float4 *>(lds + kShLinearRest);
  if (ad.m[1]) {
    float4 *P = reinterpret_cast<float4 *>(dc + i0 * 3), *M = reinterpret_cast<float4 *>(ad.m[1] + i0 * 3), *V = reinterpret_cast<float4 *>(ad.v[1] + i0 * 3);
    for (int e = threadIdx.x; e < 256 * 3 / 4; e += 256) {
      float4 p = adam_ld4(P + e), m = adam_ld4(M + e), v = adam_ld4(V + e);
      adam_one4(ad, 1, p, l4[e], m, v);
      adam_st4(P + e, p); adam_st4(M + e, m); adam_st4(V + e, v);
    }
  }
  if (ad.m[2])
    adam_stream4<256 * 45 / 4>(ad, 2, lr4, reinterpret_cast<float4 *>(rest + i0 * 45), reinterpret_cast<float4 *>(ad.m[2] + i0 * 45),
                               reinterpret_cast<float4 *>(ad.v[2] + i0 * 45));
}
// ... and the padded layout (the last, partial block; any K): element by element
__device__ __forceinline__ void adam_sh_rows(const AdamEpi &ad, const float *__restrict__ lds, float *__restrict__ dc,
                                             float *__restrict__ rest, int K, size_t i0, int nrows) {
  if (ad.m[1])
    for (int e = threadIdx.x; e < nrows * 3; e += 256) {
      const size_t at = i0 * 3 + e;
      float p = dc[at], m = ad.m[1][at], v = ad.v[1][at];
      adam_one(p, lds[(e / 3) * kShStride + e % 3], m, v, ad.omb1, ad.beta2, ad.omb2, ad.step_size[1], ad.inv_sqrt_bc2, ad.eps);
      dc[at] = p; ad.m[1][at] = m; ad.v[1][at] = v;
    }
  const int R3 = (K - 1) * 3;
  if (ad.m[2] && R3 > 0)
    for (int e = threadIdx.x; e < nrows * R3; e += 256) {
      const int r = e / R3, c = e % R3;
      const size_t at = i0 * R3 + e;
      float p = rest[at], m = ad.m[2][at], v = ad.v[2][at];
      adam_one(p, c < 45 ? lds[r * kShStride + 3 + c] : 0.0f, m, v, ad.omb1, ad.beta2, ad.omb2, ad.step_size[2], ad.inv_sqrt_bc2, ad.eps);
      rest[at] = p; ad.m[2][at] = m; ad.v[2][at] = v;
    }
}

// activations of the raw-parameter convention (gaussian_model.py:37-45: sigmoid, exp, F.normalize)
__device__ __forceinline__ float act_opacity(float v, int raw) { return (raw & 1) ? 1.0f / (1.0f + expf(-v)) : v; }
__device__ __forceinline__ float act_scale(float v, int raw) { return (raw & 2) ? expf(v) : v; }
__device__ __forceinline__ float4 act_quat(float4 q, int raw, float *inv_norm) {
  if (!(raw & 4)) { *inv_norm = 1.0f; return q; }
  const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  const float inv = 1.0f / fmaxf(n, 1e-12f);
  *inv_norm = inv;
  return make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
}

}  // namespace
}  // namespace scorp
