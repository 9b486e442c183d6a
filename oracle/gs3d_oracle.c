/*
 * gs3d_oracle.c — CPU restatement of the 3D-Gaussian-splat tile rasterizer (forward + backward).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (scorp_amd/, the shim packages, the C-ABI
 * library) may import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in `diff_gaussian_rasterization`
 * (ashawkey fork, /root/reference/.gitmodules:15-17), an un-vendored, un-pinned CUDA submodule that is
 * absent from /root/reference and has no tests or golden vectors.  This file therefore restates the
 * published algorithm (Kerbl et al. 2023, "3D Gaussian Splatting", + per-pixel depth/alpha outputs the
 * caller consumes) and anchors on the reference's own call site and helper formulas:
 *   - argument set / output tuple (color[3,H,W], radii[N], depth[1,H,W], alpha[1,H,W]); depth is the
 *     un-normalised sum (the caller divides by alpha):  gs3dgs/gaussian_renderer/__init__.py:101-114
 *   - covariance from scale+quaternion, 6-tuple order xx,xy,xz,yy,yz,zz:
 *     gs3dgs/utils/general_utils.py:79-125, gs3dgs/scene/gaussian_model.py:31-35
 *   - SH basis, direction = normalise(xyz - campos), +0.5, clamp >= 0:
 *     gs3dgs/utils/sh_utils.py:26-112, gs3dgs/gaussian_renderer/__init__.py:88-93
 *   - matrices are handed over transposed (row-vector convention), znear .01 / zfar 100:
 *     gs3dgs/scene/cameras.py:76-97, gs3dgs/utils/graphics_utils.py:38-71
 *   - means2D carries the screen-space positional gradient: gs3dgs/scene/gaussian_model.py:603-605
 * Every constant of the published algorithm that the reference tree cannot confirm is a named macro below.
 *
 * Build: see oracle/Makefile (two flavours: REAL=float -> libgs_oracle_f32.so, REAL=double -> _f64.so).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

/* ---- named constants of the published algorithm (all marked "not verifiable") ---- */
#define GS_TILE 16                 /* tile edge in pixels */
#define GS_NEAR_Z ((REAL)0.2)      /* view-space z cull threshold */
#define GS_DILATION ((REAL)0.3)    /* px^2 added to the 2-D covariance diagonal */
#define GS_FOV_GUARD ((REAL)1.3)   /* clamp of t.x/t.z, t.y/t.z in the EWA Jacobian */
#define GS_RADIUS_SIGMA ((REAL)3.0)/* extent = ceil(3 sigma_max) */
#define GS_LAMBDA_FLOOR ((REAL)0.1)/* floor under the eigenvalue discriminant */
#define GS_ALPHA_MAX ((REAL)0.99)
#define GS_ALPHA_MIN ((REAL)(1.0 / 255.0))
#define GS_T_MIN ((REAL)0.0001)
#define GS_W_EPS ((REAL)0.0000001) /* added to clip w before the perspective divide */
#define GS_DET2_EPS ((REAL)0.0000001) /* added to det^2 in the conic backward */

static const REAL SH_C0 = (REAL)0.28209479177387814;
static const REAL SH_C1 = (REAL)0.4886025119029199;
static const REAL SH_C2[5] = {(REAL)1.0925484305920792, (REAL)-1.0925484305920792, (REAL)0.31539156525252005,
                              (REAL)-1.0925484305920792, (REAL)0.5462742152960396};
static const REAL SH_C3[7] = {(REAL)-0.5900435899266435, (REAL)2.890611442640554, (REAL)-0.4570457994644658,
                              (REAL)0.3731763325901154, (REAL)-0.4570457994644658, (REAL)1.445305721320277,
                              (REAL)-0.5900435899266435};

static inline REAL rexp(REAL x) { return sizeof(REAL) == 4 ? (REAL)expf((float)x) : (REAL)exp((double)x); }
static inline REAL rsqrt_(REAL x) { return sizeof(REAL) == 4 ? (REAL)sqrtf((float)x) : (REAL)sqrt((double)x); }
static inline REAL rceil(REAL x) { return sizeof(REAL) == 4 ? (REAL)ceilf((float)x) : (REAL)ceil((double)x); }
static inline REAL rfma(REAL a, REAL b, REAL c) { return sizeof(REAL) == 4 ? (REAL)fmaf((float)a, (float)b, (float)c) : (REAL)fma((double)a, (double)b, (double)c); }
static inline REAL rmax(REAL a, REAL b) { return a > b ? a : b; }
static inline REAL rmin(REAL a, REAL b) { return a < b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

typedef struct {
  /* problem */
  int N, K, deg, W, H, tiles_x, tiles_y, has_sh, has_cov_precomp;
  REAL tanfovx, tanfovy, scale_mod;
  REAL bg[3], view[16], proj[16], campos[3];
  /* inputs (borrowed pointers, must outlive the state) */
  const REAL *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
  /* per-Gaussian forward state */
  int *radii;          /* N */
  int *rect;           /* N*4: x0,y0,x1,y1 (tile units, x1/y1 exclusive) */
  REAL *xy;            /* N*2 pixel-space centre */
  REAL *depth;         /* N view-space z */
  REAL *conic_o;       /* N*4: A,B,C,opacity */
  REAL *rgb;           /* N*3 */
  REAL *cov3D;         /* N*6 */
  uint8_t *clamped;    /* N*3 */
  /* per-tile sorted lists */
  int64_t *tile_start; /* tiles+1 */
  int *point_list;     /* D */
  int64_t D;
  /* per-pixel */
  REAL *final_T;       /* H*W */
  int *n_contrib;      /* H*W */
} GsState;

/* V[r][c] of the maths matrix lives at m[c*4+r] (the reference hands over transposed matrices). */
static inline void xform4x3(const REAL *m, const REAL *p, REAL *o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  /* depth is the sort key: pin its rounding with explicit fused multiply-adds (same chain on the GPU) */
  o[2] = rfma(m[10], p[2], rfma(m[6], p[1], rfma(m[2], p[0], m[14])));
}
static inline void xform4x4(const REAL *m, const REAL *p, REAL *o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* R(q) for q = (w,x,y,z), used as given (the caller normalises: gaussian_model.py:131-132). */
static void quat_to_R(const REAL *q, REAL R[9]) {
  REAL r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z);     R[2] = 2 * (x * z + r * y);
  R[3] = 2 * (x * y + r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
  R[6] = 2 * (x * z - r * y);     R[7] = 2 * (y * z + r * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* Sigma = (R S)(R S)^T, S = diag(mod * s); out = xx,xy,xz,yy,yz,zz */
static void cov3d_from_scale_rot(const REAL *s, REAL mod, const REAL *q, REAL *c6) {
  REAL R[9], L[9];
  quat_to_R(q, R);
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 3; k++) L[r * 3 + k] = R[r * 3 + k] * (mod * s[k]);
  c6[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
  c6[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
  c6[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
  c6[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
  c6[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
  c6[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
}

/* EWA projection pieces shared by forward and backward. M = J * Wrot (2x3). */
typedef struct { REAL t[3], txc, tyc; int clamp_x, clamp_y; REAL fx, fy, M[6]; } Ewa;

static void ewa_setup(const GsState *S, const REAL *p, Ewa *e) {
  const REAL *vm = S->view;
  xform4x3(vm, p, e->t);
  REAL limx = GS_FOV_GUARD * S->tanfovx, limy = GS_FOV_GUARD * S->tanfovy;
  REAL txtz = e->t[0] / e->t[2], tytz = e->t[1] / e->t[2];
  e->clamp_x = (txtz < -limx) || (txtz > limx);
  e->clamp_y = (tytz < -limy) || (tytz > limy);
  e->txc = rmin(limx, rmax(-limx, txtz)) * e->t[2];
  e->tyc = rmin(limy, rmax(-limy, tytz)) * e->t[2];
  e->fx = (REAL)S->W / (2 * S->tanfovx);
  e->fy = (REAL)S->H / (2 * S->tanfovy);
  REAL tz = e->t[2];
  REAL J00 = e->fx / tz, J02 = -(e->fx * e->txc) / (tz * tz);
  REAL J11 = e->fy / tz, J12 = -(e->fy * e->tyc) / (tz * tz);
  /* Wrot[r][c] = vm[c*4+r] */
  for (int c = 0; c < 3; c++) {
    e->M[c] = J00 * vm[c * 4 + 0] + J02 * vm[c * 4 + 2];
    e->M[3 + c] = J11 * vm[c * 4 + 1] + J12 * vm[c * 4 + 2];
  }
}

static void sym6_mul(const REAL *c6, const REAL *v, REAL *o) { /* o = Sigma v */
  o[0] = c6[0] * v[0] + c6[1] * v[1] + c6[2] * v[2];
  o[1] = c6[1] * v[0] + c6[3] * v[1] + c6[4] * v[2];
  o[2] = c6[2] * v[0] + c6[4] * v[1] + c6[5] * v[2];
}

static void eval_sh_rgb(const GsState *S, int i, REAL *rgb, uint8_t *clamped) {
  const REAL *p = S->means3D + 3 * (size_t)i;
  const REAL *sh = S->shs + (size_t)i * S->K * 3; /* [K][3] */
  REAL d[3] = {p[0] - S->campos[0], p[1] - S->campos[1], p[2] - S->campos[2]};
  REAL inv = 1 / rsqrt_(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  REAL x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
  for (int c = 0; c < 3; c++) {
#define SHK(k) sh[(k) * 3 + c]
    REAL r = SH_C0 * SHK(0);
    if (S->deg > 0) {
      r = r - SH_C1 * y * SHK(1) + SH_C1 * z * SHK(2) - SH_C1 * x * SHK(3);
      if (S->deg > 1) {
        REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        r = r + SH_C2[0] * xy * SHK(4) + SH_C2[1] * yz * SHK(5) + SH_C2[2] * (2 * zz - xx - yy) * SHK(6) +
            SH_C2[3] * xz * SHK(7) + SH_C2[4] * (xx - yy) * SHK(8);
        if (S->deg > 2) {
          r = r + SH_C3[0] * y * (3 * xx - yy) * SHK(9) + SH_C3[1] * xy * z * SHK(10) +
              SH_C3[2] * y * (4 * zz - xx - yy) * SHK(11) + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * SHK(12) +
              SH_C3[4] * x * (4 * zz - xx - yy) * SHK(13) + SH_C3[5] * z * (xx - yy) * SHK(14) +
              SH_C3[6] * x * (xx - 3 * yy) * SHK(15);
        }
      }
    }
#undef SHK
    r += (REAL)0.5;
    clamped[c] = r < 0;
    rgb[c] = rmax(r, 0);
  }
}

typedef struct { REAL depth; int idx; } SortItem;
static const REAL *g_sort_depth;
static int cmp_depth_idx(const void *a, const void *b) {
  int ia = *(const int *)a, ib = *(const int *)b;
  REAL da = g_sort_depth[ia], db = g_sort_depth[ib];
  if (da < db) return -1;
  if (da > db) return 1;
  return (ia > ib) - (ia < ib);
}

void gs3d_oracle_free(GsState *S) {
  if (!S) return;
  free(S->radii); free(S->rect); free(S->xy); free(S->depth); free(S->conic_o); free(S->rgb); free(S->cov3D);
  free(S->clamped); free(S->tile_start); free(S->point_list); free(S->final_T); free(S->n_contrib);
  free(S);
}

/* Blend one tile. Rows of the tile can be processed independently, so this is the OpenMP unit. */
static void blend_tile(GsState *S, int tile, REAL *out_color, REAL *out_depth, REAL *out_alpha) {
  int tx0 = (tile % S->tiles_x) * GS_TILE, ty0 = (tile / S->tiles_x) * GS_TILE;
  int64_t beg = S->tile_start[tile], end = S->tile_start[tile + 1];
  size_t HW = (size_t)S->H * S->W;
  for (int py = ty0; py < imin(ty0 + GS_TILE, S->H); py++)
    for (int px = tx0; px < imin(tx0 + GS_TILE, S->W); px++) {
      REAL T = 1, C[3] = {0, 0, 0}, Dp = 0, Wt = 0;
      int contributor = 0, last = 0;
      for (int64_t k = beg; k < end; k++) {
        contributor++;
        int g = S->point_list[k];
        const REAL *co = S->conic_o + 4 * (size_t)g;
        REAL dx = S->xy[2 * (size_t)g] - (REAL)px, dy = S->xy[2 * (size_t)g + 1] - (REAL)py;
        REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0) continue;
        REAL alpha = rmin(GS_ALPHA_MAX, co[3] * rexp(power));
        if (alpha < GS_ALPHA_MIN) continue;
        REAL test_T = T * (1 - alpha);
        if (test_T < GS_T_MIN) break;
        REAL w = alpha * T;
        for (int c = 0; c < 3; c++) C[c] += S->rgb[3 * (size_t)g + c] * w;
        Dp += S->depth[g] * w;
        Wt += w;
        T = test_T;
        last = contributor;
      }
      size_t pix = (size_t)py * S->W + px;
      S->final_T[pix] = T;
      S->n_contrib[pix] = last;
      for (int c = 0; c < 3; c++) out_color[c * HW + pix] = C[c] + T * S->bg[c];
      out_depth[pix] = Dp;
      out_alpha[pix] = Wt;
    }
}

/*
 * Forward. All float arrays are REAL.  Optional inputs may be NULL in the pairs
 * (shs | colors_precomp) and (scales+rotations | cov3D_precomp), as at the reference call site.
 * Returns a state handle for gs3d_oracle_backward (NULL on allocation failure).
 */
GsState *gs3d_oracle_forward(int N, int K, int deg, int W, int H, REAL tanfovx, REAL tanfovy, REAL scale_mod,
                             const REAL *bg, const REAL *view, const REAL *proj, const REAL *campos,
                             const REAL *means3D, const REAL *shs, const REAL *colors_precomp,
                             const REAL *opacities, const REAL *scales, const REAL *rotations,
                             const REAL *cov3D_precomp, int prefiltered,
                             REAL *out_color, int *out_radii, REAL *out_depth, REAL *out_alpha) {
  (void)prefiltered; /* reference always passes False (gaussian_renderer/__init__.py:62) */
  GsState *S = (GsState *)calloc(1, sizeof(GsState));
  if (!S) return NULL;
  S->N = N; S->K = K; S->deg = deg; S->W = W; S->H = H;
  S->tiles_x = (W + GS_TILE - 1) / GS_TILE; S->tiles_y = (H + GS_TILE - 1) / GS_TILE;
  S->tanfovx = tanfovx; S->tanfovy = tanfovy; S->scale_mod = scale_mod;
  memcpy(S->bg, bg, 3 * sizeof(REAL)); memcpy(S->view, view, 16 * sizeof(REAL));
  memcpy(S->proj, proj, 16 * sizeof(REAL)); memcpy(S->campos, campos, 3 * sizeof(REAL));
  S->means3D = means3D; S->shs = shs; S->colors_precomp = colors_precomp; S->opacities = opacities;
  S->scales = scales; S->rotations = rotations; S->cov3D_precomp = cov3D_precomp;
  S->has_sh = colors_precomp == NULL; S->has_cov_precomp = cov3D_precomp != NULL;
  int tiles = S->tiles_x * S->tiles_y;
  size_t n = (size_t)(N > 0 ? N : 1), HW = (size_t)H * W;
  S->radii = (int *)calloc(n, sizeof(int)); S->rect = (int *)calloc(n * 4, sizeof(int));
  S->xy = (REAL *)calloc(n * 2, sizeof(REAL)); S->depth = (REAL *)calloc(n, sizeof(REAL));
  S->conic_o = (REAL *)calloc(n * 4, sizeof(REAL)); S->rgb = (REAL *)calloc(n * 3, sizeof(REAL));
  S->cov3D = (REAL *)calloc(n * 6, sizeof(REAL)); S->clamped = (uint8_t *)calloc(n * 3, 1);
  S->tile_start = (int64_t *)calloc((size_t)tiles + 1, sizeof(int64_t));
  S->final_T = (REAL *)calloc(HW ? HW : 1, sizeof(REAL)); S->n_contrib = (int *)calloc(HW ? HW : 1, sizeof(int));

  /* ---- per-Gaussian preprocess ---- */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    const REAL *p = means3D + 3 * (size_t)i;
    Ewa e;
    ewa_setup(S, p, &e);
    if (e.t[2] <= GS_NEAR_Z) continue;
    REAL hom[4];
    xform4x4(S->proj, p, hom);
    REAL pw = 1 / (hom[3] + GS_W_EPS);
    REAL ndc[2] = {hom[0] * pw, hom[1] * pw};
    REAL *c6 = S->cov3D + 6 * (size_t)i;
    if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * (size_t)i, 6 * sizeof(REAL));
    else cov3d_from_scale_rot(scales + 3 * (size_t)i, scale_mod, rotations + 4 * (size_t)i, c6);
    REAL s0[3], s1[3];
    sym6_mul(c6, e.M, s0); sym6_mul(c6, e.M + 3, s1);
    REAL a = e.M[0] * s0[0] + e.M[1] * s0[1] + e.M[2] * s0[2] + GS_DILATION;
    REAL b = e.M[0] * s1[0] + e.M[1] * s1[1] + e.M[2] * s1[2];
    REAL c = e.M[3] * s1[0] + e.M[4] * s1[1] + e.M[5] * s1[2] + GS_DILATION;
    REAL det = a * c - b * b;
    if (det == 0) continue;
    REAL det_inv = 1 / det;
    REAL mid = (REAL)0.5 * (a + c);
    REAL disc = rsqrt_(rmax(GS_LAMBDA_FLOOR, mid * mid - det));
    REAL lam = rmax(mid + disc, mid - disc);
    int radius = (int)rceil(GS_RADIUS_SIGMA * rsqrt_(lam));
    REAL px = ((ndc[0] + 1) * W - 1) * (REAL)0.5, py = ((ndc[1] + 1) * H - 1) * (REAL)0.5;
    int x0 = imin(S->tiles_x, imax(0, (int)((px - radius) / GS_TILE)));
    int y0 = imin(S->tiles_y, imax(0, (int)((py - radius) / GS_TILE)));
    int x1 = imin(S->tiles_x, imax(0, (int)((px + radius + GS_TILE - 1) / GS_TILE)));
    int y1 = imin(S->tiles_y, imax(0, (int)((py + radius + GS_TILE - 1) / GS_TILE)));
    if ((x1 - x0) * (y1 - y0) == 0) continue;
    if (S->has_sh) eval_sh_rgb(S, i, S->rgb + 3 * (size_t)i, S->clamped + 3 * (size_t)i);
    else memcpy(S->rgb + 3 * (size_t)i, colors_precomp + 3 * (size_t)i, 3 * sizeof(REAL));
    S->depth[i] = e.t[2];
    S->radii[i] = radius;
    S->xy[2 * (size_t)i] = px; S->xy[2 * (size_t)i + 1] = py;
    REAL *co = S->conic_o + 4 * (size_t)i;
    co[0] = c * det_inv; co[1] = -b * det_inv; co[2] = a * det_inv; co[3] = opacities[i];
    int *rc = S->rect + 4 * (size_t)i;
    rc[0] = x0; rc[1] = y0; rc[2] = x1; rc[3] = y1;
  }
  if (out_radii) memcpy(out_radii, S->radii, (size_t)N * sizeof(int));

  /* ---- bin into tiles (counting sort by tile), then depth-sort each tile by (depth, index) ---- */
  for (int i = 0; i < N; i++) {
    if (S->radii[i] <= 0) continue;
    const int *rc = S->rect + 4 * (size_t)i;
    for (int y = rc[1]; y < rc[3]; y++)
      for (int x = rc[0]; x < rc[2]; x++) S->tile_start[y * S->tiles_x + x + 1]++;
  }
  for (int t = 0; t < tiles; t++) S->tile_start[t + 1] += S->tile_start[t];
  S->D = S->tile_start[tiles];
  S->point_list = (int *)malloc((size_t)(S->D ? S->D : 1) * sizeof(int));
  int64_t *cursor = (int64_t *)malloc((size_t)(tiles ? tiles : 1) * sizeof(int64_t));
  memcpy(cursor, S->tile_start, (size_t)tiles * sizeof(int64_t));
  for (int i = 0; i < N; i++) {
    if (S->radii[i] <= 0) continue;
    const int *rc = S->rect + 4 * (size_t)i;
    for (int y = rc[1]; y < rc[3]; y++)
      for (int x = rc[0]; x < rc[2]; x++) S->point_list[cursor[y * S->tiles_x + x]++] = i;
  }
  free(cursor);
  g_sort_depth = S->depth; /* read-only while sorting */
#pragma omp parallel for schedule(dynamic, 8)
  for (int t = 0; t < tiles; t++)
    qsort(S->point_list + S->tile_start[t], (size_t)(S->tile_start[t + 1] - S->tile_start[t]), sizeof(int),
          cmp_depth_idx);

  /* ---- per-tile front-to-back blend ---- */
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < tiles; t++) blend_tile(S, t, out_color, out_depth, out_alpha);
  return S;
}

int64_t gs3d_oracle_num_pairs(const GsState *S) { return S->D; }

/* expose forward intermediates so stage-level parity tests can compare them */
void gs3d_oracle_geom(const GsState *S, REAL *xy, REAL *depth, REAL *conic_o, REAL *rgb, int *rect) {
  size_t n = (size_t)S->N;
  if (xy) memcpy(xy, S->xy, n * 2 * sizeof(REAL));
  if (depth) memcpy(depth, S->depth, n * sizeof(REAL));
  if (conic_o) memcpy(conic_o, S->conic_o, n * 4 * sizeof(REAL));
  if (rgb) memcpy(rgb, S->rgb, n * 3 * sizeof(REAL));
  if (rect) memcpy(rect, S->rect, n * 4 * sizeof(int));
}
void gs3d_oracle_tiles(const GsState *S, int64_t *tile_start, int *point_list) {
  if (tile_start) memcpy(tile_start, S->tile_start, ((size_t)S->tiles_x * S->tiles_y + 1) * sizeof(int64_t));
  if (point_list) memcpy(point_list, S->point_list, (size_t)S->D * sizeof(int));
}

static int g_parallel_backward = 0;
void gs3d_oracle_parallel_backward(int on) { g_parallel_backward = on; }
#define ACC(dst, val) do { REAL v_ = (val); if (par) { _Pragma("omp atomic") dst += v_; } else { dst += v_; } } while (0)

/*
 * Backward. dL_dcolor[3,H,W], dL_ddepth[H,W], dL_dalpha[H,W] in; per-Gaussian gradients out (each may be
 * NULL, each is fully overwritten): means3D[N,3], means2D[N,3] (x,y in NDC-scaled units, z = 0), shs[N,K,3],
 * colors[N,3], opacities[N], scales[N,3], rotations[N,4], cov3D[N,6].  The state is read-only here, so the
 * backward can be run repeatedly on one forward (utils/mask.py:52,65,89 relies on that).
 */
void gs3d_oracle_backward(const GsState *S, const REAL *dL_dcolor, const REAL *dL_ddepth_pix, const REAL *dL_dalpha_pix,
                          REAL *g_means3D, REAL *g_means2D, REAL *g_shs, REAL *g_colors, REAL *g_opac,
                          REAL *g_scales, REAL *g_rot, REAL *g_cov3D) {
  int N = S->N, W = S->W, H = S->H;
  size_t n = (size_t)(N > 0 ? N : 1), HW = (size_t)H * W;
  /* screen-space accumulators */
  REAL *a_xy = (REAL *)calloc(n * 2, sizeof(REAL));   /* dL/d(pixel xy) * 0.5*W|H */
  REAL *a_con = (REAL *)calloc(n * 3, sizeof(REAL));  /* true dL/dA, dL/dB, dL/dC */
  REAL *a_op = (REAL *)calloc(n, sizeof(REAL));
  REAL *a_rgb = (REAL *)calloc(n * 3, sizeof(REAL));
  REAL *a_dep = (REAL *)calloc(n, sizeof(REAL));
  int tiles = S->tiles_x * S->tiles_y;
  /* Serial by default: the accumulation order is then deterministic (tests compare runs bit for bit). The CPU
   * baseline timing sets gs3d_oracle_parallel_backward(1): tiles in parallel, accumulators updated atomically. */
  const int par = g_parallel_backward;
#pragma omp parallel for schedule(dynamic, 4) if (par)
  for (int tile = 0; tile < tiles; tile++) {
    int tx0 = (tile % S->tiles_x) * GS_TILE, ty0 = (tile / S->tiles_x) * GS_TILE;
    int64_t beg = S->tile_start[tile];
    for (int py = ty0; py < imin(ty0 + GS_TILE, H); py++)
      for (int px = tx0; px < imin(tx0 + GS_TILE, W); px++) {
        size_t pix = (size_t)py * W + px;
        REAL T_final = S->final_T[pix], T = T_final;
        REAL dpix[3] = {dL_dcolor ? dL_dcolor[pix] : 0, dL_dcolor ? dL_dcolor[HW + pix] : 0,
                        dL_dcolor ? dL_dcolor[2 * HW + pix] : 0};
        REAL ddep = dL_ddepth_pix ? dL_ddepth_pix[pix] : 0, dalp = dL_dalpha_pix ? dL_dalpha_pix[pix] : 0;
        REAL bg_dot = S->bg[0] * dpix[0] + S->bg[1] * dpix[1] + S->bg[2] * dpix[2];
        REAL acc_c[3] = {0, 0, 0}, acc_d = 0, acc_a = 0, last_alpha = 0, last_c[3] = {0, 0, 0}, last_d = 0;
        for (int64_t k = beg + S->n_contrib[pix] - 1; k >= beg; k--) {
          int g = S->point_list[k];
          const REAL *co = S->conic_o + 4 * (size_t)g;
          REAL dx = S->xy[2 * (size_t)g] - (REAL)px, dy = S->xy[2 * (size_t)g + 1] - (REAL)py;
          REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0) continue;
          REAL G = rexp(power);
          REAL alpha = rmin(GS_ALPHA_MAX, co[3] * G);
          if (alpha < GS_ALPHA_MIN) continue;
          T = T / (1 - alpha);
          REAL w = alpha * T;
          REAL dL_dal = 0;
          for (int c = 0; c < 3; c++) {
            REAL col = S->rgb[3 * (size_t)g + c];
            acc_c[c] = last_alpha * last_c[c] + (1 - last_alpha) * acc_c[c];
            last_c[c] = col;
            dL_dal += (col - acc_c[c]) * dpix[c];
            ACC(a_rgb[3 * (size_t)g + c], w * dpix[c]);
          }
          REAL dep = S->depth[g];
          acc_d = last_alpha * last_d + (1 - last_alpha) * acc_d;
          last_d = dep;
          dL_dal += (dep - acc_d) * ddep;
          ACC(a_dep[g], w * ddep);
          acc_a = last_alpha + (1 - last_alpha) * acc_a;
          dL_dal += (1 - acc_a) * dalp;
          dL_dal *= T;
          last_alpha = alpha;
          dL_dal += (-T_final / (1 - alpha)) * bg_dot;
          REAL dL_dG = co[3] * dL_dal;
          REAL gdx = G * dx, gdy = G * dy;
          REAL dG_ddx = -gdx * co[0] - gdy * co[1], dG_ddy = -gdy * co[2] - gdx * co[1];
          ACC(a_xy[2 * (size_t)g], dL_dG * dG_ddx * (REAL)0.5 * W);
          ACC(a_xy[2 * (size_t)g + 1], dL_dG * dG_ddy * (REAL)0.5 * H);
          ACC(a_con[3 * (size_t)g], (REAL)-0.5 * gdx * dx * dL_dG);
          ACC(a_con[3 * (size_t)g + 1], -gdx * dy * dL_dG);
          ACC(a_con[3 * (size_t)g + 2], (REAL)-0.5 * gdy * dy * dL_dG);
          ACC(a_op[g], G * dL_dal);
        }
      }
  }

  /* ---- per-Gaussian chain ---- */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    REAL gm[3] = {0, 0, 0}, gs[3] = {0, 0, 0}, gq[4] = {0, 0, 0, 0}, gc6[6] = {0, 0, 0, 0, 0, 0};
    if (g_shs) memset(g_shs + (size_t)i * S->K * 3, 0, (size_t)S->K * 3 * sizeof(REAL));
    if (S->radii[i] > 0) {
      const REAL *p = S->means3D + 3 * (size_t)i;
      const REAL *vm = S->view, *pm = S->proj;
      Ewa e;
      ewa_setup(S, p, &e);
      const REAL *c6 = S->cov3D + 6 * (size_t)i;
      /* recompute cov2D */
      REAL s0[3], s1[3];
      sym6_mul(c6, e.M, s0); sym6_mul(c6, e.M + 3, s1);
      REAL a = e.M[0] * s0[0] + e.M[1] * s0[1] + e.M[2] * s0[2] + GS_DILATION;
      REAL b = e.M[0] * s1[0] + e.M[1] * s1[1] + e.M[2] * s1[2];
      REAL c = e.M[3] * s1[0] + e.M[4] * s1[1] + e.M[5] * s1[2] + GS_DILATION;
      REAL det = a * c - b * b, d2 = 1 / (det * det + GS_DET2_EPS);
      REAL gA = a_con[3 * (size_t)i], gB = a_con[3 * (size_t)i + 1], gC = a_con[3 * (size_t)i + 2];
      /* conic = inverse([[a,b],[b,c]]): A=c/det, B=-b/det, C=a/det */
      REAL ga = d2 * (-c * c * gA + b * c * gB - b * b * gC);
      REAL gc = d2 * (-b * b * gA + a * b * gB - a * a * gC);
      REAL gb = d2 * (2 * b * c * gA - (a * c + b * b) * gB + 2 * a * b * gC);
      /* cov2D = M Sigma M^T ; G2 = [[ga, gb/2],[gb/2, gc]] */
      REAL h = (REAL)0.5 * gb;
      const REAL *M0 = e.M, *M1 = e.M + 3;
      /* dL/dSigma (unique entries; off-diagonals appear twice) */
      REAL F[9];
      for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++)
          F[r * 3 + q] = ga * M0[r] * M0[q] + h * (M0[r] * M1[q] + M1[r] * M0[q]) + gc * M1[r] * M1[q];
      gc6[0] = F[0]; gc6[1] = 2 * F[1]; gc6[2] = 2 * F[2]; gc6[3] = F[4]; gc6[4] = 2 * F[5]; gc6[5] = F[8];
      /* dL/dM = 2 G2 M Sigma */
      REAL gM0[3], gM1[3];
      for (int q = 0; q < 3; q++) {
        gM0[q] = 2 * (ga * s0[q] + h * s1[q]);
        gM1[q] = 2 * (h * s0[q] + gc * s1[q]);
      }
      /* M = J Wrot  ->  dL/dJ = dL/dM Wrot^T */
      REAL gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
      for (int q = 0; q < 3; q++) {
        gJ00 += gM0[q] * vm[q * 4 + 0]; gJ02 += gM0[q] * vm[q * 4 + 2];
        gJ11 += gM1[q] * vm[q * 4 + 1]; gJ12 += gM1[q] * vm[q * 4 + 2];
      }
      REAL tz = e.t[2], tz2 = 1 / (tz * tz), tz3 = tz2 / tz;
      REAL gt[3];
      gt[0] = e.clamp_x ? 0 : -e.fx * tz2 * gJ02;
      gt[1] = e.clamp_y ? 0 : -e.fy * tz2 * gJ12;
      gt[2] = -e.fx * tz2 * gJ00 - e.fy * tz2 * gJ11 + 2 * e.fx * e.txc * tz3 * gJ02 + 2 * e.fy * e.tyc * tz3 * gJ12;
      /* depth output: d(t.z)/d(mean) = third row of Wrot */
      gt[2] += a_dep[i];
      for (int q = 0; q < 3; q++) gm[q] += vm[q * 4 + 0] * gt[0] + vm[q * 4 + 1] * gt[1] + vm[q * 4 + 2] * gt[2];
      /* screen position: ndc = hom.xy / (hom.w + eps) */
      REAL hom[4];
      xform4x4(pm, p, hom);
      REAL pw = 1 / (hom[3] + GS_W_EPS);
      REAL gx = a_xy[2 * (size_t)i], gy = a_xy[2 * (size_t)i + 1];
      for (int q = 0; q < 3; q++) {
        gm[q] += (pm[q * 4 + 0] * pw - pm[q * 4 + 3] * hom[0] * pw * pw) * gx +
                 (pm[q * 4 + 1] * pw - pm[q * 4 + 3] * hom[1] * pw * pw) * gy;
      }
      /* colour: SH chain (respecting the clamp mask) or direct */
      if (S->has_sh) {
        const REAL *sh = S->shs + (size_t)i * S->K * 3;
        REAL d[3] = {p[0] - S->campos[0], p[1] - S->campos[1], p[2] - S->campos[2]};
        REAL len2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], inv = 1 / rsqrt_(len2);
        REAL x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
        REAL gdir[3] = {0, 0, 0};
        for (int ch = 0; ch < 3; ch++) {
          REAL gr = S->clamped[3 * (size_t)i + ch] ? 0 : a_rgb[3 * (size_t)i + ch];
          REAL *gsh = g_shs ? g_shs + (size_t)i * S->K * 3 : NULL;
#define SHK(k) sh[(k) * 3 + ch]
#define GSH(k, v) do { if (gsh) gsh[(k) * 3 + ch] = (v) * gr; } while (0)
          REAL rx = 0, ry = 0, rz = 0;
          GSH(0, SH_C0);
          if (S->deg > 0) {
            GSH(1, -SH_C1 * y); GSH(2, SH_C1 * z); GSH(3, -SH_C1 * x);
            rx = -SH_C1 * SHK(3); ry = -SH_C1 * SHK(1); rz = SH_C1 * SHK(2);
            if (S->deg > 1) {
              REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
              GSH(4, SH_C2[0] * xy); GSH(5, SH_C2[1] * yz); GSH(6, SH_C2[2] * (2 * zz - xx - yy));
              GSH(7, SH_C2[3] * xz); GSH(8, SH_C2[4] * (xx - yy));
              rx += SH_C2[0] * y * SHK(4) + SH_C2[2] * 2 * -x * SHK(6) + SH_C2[3] * z * SHK(7) + SH_C2[4] * 2 * x * SHK(8);
              ry += SH_C2[0] * x * SHK(4) + SH_C2[1] * z * SHK(5) + SH_C2[2] * 2 * -y * SHK(6) + SH_C2[4] * 2 * -y * SHK(8);
              rz += SH_C2[1] * y * SHK(5) + SH_C2[2] * 2 * 2 * z * SHK(6) + SH_C2[3] * x * SHK(7);
              if (S->deg > 2) {
                GSH(9, SH_C3[0] * y * (3 * xx - yy)); GSH(10, SH_C3[1] * xy * z);
                GSH(11, SH_C3[2] * y * (4 * zz - xx - yy)); GSH(12, SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy));
                GSH(13, SH_C3[4] * x * (4 * zz - xx - yy)); GSH(14, SH_C3[5] * z * (xx - yy));
                GSH(15, SH_C3[6] * x * (xx - 3 * yy));
                rx += SH_C3[0] * SHK(9) * 3 * 2 * xy + SH_C3[1] * SHK(10) * yz + SH_C3[2] * SHK(11) * -2 * xy +
                      SH_C3[3] * SHK(12) * -3 * 2 * xz + SH_C3[4] * SHK(13) * (-3 * xx + 4 * zz - yy) +
                      SH_C3[5] * SHK(14) * 2 * xz + SH_C3[6] * SHK(15) * 3 * (xx - yy);
                ry += SH_C3[0] * SHK(9) * 3 * (xx - yy) + SH_C3[1] * SHK(10) * xz +
                      SH_C3[2] * SHK(11) * (-3 * yy + 4 * zz - xx) + SH_C3[3] * SHK(12) * -3 * 2 * yz +
                      SH_C3[4] * SHK(13) * -2 * xy + SH_C3[5] * SHK(14) * -2 * yz + SH_C3[6] * SHK(15) * -3 * 2 * xy;
                rz += SH_C3[1] * SHK(10) * xy + SH_C3[2] * SHK(11) * 4 * 2 * yz +
                      SH_C3[3] * SHK(12) * 3 * (2 * zz - xx - yy) + SH_C3[4] * SHK(13) * 4 * 2 * xz +
                      SH_C3[5] * SHK(14) * (xx - yy);
              }
            }
          }
#undef SHK
#undef GSH
          gdir[0] += rx * gr; gdir[1] += ry * gr; gdir[2] += rz * gr;
        }
        /* through the normalisation dir = d/|d| */
        REAL dot = x * gdir[0] + y * gdir[1] + z * gdir[2];
        gm[0] += (gdir[0] - x * dot) * inv; gm[1] += (gdir[1] - y * dot) * inv; gm[2] += (gdir[2] - z * dot) * inv;
      }
      /* Sigma = L L^T, L = R S */
      if (!S->has_cov_precomp) {
        const REAL *sc = S->scales + 3 * (size_t)i, *q = S->rotations + 4 * (size_t)i;
        REAL R[9];
        quat_to_R(q, R);
        REAL Gs[9] = {gc6[0], (REAL)0.5 * gc6[1], (REAL)0.5 * gc6[2], (REAL)0.5 * gc6[1], gc6[3], (REAL)0.5 * gc6[4],
                      (REAL)0.5 * gc6[2], (REAL)0.5 * gc6[4], gc6[5]};
        REAL sm[3] = {S->scale_mod * sc[0], S->scale_mod * sc[1], S->scale_mod * sc[2]};
        REAL gL[9], gR[9];
        for (int r = 0; r < 3; r++)
          for (int k = 0; k < 3; k++) {
            REAL acc = 0;
            for (int m = 0; m < 3; m++) acc += Gs[r * 3 + m] * R[m * 3 + k] * sm[k];
            gL[r * 3 + k] = 2 * acc;
          }
        for (int k = 0; k < 3; k++) {
          gs[k] = S->scale_mod * (R[k] * gL[k] + R[3 + k] * gL[3 + k] + R[6 + k] * gL[6 + k]);
          for (int r = 0; r < 3; r++) gR[r * 3 + k] = gL[r * 3 + k] * sm[k];
        }
        REAL r_ = q[0], x = q[1], y = q[2], z = q[3];
        gq[0] = 2 * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
        gq[1] = 2 * (y * gR[1] + z * gR[2] + y * gR[3] - 2 * x * gR[4] - r_ * gR[5] + z * gR[6] + r_ * gR[7] - 2 * x * gR[8]);
        gq[2] = 2 * (-2 * y * gR[0] + x * gR[1] + r_ * gR[2] + x * gR[3] + z * gR[5] - r_ * gR[6] + z * gR[7] - 2 * y * gR[8]);
        gq[3] = 2 * (-2 * z * gR[0] - r_ * gR[1] + x * gR[2] + r_ * gR[3] - 2 * z * gR[4] + y * gR[5] + x * gR[6] + y * gR[7]);
      }
    }
    if (g_means3D) memcpy(g_means3D + 3 * (size_t)i, gm, sizeof(gm));
    if (g_means2D) { g_means2D[3 * (size_t)i] = a_xy[2 * (size_t)i]; g_means2D[3 * (size_t)i + 1] = a_xy[2 * (size_t)i + 1]; g_means2D[3 * (size_t)i + 2] = 0; }
    if (g_colors) memcpy(g_colors + 3 * (size_t)i, a_rgb + 3 * (size_t)i, 3 * sizeof(REAL));
    if (g_opac) g_opac[i] = a_op[i];
    if (g_scales) memcpy(g_scales + 3 * (size_t)i, gs, sizeof(gs));
    if (g_rot) memcpy(g_rot + 4 * (size_t)i, gq, sizeof(gq));
    if (g_cov3D) memcpy(g_cov3D + 6 * (size_t)i, gc6, sizeof(gc6));
  }
  free(a_xy); free(a_con); free(a_op); free(a_rgb); free(a_dep);
}
