/*
 * gs2d_oracle.c — CPU restatement of the 2D-Gaussian-splat (surfel) tile rasterizer, forward + backward.
 *
 * TEST INFRASTRUCTURE ONLY (see gs3d_oracle.c): never imported, linked or called by the product path.
 *
 * PARITY UNPINNED: the arithmetic lives in `diff_surfel_rasterization` (hbb1, /root/reference/.gitmodules:7-10), an
 * un-vendored, un-pinned CUDA submodule absent from /root/reference, without tests or golden vectors.  This file
 * restates the published algorithm (Huang et al. 2024, "2D Gaussian Splatting for Geometrically Accurate Radiance
 * Fields": ray-splat intersection in the splat's uv space, low-pass with a screen-space Gaussian, depth distortion
 * and normal maps) and anchors on the reference's own call site:
 *   - arguments / output triple (color[3,H,W], radii[N], allmap[7,H,W]); scales are [N,2]; channel order of allmap:
 *     0 expected depth (un-normalised), 1 alpha, 2-4 normal (view space), 5 median depth, 6 distortion:
 *     gs2dgs/gaussian_renderer/__init__.py:111-148
 *   - precomputed transform layout [N,9] = column-major 3x3 of splat2world^T[:, [0,1,3]] @ world2pix[:, [0,1,3]]:
 *     gs2dgs/gaussian_renderer/__init__.py:78-89
 *   - SH / matrix / quaternion conventions as in gs3d_oracle.c.
 * Constants the reference tree cannot confirm are named macros.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

#define GS_TILE 16
#define GS_NEAR_Z ((REAL)0.2)          /* view-space z cull, and near plane of the depth remap */
#define GS_FAR_Z ((REAL)100.0)         /* far plane of the depth remap used by the distortion term */
#define GS_CUTOFF ((REAL)3.0)          /* splat extent in sigmas */
#define GS_FILTER_SIZE ((REAL)0.707106)/* minimum screen-space sigma (px): radius floor = cutoff * this */
#define GS_FILTER_INV_SQ ((REAL)2.0)   /* 1 / FilterSize^2: the screen-space low-pass Gaussian */
#define GS_EXTENT_FLOOR ((REAL)0.0001)
#define GS_ALPHA_MAX ((REAL)0.99)
#define GS_ALPHA_MIN ((REAL)(1.0 / 255.0))
#define GS_T_MIN ((REAL)0.0001)

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
  int N, K, deg, W, H, tiles_x, tiles_y, has_sh, has_T_precomp;
  REAL scale_mod;
  REAL bg[3], view[16], proj[16], campos[3];
  const REAL *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *T_precomp;
  int *radii, *rect;
  REAL *T;        /* N*9: Tu(3) Tv(3) Tw(3) = rows of the (u,v,1) -> (x*w, y*w, w) matrix */
  REAL *xy;       /* N*2: screen centre of the 3-sigma box */
  REAL *depth;    /* N: view-space z of the centre (sort key) */
  REAL *nrm_o;    /* N*4: view-space normal (flipped toward the camera), opacity */
  REAL *rgb;      /* N*3 */
  REAL *mult;     /* N: the dual-visibility sign */
  uint8_t *clamped;
  int64_t *tile_start;
  int *point_list;
  int64_t D;
  REAL *final_T;  /* 3*H*W: T_final, M1, M2 */
  int *n_contrib; /* 2*H*W: last contributor, median contributor */
} Gs2State;

static void quat_to_R(const REAL *q, REAL R[9]) {
  REAL r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z);     R[2] = 2 * (x * z + r * y);
  R[3] = 2 * (x * y + r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
  R[6] = 2 * (x * z - r * y);     R[7] = 2 * (y * z + r * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* Q = ndc2pix(3x4) * Proj(4x4):  rows map a homogeneous world point to (x_pix*w, y_pix*w, w). */
static void pixel_rows(const Gs2State *S, REAL Q[3][4]) {
  const REAL *pm = S->proj;
  for (int c = 0; c < 4; c++) {
    REAL p0 = pm[c * 4 + 0], p1 = pm[c * 4 + 1], p3 = pm[c * 4 + 3];
    Q[0][c] = (REAL)0.5 * S->W * p0 + (REAL)0.5 * (S->W - 1) * p3;
    Q[1][c] = (REAL)0.5 * S->H * p1 + (REAL)0.5 * (S->H - 1) * p3;
    Q[2][c] = p3;
  }
}

static void eval_sh_rgb(const Gs2State *S, int i, REAL *rgb, uint8_t *clamped) {
  const REAL *p = S->means3D + 3 * (size_t)i;
  const REAL *sh = S->shs + (size_t)i * S->K * 3;
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

static const REAL *g_sort_depth;
static int cmp_depth_idx(const void *a, const void *b) {
  int ia = *(const int *)a, ib = *(const int *)b;
  REAL da = g_sort_depth[ia], db = g_sort_depth[ib];
  if (da < db) return -1;
  if (da > db) return 1;
  return (ia > ib) - (ia < ib);
}

void gs2d_oracle_free(Gs2State *S) {
  if (!S) return;
  free(S->radii); free(S->rect); free(S->T); free(S->xy); free(S->depth); free(S->nrm_o); free(S->rgb); free(S->mult);
  free(S->clamped); free(S->tile_start); free(S->point_list); free(S->final_T); free(S->n_contrib);
  free(S);
}

/* one ray-splat evaluation shared by forward and backward */
typedef struct { REAL k[3], l[3], p[3], s[2], rho3d, rho2d, d[2], depth, G, alpha; int use3d; } Hit;
static int eval_hit(const Gs2State *S, int g, REAL px, REAL py, Hit *h) {
  const REAL *T = S->T + 9 * (size_t)g;
  const REAL *Tu = T, *Tv = T + 3, *Tw = T + 6;
  for (int q = 0; q < 3; q++) { h->k[q] = px * Tw[q] - Tu[q]; h->l[q] = py * Tw[q] - Tv[q]; }
  h->p[0] = h->k[1] * h->l[2] - h->k[2] * h->l[1];
  h->p[1] = h->k[2] * h->l[0] - h->k[0] * h->l[2];
  h->p[2] = h->k[0] * h->l[1] - h->k[1] * h->l[0];
  if (h->p[2] == 0) return 0;
  h->s[0] = h->p[0] / h->p[2]; h->s[1] = h->p[1] / h->p[2];
  h->rho3d = h->s[0] * h->s[0] + h->s[1] * h->s[1];
  h->d[0] = S->xy[2 * (size_t)g] - px; h->d[1] = S->xy[2 * (size_t)g + 1] - py;
  h->rho2d = GS_FILTER_INV_SQ * (h->d[0] * h->d[0] + h->d[1] * h->d[1]);
  h->use3d = h->rho3d <= h->rho2d;
  REAL rho = rmin(h->rho3d, h->rho2d);
  h->depth = h->use3d ? (h->s[0] * Tw[0] + h->s[1] * Tw[1]) + Tw[2] : Tw[2];
  if (h->depth < GS_NEAR_Z) return 0;
  REAL power = (REAL)-0.5 * rho;
  if (power > 0) return 0;
  h->G = rexp(power);
  h->alpha = rmin(GS_ALPHA_MAX, S->nrm_o[4 * (size_t)g + 3] * h->G);
  if (h->alpha < GS_ALPHA_MIN) return 0;
  return 1;
}

static void blend_tile(Gs2State *S, int tile, REAL *out_color, REAL *allmap) {
  int tx0 = (tile % S->tiles_x) * GS_TILE, ty0 = (tile / S->tiles_x) * GS_TILE;
  int64_t beg = S->tile_start[tile], end = S->tile_start[tile + 1];
  size_t HW = (size_t)S->H * S->W;
  const REAL fn = GS_FAR_Z / (GS_FAR_Z - GS_NEAR_Z);
  for (int py = ty0; py < imin(ty0 + GS_TILE, S->H); py++)
    for (int px = tx0; px < imin(tx0 + GS_TILE, S->W); px++) {
      REAL T = 1, C[3] = {0, 0, 0}, Nn[3] = {0, 0, 0}, Dp = 0, M1 = 0, M2 = 0, dist = 0, med = 0;
      int contributor = 0, last = 0, med_c = 0;
      for (int64_t kk = beg; kk < end; kk++) {
        contributor++;
        int g = S->point_list[kk];
        Hit h;
        if (!eval_hit(S, g, (REAL)px, (REAL)py, &h)) continue;
        REAL test_T = T * (1 - h.alpha);
        if (test_T < GS_T_MIN) break;
        REAL w = h.alpha * T;
        REAL A = 1 - T;
        REAL m = fn * (1 - GS_NEAR_Z / h.depth);
        dist += (m * m * A + M2 - 2 * m * M1) * w;
        Dp += h.depth * w; M1 += m * w; M2 += m * m * w;
        if (T > (REAL)0.5) { med = h.depth; med_c = contributor; }
        for (int c = 0; c < 3; c++) { Nn[c] += S->nrm_o[4 * (size_t)g + c] * w; C[c] += S->rgb[3 * (size_t)g + c] * w; }
        T = test_T;
        last = contributor;
      }
      size_t pix = (size_t)py * S->W + px;
      S->final_T[pix] = T; S->final_T[HW + pix] = M1; S->final_T[2 * HW + pix] = M2;
      S->n_contrib[pix] = last; S->n_contrib[HW + pix] = med_c;
      for (int c = 0; c < 3; c++) out_color[c * HW + pix] = C[c] + T * S->bg[c];
      allmap[pix] = Dp; allmap[HW + pix] = 1 - T;
      for (int c = 0; c < 3; c++) allmap[(2 + c) * HW + pix] = Nn[c];
      allmap[5 * HW + pix] = med; allmap[6 * HW + pix] = dist;
    }
}

Gs2State *gs2d_oracle_forward(int N, int K, int deg, int W, int H, REAL tanfovx, REAL tanfovy, REAL scale_mod,
                              const REAL *bg, const REAL *view, const REAL *proj, const REAL *campos,
                              const REAL *means3D, const REAL *shs, const REAL *colors_precomp, const REAL *opacities,
                              const REAL *scales, const REAL *rotations, const REAL *T_precomp,
                              REAL *out_color, int *out_radii, REAL *allmap) {
  (void)tanfovx; (void)tanfovy;
  Gs2State *S = (Gs2State *)calloc(1, sizeof(Gs2State));
  if (!S) return NULL;
  S->N = N; S->K = K; S->deg = deg; S->W = W; S->H = H;
  S->tiles_x = (W + GS_TILE - 1) / GS_TILE; S->tiles_y = (H + GS_TILE - 1) / GS_TILE;
  S->scale_mod = scale_mod;
  memcpy(S->bg, bg, 3 * sizeof(REAL)); memcpy(S->view, view, 16 * sizeof(REAL));
  memcpy(S->proj, proj, 16 * sizeof(REAL)); memcpy(S->campos, campos, 3 * sizeof(REAL));
  S->means3D = means3D; S->shs = shs; S->colors_precomp = colors_precomp; S->opacities = opacities;
  S->scales = scales; S->rotations = rotations; S->T_precomp = T_precomp;
  S->has_sh = colors_precomp == NULL; S->has_T_precomp = T_precomp != NULL;
  int tiles = S->tiles_x * S->tiles_y;
  size_t n = (size_t)(N > 0 ? N : 1), HW = (size_t)H * W;
  S->radii = (int *)calloc(n, sizeof(int)); S->rect = (int *)calloc(n * 4, sizeof(int));
  S->T = (REAL *)calloc(n * 9, sizeof(REAL)); S->xy = (REAL *)calloc(n * 2, sizeof(REAL));
  S->depth = (REAL *)calloc(n, sizeof(REAL)); S->nrm_o = (REAL *)calloc(n * 4, sizeof(REAL));
  S->rgb = (REAL *)calloc(n * 3, sizeof(REAL)); S->mult = (REAL *)calloc(n, sizeof(REAL));
  S->clamped = (uint8_t *)calloc(n * 3, 1);
  S->tile_start = (int64_t *)calloc((size_t)tiles + 1, sizeof(int64_t));
  S->final_T = (REAL *)calloc(3 * (HW ? HW : 1), sizeof(REAL)); S->n_contrib = (int *)calloc(2 * (HW ? HW : 1), sizeof(int));
  REAL Q[3][4];
  pixel_rows(S, Q);
  const REAL *vm = S->view;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    const REAL *p = means3D + 3 * (size_t)i;
    REAL pv[3];
    pv[0] = vm[0] * p[0] + vm[4] * p[1] + vm[8] * p[2] + vm[12];
    pv[1] = vm[1] * p[0] + vm[5] * p[1] + vm[9] * p[2] + vm[13];
    pv[2] = rfma(vm[10], p[2], rfma(vm[6], p[1], rfma(vm[2], p[0], vm[14])));
    if (pv[2] <= GS_NEAR_Z) continue;
    REAL *T = S->T + 9 * (size_t)i;
    REAL nv[3];
    if (T_precomp) {
      memcpy(T, T_precomp + 9 * (size_t)i, 9 * sizeof(REAL));
      nv[0] = 0; nv[1] = 0; nv[2] = 1;
    } else {
      REAL R[9];
      quat_to_R(rotations + 4 * (size_t)i, R);
      REAL sx = scale_mod * scales[2 * (size_t)i], sy = scale_mod * scales[2 * (size_t)i + 1];
      for (int r = 0; r < 3; r++) {
        T[r * 3 + 0] = sx * (Q[r][0] * R[0] + Q[r][1] * R[3] + Q[r][2] * R[6]);
        T[r * 3 + 1] = sy * (Q[r][0] * R[1] + Q[r][1] * R[4] + Q[r][2] * R[7]);
        T[r * 3 + 2] = Q[r][0] * p[0] + Q[r][1] * p[1] + Q[r][2] * p[2] + Q[r][3];
      }
      for (int r = 0; r < 3; r++) nv[r] = vm[0 * 4 + r] * R[2] + vm[1 * 4 + r] * R[5] + vm[2 * 4 + r] * R[8];
    }
    REAL cosv = -(pv[0] * nv[0] + pv[1] * nv[1] + pv[2] * nv[2]);
    if (cosv == 0) continue;
    REAL mult = cosv > 0 ? 1 : -1;
    const REAL *Tu = T, *Tv = T + 3, *Tw = T + 6;
    const REAL t[3] = {GS_CUTOFF * GS_CUTOFF, GS_CUTOFF * GS_CUTOFF, -1};
    REAL dd = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
    if (dd == 0) continue;
    REAL f[3] = {t[0] / dd, t[1] / dd, t[2] / dd};
    REAL cx = f[0] * Tu[0] * Tw[0] + f[1] * Tu[1] * Tw[1] + f[2] * Tu[2] * Tw[2];
    REAL cy = f[0] * Tv[0] * Tw[0] + f[1] * Tv[1] * Tw[1] + f[2] * Tv[2] * Tw[2];
    REAL tx = f[0] * Tu[0] * Tu[0] + f[1] * Tu[1] * Tu[1] + f[2] * Tu[2] * Tu[2];
    REAL ty = f[0] * Tv[0] * Tv[0] + f[1] * Tv[1] * Tv[1] + f[2] * Tv[2] * Tv[2];
    REAL ex = rsqrt_(rmax(GS_EXTENT_FLOOR, cx * cx - tx)), ey = rsqrt_(rmax(GS_EXTENT_FLOOR, cy * cy - ty));
    int radius = (int)rceil(rmax(rmax(ex, ey), GS_CUTOFF * GS_FILTER_SIZE));
    int x0 = imin(S->tiles_x, imax(0, (int)((cx - radius) / GS_TILE)));
    int y0 = imin(S->tiles_y, imax(0, (int)((cy - radius) / GS_TILE)));
    int x1 = imin(S->tiles_x, imax(0, (int)((cx + radius + GS_TILE - 1) / GS_TILE)));
    int y1 = imin(S->tiles_y, imax(0, (int)((cy + radius + GS_TILE - 1) / GS_TILE)));
    if ((x1 - x0) * (y1 - y0) == 0) continue;
    if (S->has_sh) eval_sh_rgb(S, i, S->rgb + 3 * (size_t)i, S->clamped + 3 * (size_t)i);
    else memcpy(S->rgb + 3 * (size_t)i, colors_precomp + 3 * (size_t)i, 3 * sizeof(REAL));
    S->depth[i] = pv[2]; S->radii[i] = radius; S->mult[i] = mult;
    S->xy[2 * (size_t)i] = cx; S->xy[2 * (size_t)i + 1] = cy;
    REAL *no = S->nrm_o + 4 * (size_t)i;
    no[0] = mult * nv[0]; no[1] = mult * nv[1]; no[2] = mult * nv[2]; no[3] = opacities[i];
    int *rc = S->rect + 4 * (size_t)i;
    rc[0] = x0; rc[1] = y0; rc[2] = x1; rc[3] = y1;
  }
  if (out_radii) memcpy(out_radii, S->radii, (size_t)N * sizeof(int));
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
  g_sort_depth = S->depth;
#pragma omp parallel for schedule(dynamic, 8)
  for (int t = 0; t < tiles; t++)
    qsort(S->point_list + S->tile_start[t], (size_t)(S->tile_start[t + 1] - S->tile_start[t]), sizeof(int), cmp_depth_idx);
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < tiles; t++) blend_tile(S, t, out_color, allmap);
  return S;
}

int64_t gs2d_oracle_num_pairs(const Gs2State *S) { return S->D; }
void gs2d_oracle_geom(const Gs2State *S, REAL *T, REAL *xy, REAL *depth, REAL *nrm_o, REAL *rgb, int *rect) {
  size_t n = (size_t)S->N;
  if (T) memcpy(T, S->T, n * 9 * sizeof(REAL));
  if (xy) memcpy(xy, S->xy, n * 2 * sizeof(REAL));
  if (depth) memcpy(depth, S->depth, n * sizeof(REAL));
  if (nrm_o) memcpy(nrm_o, S->nrm_o, n * 4 * sizeof(REAL));
  if (rgb) memcpy(rgb, S->rgb, n * 3 * sizeof(REAL));
  if (rect) memcpy(rect, S->rect, n * 4 * sizeof(int));
}
void gs2d_oracle_tiles(const Gs2State *S, int64_t *tile_start, int *point_list) {
  if (tile_start) memcpy(tile_start, S->tile_start, ((size_t)S->tiles_x * S->tiles_y + 1) * sizeof(int64_t));
  if (point_list) memcpy(point_list, S->point_list, (size_t)S->D * sizeof(int));
}

/*
 * Backward.  dL_dcolor[3,H,W], dL_dallmap[7,H,W] in; per-surfel gradients out (each may be NULL):
 * means3D[N,3], means2D[N,3] (the densification statistic: dL/dTu.z, dL/dTv.z scaled by depth*W/2, depth*H/2),
 * shs[N,K,3], colors[N,3], opacities[N], scales[N,2], rotations[N,4], transmat[N,9] (precomputed-transform path).
 */
/* Serial by default (deterministic accumulation order).  gs2d_oracle_parallel_backward(1): tiles in parallel, the
 * per-surfel accumulators updated atomically - same arithmetic per term, summation order no longer fixed.  For the
 * full-size parity tests and timings only (tests/test_fullsize_gpu.py), like gs3d_oracle_parallel_backward. */
static int g2_parallel_backward = 0;
void gs2d_oracle_parallel_backward(int on) { g2_parallel_backward = on; }
#define ACC2(dst, val) do { REAL v_ = (val); if (par) { _Pragma("omp atomic") dst += v_; } else { dst += v_; } } while (0)

void gs2d_oracle_backward(const Gs2State *S, const REAL *dL_dcolor, const REAL *dL_dallmap, REAL *g_means3D,
                          REAL *g_means2D, REAL *g_shs, REAL *g_colors, REAL *g_opac, REAL *g_scales, REAL *g_rot,
                          REAL *g_transmat) {
  int N = S->N, W = S->W, H = S->H;
  size_t n = (size_t)(N > 0 ? N : 1), HW = (size_t)H * W;
  REAL *a_T = (REAL *)calloc(n * 9, sizeof(REAL)), *a_xy = (REAL *)calloc(n * 2, sizeof(REAL));
  REAL *a_nrm = (REAL *)calloc(n * 3, sizeof(REAL)), *a_op = (REAL *)calloc(n, sizeof(REAL));
  REAL *a_rgb = (REAL *)calloc(n * 3, sizeof(REAL));
  int tiles = S->tiles_x * S->tiles_y;
  const REAL fn = GS_FAR_Z / (GS_FAR_Z - GS_NEAR_Z);
  const int par = g2_parallel_backward;
#pragma omp parallel for schedule(dynamic, 4) if (par)
  for (int tile = 0; tile < tiles; tile++) {
    int tx0 = (tile % S->tiles_x) * GS_TILE, ty0 = (tile / S->tiles_x) * GS_TILE;
    int64_t beg = S->tile_start[tile];
    for (int py = ty0; py < imin(ty0 + GS_TILE, H); py++)
      for (int px = tx0; px < imin(tx0 + GS_TILE, W); px++) {
        size_t pix = (size_t)py * W + px;
        REAL T_final = S->final_T[pix], T = T_final;
        REAL final_D = S->final_T[HW + pix], final_D2 = S->final_T[2 * HW + pix], final_A = 1 - T_final;
        int med_c = S->n_contrib[HW + pix];
        REAL dpix[3] = {0, 0, 0}, dnrm[3] = {0, 0, 0}, ddep = 0, dacc = 0, dreg = 0, dmed = 0;
        if (dL_dcolor) for (int c = 0; c < 3; c++) dpix[c] = dL_dcolor[c * HW + pix];
        if (dL_dallmap) {
          ddep = dL_dallmap[pix]; dacc = dL_dallmap[HW + pix];
          for (int c = 0; c < 3; c++) dnrm[c] = dL_dallmap[(2 + c) * HW + pix];
          dmed = dL_dallmap[5 * HW + pix]; dreg = dL_dallmap[6 * HW + pix];
        }
        REAL bg_dot = S->bg[0] * dpix[0] + S->bg[1] * dpix[1] + S->bg[2] * dpix[2];
        REAL acc_c[3] = {0, 0, 0}, acc_n[3] = {0, 0, 0}, acc_d = 0, acc_a = 0, last_alpha = 0, last_c[3] = {0, 0, 0},
             last_n[3] = {0, 0, 0}, last_d = 0, last_dL_dT = 0;
        for (int64_t kk = beg + S->n_contrib[pix] - 1; kk >= beg; kk--) {
          int g = S->point_list[kk];
          int contributor1 = (int)(kk - beg) + 1; /* 1-based list position */
          Hit h;
          if (!eval_hit(S, g, (REAL)px, (REAL)py, &h)) continue;
          const REAL *Tw = S->T + 9 * (size_t)g + 6;
          const REAL *no = S->nrm_o + 4 * (size_t)g;
          T = T / (1 - h.alpha);
          REAL w = h.alpha * T;
          REAL dL_dal = 0, dL_dz = 0;
          for (int c = 0; c < 3; c++) {
            REAL col = S->rgb[3 * (size_t)g + c];
            acc_c[c] = last_alpha * last_c[c] + (1 - last_alpha) * acc_c[c];
            last_c[c] = col;
            dL_dal += (col - acc_c[c]) * dpix[c];
            ACC2(a_rgb[3 * (size_t)g + c], w * dpix[c]);
          }
          REAL m_d = fn * (1 - GS_NEAR_Z / h.depth);
          REAL dmd_dd = (GS_FAR_Z * GS_NEAR_Z) / ((GS_FAR_Z - GS_NEAR_Z) * h.depth * h.depth);
          if (contributor1 == med_c) dL_dz += dmed;
          REAL dL_dweight = (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dreg;
          dL_dal += dL_dweight - last_dL_dT;
          last_dL_dT = dL_dweight * h.alpha + (1 - h.alpha) * last_dL_dT;
          dL_dz += 2 * w * (m_d * final_A - final_D) * dreg * dmd_dd;
          acc_d = last_alpha * last_d + (1 - last_alpha) * acc_d;
          last_d = h.depth;
          dL_dal += (h.depth - acc_d) * ddep;
          acc_a = last_alpha + (1 - last_alpha) * acc_a;
          dL_dal += (1 - acc_a) * dacc;
          for (int c = 0; c < 3; c++) {
            acc_n[c] = last_alpha * last_n[c] + (1 - last_alpha) * acc_n[c];
            last_n[c] = no[c];
            dL_dal += (no[c] - acc_n[c]) * dnrm[c];
            ACC2(a_nrm[3 * (size_t)g + c], w * dnrm[c]);
          }
          dL_dal *= T;
          last_alpha = h.alpha;
          dL_dal += (-T_final / (1 - h.alpha)) * bg_dot;
          REAL dL_dG = no[3] * dL_dal;
          dL_dz += w * ddep;
          REAL *aT = a_T + 9 * (size_t)g;
          if (h.use3d) {
            REAL ds0 = dL_dG * -h.G * h.s[0] + dL_dz * Tw[0], ds1 = dL_dG * -h.G * h.s[1] + dL_dz * Tw[1];
            REAL q0 = ds0 / h.p[2], q1 = ds1 / h.p[2];
            REAL dp[3] = {q0, q1, -(q0 * h.s[0] + q1 * h.s[1])};
            REAL dk[3] = {h.l[1] * dp[2] - h.l[2] * dp[1], h.l[2] * dp[0] - h.l[0] * dp[2], h.l[0] * dp[1] - h.l[1] * dp[0]};
            REAL dl[3] = {dp[1] * h.k[2] - dp[2] * h.k[1], dp[2] * h.k[0] - dp[0] * h.k[2], dp[0] * h.k[1] - dp[1] * h.k[0]};
            const REAL dz_dTw[3] = {h.s[0], h.s[1], 1};
            for (int q = 0; q < 3; q++) {
              ACC2(aT[q], -dk[q]);
              ACC2(aT[3 + q], -dl[q]);
              ACC2(aT[6 + q], (REAL)px * dk[q] + (REAL)py * dl[q] + dL_dz * dz_dTw[q]);
            }
          } else {
            ACC2(a_xy[2 * (size_t)g], dL_dG * (-h.G * GS_FILTER_INV_SQ * h.d[0]));
            ACC2(a_xy[2 * (size_t)g + 1], dL_dG * (-h.G * GS_FILTER_INV_SQ * h.d[1]));
            ACC2(aT[8], dL_dz);
          }
          ACC2(a_op[g], h.G * dL_dal);
        }
      }
  }
  REAL Q[3][4];
  pixel_rows(S, Q);
  const REAL *vm = S->view;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    REAL gm[3] = {0, 0, 0}, gs[2] = {0, 0}, gq[4] = {0, 0, 0, 0}, gT[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gm2[3] = {0, 0, 0};
    if (g_shs) memset(g_shs + (size_t)i * S->K * 3, 0, (size_t)S->K * 3 * sizeof(REAL));
    if (S->radii[i] > 0) {
      const REAL *p = S->means3D + 3 * (size_t)i;
      const REAL *T = S->T + 9 * (size_t)i;
      const REAL *Tu = T, *Tv = T + 3, *Tw = T + 6;
      memcpy(gT, a_T + 9 * (size_t)i, 9 * sizeof(REAL));
      /* densification statistic reads the blend-accumulated dL/dT (before the centre terms are folded in) */
      gm2[0] = gT[2] * S->depth[i] * (REAL)0.5 * W;
      gm2[1] = gT[5] * S->depth[i] * (REAL)0.5 * H;
      REAL gx = a_xy[2 * (size_t)i], gy = a_xy[2 * (size_t)i + 1];
      if (gx != 0 || gy != 0) { /* centre of the 3-sigma box as a function of T */
        const REAL t[3] = {GS_CUTOFF * GS_CUTOFF, GS_CUTOFF * GS_CUTOFF, -1};
        REAL dd = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
        REAL f[3] = {t[0] / dd, t[1] / dd, t[2] / dd};
        REAL dLdd = 0;
        for (int q = 0; q < 3; q++) {
          gT[q] += gx * f[q] * Tw[q];
          gT[3 + q] += gy * f[q] * Tw[q];
          gT[6 + q] += gx * f[q] * Tu[q] + gy * f[q] * Tv[q];
          dLdd += (gx * Tu[q] * Tw[q] + gy * Tv[q] * Tw[q]) * f[q];
        }
        dLdd *= -1 / dd;
        for (int q = 0; q < 3; q++) gT[6 + q] += dLdd * 2 * t[q] * Tw[q];
      }
      if (!S->has_T_precomp) {
        const REAL *q = S->rotations + 4 * (size_t)i;
        REAL R[9];
        quat_to_R(q, R);
        REAL sx = S->scale_mod * S->scales[2 * (size_t)i], sy = S->scale_mod * S->scales[2 * (size_t)i + 1];
        /* dL/d(column j of [tu*sx | tv*sy | p]) = sum_r gT[r][j] * Q[r].xyz */
        REAL gh[3][3];
        for (int j = 0; j < 3; j++)
          for (int c = 0; c < 3; c++) gh[j][c] = gT[0 * 3 + j] * Q[0][c] + gT[1 * 3 + j] * Q[1][c] + gT[2 * 3 + j] * Q[2][c];
        for (int c = 0; c < 3; c++) gm[c] = gh[2][c];
        REAL gtn_v[3] = {S->mult[i] * a_nrm[3 * (size_t)i], S->mult[i] * a_nrm[3 * (size_t)i + 1], S->mult[i] * a_nrm[3 * (size_t)i + 2]};
        REAL gtn[3];
        for (int c = 0; c < 3; c++) gtn[c] = vm[c * 4 + 0] * gtn_v[0] + vm[c * 4 + 1] * gtn_v[1] + vm[c * 4 + 2] * gtn_v[2];
        REAL gR[9];
        for (int r = 0; r < 3; r++) { gR[r * 3 + 0] = gh[0][r] * sx; gR[r * 3 + 1] = gh[1][r] * sy; gR[r * 3 + 2] = gtn[r]; }
        gs[0] = S->scale_mod * (gh[0][0] * R[0] + gh[0][1] * R[3] + gh[0][2] * R[6]);
        gs[1] = S->scale_mod * (gh[1][0] * R[1] + gh[1][1] * R[4] + gh[1][2] * R[7]);
        REAL r_ = q[0], x = q[1], y = q[2], z = q[3];
        gq[0] = 2 * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
        gq[1] = 2 * (y * gR[1] + z * gR[2] + y * gR[3] - 2 * x * gR[4] - r_ * gR[5] + z * gR[6] + r_ * gR[7] - 2 * x * gR[8]);
        gq[2] = 2 * (-2 * y * gR[0] + x * gR[1] + r_ * gR[2] + x * gR[3] + z * gR[5] - r_ * gR[6] + z * gR[7] - 2 * y * gR[8]);
        gq[3] = 2 * (-2 * z * gR[0] - r_ * gR[1] + x * gR[2] + r_ * gR[3] - 2 * z * gR[4] + y * gR[5] + x * gR[6] + y * gR[7]);
      }
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
              rz += SH_C2[1] * y * SHK(5) + SH_C2[2] * 4 * z * SHK(6) + SH_C2[3] * x * SHK(7);
              if (S->deg > 2) {
                GSH(9, SH_C3[0] * y * (3 * xx - yy)); GSH(10, SH_C3[1] * xy * z);
                GSH(11, SH_C3[2] * y * (4 * zz - xx - yy)); GSH(12, SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy));
                GSH(13, SH_C3[4] * x * (4 * zz - xx - yy)); GSH(14, SH_C3[5] * z * (xx - yy));
                GSH(15, SH_C3[6] * x * (xx - 3 * yy));
                rx += SH_C3[0] * SHK(9) * 6 * xy + SH_C3[1] * SHK(10) * yz + SH_C3[2] * SHK(11) * -2 * xy +
                      SH_C3[3] * SHK(12) * -6 * xz + SH_C3[4] * SHK(13) * (-3 * xx + 4 * zz - yy) +
                      SH_C3[5] * SHK(14) * 2 * xz + SH_C3[6] * SHK(15) * 3 * (xx - yy);
                ry += SH_C3[0] * SHK(9) * 3 * (xx - yy) + SH_C3[1] * SHK(10) * xz +
                      SH_C3[2] * SHK(11) * (-3 * yy + 4 * zz - xx) + SH_C3[3] * SHK(12) * -6 * yz +
                      SH_C3[4] * SHK(13) * -2 * xy + SH_C3[5] * SHK(14) * -2 * yz + SH_C3[6] * SHK(15) * -6 * xy;
                rz += SH_C3[1] * SHK(10) * xy + SH_C3[2] * SHK(11) * 8 * yz +
                      SH_C3[3] * SHK(12) * 3 * (2 * zz - xx - yy) + SH_C3[4] * SHK(13) * 8 * xz +
                      SH_C3[5] * SHK(14) * (xx - yy);
              }
            }
          }
#undef SHK
#undef GSH
          gdir[0] += rx * gr; gdir[1] += ry * gr; gdir[2] += rz * gr;
        }
        REAL dot = x * gdir[0] + y * gdir[1] + z * gdir[2];
        gm[0] += (gdir[0] - x * dot) * inv; gm[1] += (gdir[1] - y * dot) * inv; gm[2] += (gdir[2] - z * dot) * inv;
      }
    }
    if (g_means3D) memcpy(g_means3D + 3 * (size_t)i, gm, sizeof(gm));
    if (g_means2D) memcpy(g_means2D + 3 * (size_t)i, gm2, sizeof(gm2));
    if (g_colors) memcpy(g_colors + 3 * (size_t)i, a_rgb + 3 * (size_t)i, 3 * sizeof(REAL));
    if (g_opac) g_opac[i] = a_op[i];
    if (g_scales) memcpy(g_scales + 2 * (size_t)i, gs, sizeof(gs));
    if (g_rot) memcpy(g_rot + 4 * (size_t)i, gq, sizeof(gq));
    if (g_transmat) memcpy(g_transmat + 9 * (size_t)i, gT, sizeof(gT));
  }
  free(a_T); free(a_xy); free(a_nrm); free(a_op); free(a_rgb);
}
