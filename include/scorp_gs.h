/*
 * scorp_gs.h — C ABI of libscorp_gs.so, the MI355X (gfx950) Gaussian-splat rasterization backend.
 *
 * This is the drop-in boundary for the one hot path of PolySummit/SCORP: everything below
 * `GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
 * cov3D_precomp)` in gs3dgs/gaussian_renderer/__init__.py:51-66,101-111 (and its 2DGS twin,
 * gs2dgs/gaussian_renderer/__init__.py:51-67,111-120) plus `simple_knn._C.distCUDA2`
 * (gs3dgs/scene/gaussian_model.py:22,177).  The reference binds those through three third-party CUDA torch
 * extensions (`diff_gaussian_rasterization`, `diff_surfel_rasterization`, `simple_knn`; .gitmodules:7-17)
 * whose source is not in the reference tree, so there is no upstream C signature to match: the entry points
 * below are what a ctypes / pybind / cgo binding for this path binds instead.  Plain C: raw device pointers,
 * ints and floats; no torch or pybind types.
 *
 * Conventions (all fixed by the reference's call site):
 *   - every array is fp32, contiguous, resident on the current HIP device, owned by the caller;
 *   - matrices are the 16 floats of the *transposed* 4x4 the reference stores (row-vector convention,
 *     gs3dgs/scene/cameras.py:82-97), i.e. element [r][c] of the maths matrix is m[c*4 + r];
 *   - quaternions are (w,x,y,z) and already normalised by the caller (gaussian_model.py:131-132);
 *   - cov3D_precomp is xx,xy,xz,yy,yz,zz (gs3dgs/utils/general_utils.py:79-88);
 *   - shs is [N, sh_coeffs, 3]; only the first (sh_degree+1)^2 coefficients are read;
 *   - exactly one of {shs, colors_precomp} and one of {scales+rotations, cov3D_precomp} is non-NULL;
 *   - outputs: color[3,H,W] (background composited), depth[1,H,W] = sum z*alpha*T (NOT normalised: the caller
 *     divides by alpha, gaussian_renderer/__init__.py:113), alpha[1,H,W] = sum alpha*T, radii[N] int32.
 *
 * Threading: re-entrant; all work is enqueued on the stream passed in; the only host synchronisation is in
 * scorp_gs3d_num_pairs().  Errors: 0 = ok, negative = failure with a message in scorp_last_error()
 * (thread-local).  With `debug` set every kernel is followed by a stream sync + error check
 * (the reference's `pipe.debug`, gaussian_renderer/__init__.py:63).
 */
#ifndef SCORP_GS_H
#define SCORP_GS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *scorp_stream_t; /* a hipStream_t (NULL = the default stream) */

#define SCORP_OK 0
#define SCORP_ERR_INVALID (-1)   /* bad argument (NULL, size, shape) */
#define SCORP_ERR_HIP (-2)       /* a HIP call or kernel failed */
#define SCORP_ERR_OVERFLOW (-3)  /* the pair buffer was smaller than the number of (tile,splat) pairs */

/* The 12 fields of GaussianRasterizationSettings (gaussian_renderer/__init__.py:51-64) + the 8 call arguments. */
typedef struct ScorpGs3dInputs {
  int32_t num_gaussians;    /* N */
  int32_t sh_degree;        /* active degree, 0..3 */
  int32_t sh_coeffs;        /* coefficients allocated per Gaussian in `shs` (= (max_sh_degree+1)^2) */
  int32_t image_width;
  int32_t image_height;
  float tanfovx;
  float tanfovy;
  float scale_modifier;
  int32_t prefiltered;      /* accepted for signature parity; the reference always passes False */
  int32_t debug;
  const float *bg;          /* [3]  device */
  const float *viewmatrix;  /* [16] device */
  const float *projmatrix;  /* [16] device */
  const float *campos;      /* [3]  device */
  const float *means3D;     /* [N,3] */
  const float *shs;         /* [N,sh_coeffs,3] or NULL */
  const float *colors_precomp; /* [N,3] or NULL */
  const float *opacities;   /* [N] (the reference passes [N,1]) */
  const float *scales;      /* [N,3] or NULL */
  const float *rotations;   /* [N,4] or NULL */
  const float *cov3D_precomp; /* [N,6] or NULL */
  /* ---- optional "raw parameter" convention (zero / NULL = the reference call-site convention above) ----
   * The GaussianModel stores logit opacity, log scale, an un-normalised quaternion and the SH coefficients split
   * into _features_dc[N,1,3] / _features_rest[N,K-1,3]; its properties activate and torch.cat them on every view
   * (gs3dgs/scene/gaussian_model.py:126-146).  With these fields the kernels do that themselves and the backward
   * returns gradients w.r.t. the raw values, so the training harness skips six elementwise/cat kernels per view. */
  const float *shs_rest;    /* non-NULL: `shs` is [N,1,3] (degree 0) and shs_rest is [N,sh_coeffs-1,3] */
  int32_t raw_params;       /* bit 0: opacities are logits (sigmoid); bit 1: scales are logs (exp); bit 2: rotations
                               are un-normalised (x / max(|x|, 1e-12)) */
  /* ---- optional: V views of the SAME Gaussians in one launch set (0 or 1 = one view, the reference's call) ----
   * The align loop renders one object from ~15 cameras per pose hypothesis, forward only
   * (align_3dgs_clpe_9dof.py:157-169,336-368); 100 k Gaussians at 800x800 cannot fill the chip, so a single view is a
   * chain of launch-latency-bound kernels.  With num_views = V > 1: viewmatrix / projmatrix / campos are arrays of V
   * entries ([V,16], [V,16], [V,3]), image_height is the height of ONE view (a multiple of 16) and every per-view size
   * is multiplied by V: the V images are rendered as one image of V * image_height rows (view v = rows
   * [v * image_height, (v + 1) * image_height)), Gaussian i of view v is "virtual Gaussian" v * N + i.  So
   * out_radii is [V * N], color [3, V * H, W], depth / alpha [V * H, W]; state / pairs are sized with
   * scorp_gs3d_state_bytes(V * N, W, V * H).  Forward only: scorp_gs3d_backward* refuse such inputs. */
  int32_t num_views;
} ScorpGs3dInputs;

/* Gradients w.r.t. the 8 call arguments; any pointer may be NULL (not wanted). Each is fully overwritten. */
typedef struct ScorpGs3dGrads {
  float *means3D;        /* [N,3] */
  float *means2D;        /* [N,3]: (dL/dndc_x, dL/dndc_y, 0) — consumed by add_densification_stats, gaussian_model.py:603-605 */
  float *shs;            /* [N,sh_coeffs,3] */
  float *colors_precomp; /* [N,3] */
  float *opacities;      /* [N] */
  float *scales;         /* [N,3] */
  float *rotations;      /* [N,4] */
  float *cov3D_precomp;  /* [N,6] */
  float *shs_rest;       /* [N,sh_coeffs-1,3] when the forward was given shs_rest (then `shs` is [N,1,3]) */
} ScorpGs3dGrads;

int scorp_version(void);
/* First 16 hex digits of the sha256 over the kernel sources (csrc/ *.hip, *.hpp, this header) the library was built
 * from; profiles/traffic.json and profiles/valu_mix.json carry the same stamp. */
const char *scorp_source_sha(void);
const char *scorp_last_error(void);

/* ---- workspace sizing (pure host arithmetic) ---- */
/* Forward state: per-Gaussian projected records, per-tile ranges, per-pixel final-T / last-contributor. */
size_t scorp_gs3d_state_bytes(int32_t num_gaussians, int32_t image_width, int32_t image_height);
/* Pair buffer for `capacity` (tile,splat) pairs: unsorted 64-bit keys + the depth-sorted 32-bit splat list.  After the
 * sort the key region is reused by the render for one byte per (8x8 pixel block, list entry): whether the splat's
 * footprint reaches the block.  The backward replays exactly those entries, so it must get the SAME pair buffer the
 * render filled (it only reads it). */
size_t scorp_gs3d_pairs_bytes(uint64_t capacity);
/* Scratch of one backward call (per-Gaussian screen-space gradient accumulators). */
size_t scorp_gs3d_backward_scratch_bytes(int32_t num_gaussians);

/* ---- forward, in three steps so the caller owns every allocation ---- */
/* 1. project + cull every Gaussian, count splats per tile, prefix-sum the counts. Enqueue only.
 *    Writes radii[N]. `state` must be scorp_gs3d_state_bytes() large and 256-byte aligned. */
int scorp_gs3d_preprocess(const ScorpGs3dInputs *in, int32_t *out_radii, void *state, size_t state_bytes,
                          scorp_stream_t stream);
/* 2. number of (tile,splat) pairs of the preprocess just enqueued. Synchronises the stream (one 8-byte D2H).
 *    A caller that sizes the pair buffer from a previous view may skip this and check
 *    scorp_gs3d_check_overflow() later instead. */
int scorp_gs3d_num_pairs(const void *state, scorp_stream_t stream, uint64_t *num_pairs);
/* 3. bucket pairs by tile, depth-sort each tile, blend front to back. Enqueue only. If the pair buffer is too
 *    small nothing is written out of bounds, outputs are undefined and scorp_gs3d_check_overflow() reports it. */
int scorp_gs3d_render(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity, float *out_color,
                      float *out_depth, float *out_alpha, scorp_stream_t stream);
/* 3'. the same images with nothing left behind for scorp_gs3d_backward (no per-pixel state, no cull verdicts in the
 *     pair buffer): for the calls the reference makes under torch.no_grad() (align_3dgs_clpe_9dof.py:157-169 scoring
 *     renders, evaluation views).  Same arguments, same outputs bit for bit. */
int scorp_gs3d_render_image(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity, float *out_color,
                            float *out_depth, float *out_alpha, scorp_stream_t stream);
/* 3''. render-and-compare in one: step 3 for a caller that only wants to know how far the render is from a target (the
 *      rotation sweep's scoring, align_3dgs_clpe_9dof.py:80-111 / :336-368, as scorp_gs3d_pose_score_accumulate defines it).
 *      No image is written and no colour accumulated; the image's rows are cut into bands of rows_per_score rows (a
 *      multiple of 16 dividing the image height - with num_views = V stacked views: V * image_height, or a multiple of it
 *      when several hypotheses share the launch), every band is compared with the SAME target tgt_depth / tgt_alpha
 *      [rows_per_score, W] (tgt_depth normalised, as scorp_gs3d_render_tail writes it), and
 *        acc[j] += scale * sum over band j of |alpha - tgt_alpha| + |nan_to_num(depth / alpha, 0, 0) - tgt_depth|.
 *      The per-block partial sums are added in a fixed order: two runs give the same bits. */
int scorp_gs3d_render_score(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity, const float *tgt_depth,
                            const float *tgt_alpha, int32_t rows_per_score, float scale, float *acc, scorp_stream_t stream);
/* Synchronises; returns SCORP_ERR_OVERFLOW if the last render on this state needed more than `capacity` pairs
 * (and the needed count in *num_pairs), SCORP_OK otherwise. */
int scorp_gs3d_check_overflow(const void *state, scorp_stream_t stream, uint64_t *num_pairs);

/* ---- backward ---- */
/* `state` and `pairs` are the buffers of the matching forward and are only READ, so several backward passes
 * may run on one forward (utils/mask.py:52,65,89 does). dL_dcolor[3,H,W]; dL_ddepth / dL_dalpha [H,W] or NULL. */
int scorp_gs3d_backward(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                        const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                        const ScorpGs3dGrads *grads, void *scratch, size_t scratch_bytes, scorp_stream_t stream);
/* The same with `flags`.  The blend backward reduces over a block's pixels on the matrix cores; by default the two
 * factors travel as two fp16 terms each (22 bits, exact products, fp32 accumulation: "split" form, the fast one).
 * SCORP_BACKWARD_EXACT_FP32 selects fp32 MFMAs throughout (the form the reference CUDA's fp32 arithmetic corresponds
 * to; ~1.3x the blend-backward time): the parity tests compare the two.  scorp_gs3d_backward == flags 0. */
#define SCORP_BACKWARD_EXACT_FP32 1u
/* SCORP_BACKWARD_SCRATCH_ZEROED: the caller guarantees that the first num_gaussians * 64 bytes of `scratch` (the
 * per-Gaussian accumulator rows) are zero when the call starts, so the library skips its 64 MB-per-million fill.
 * scorp_gs3d_train_view uses it: there the blend FORWARD's waves, whose memory pipes idle, clear the rows on the way. */
#define SCORP_BACKWARD_SCRATCH_ZEROED 2u
/* SCORP_BACKWARD_DETERMINISTIC: no float atomics.  Every (8x8 block, hit) writes its ten sums as one plain 64-byte row
 * partial[4 * pair + block], pair = the ordinal of its (Gaussian, tile) pair in Gaussian-major order (an exclusive scan of
 * the Gaussians' tile counts + the tile's rank in the Gaussian's tile mask), with one flag byte per row; a second kernel
 * then adds, per Gaussian, its rows - contiguous by construction - in a FIXED order: its tiles in mask order, the four
 * blocks of a tile in order.  Two runs on the same inputs give the same bits: `get_mask3d` votes on the SIGN of repeated
 * backward passes on one forward (utils/mask.py:52,65,89,124), which atomic accumulation order can flip for sums near
 * zero.  Needs the larger scratch of scorp_gs3d_backward_scratch_bytes_ex. */
#define SCORP_BACKWARD_DETERMINISTIC 4u
/* scratch size for scorp_gs3d_backward_ex with `flags`: the accumulator rows (num_gaussians * 64 bytes), plus, for
 * SCORP_BACKWARD_DETERMINISTIC, num_gaussians + 1 pair ordinals, 4 * capacity flag bytes and 4 * capacity rows of 64 bytes */
size_t scorp_gs3d_backward_scratch_bytes_ex(int32_t num_gaussians, int32_t image_width, int32_t image_height, uint64_t capacity,
                                            uint32_t flags);
int scorp_gs3d_backward_ex(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                           const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                           const ScorpGs3dGrads *grads, void *scratch, size_t scratch_bytes, uint32_t flags,
                           scorp_stream_t stream);

/* ---- introspection for stage-level parity tests (device->host copies; synchronises) ---- */
/* xy[N,2], depth[N], conic_opacity[N,4], rgb[N,3], rect[N,4] (tile units, max exclusive); any may be NULL. */
int scorp_gs3d_debug_geom(const void *state, int32_t num_gaussians, int32_t image_width, int32_t image_height,
                          float *xy, float *depth, float *conic_opacity, float *rgb, int32_t *rect,
                          scorp_stream_t stream);
/* Work statistics of the last scorp_gs3d_render on this state: out3[0] = (8x8 block, splat) iterations the blend forward
 * ran (each evaluates 64 pixel-splat pairs: P = 64 * out3[0]), out3[1] = those the blend backward replays (per block, up
 * to its deepest last contributor), out3[2] = number of blocks. */
int scorp_gs3d_debug_work(const void *state, int32_t num_gaussians, int32_t image_width, int32_t image_height,
                          uint64_t *out3, scorp_stream_t stream);
/* tile_start[tiles+1] (uint32) and the sorted splat list point_list[num_pairs] (uint32), both in RASTER tile order
 * (whatever order the lists have in the pair buffer: cell-major under the two-level binning). */
int scorp_gs3d_debug_tiles(const void *state, const void *pairs, uint64_t capacity, int32_t num_gaussians,
                           int32_t image_width, int32_t image_height, uint32_t *tile_start, uint32_t *point_list,
                           scorp_stream_t stream);

/* ---- 2D Gaussian splatting (surfels): replaces diff_surfel_rasterization (gs2dgs/gaussian_renderer/__init__.py:111-120) ----
 * Same ScorpGs3dInputs / ScorpGs3dGrads structs with two reinterpretations: `scales` is [N,2]
 * (gs2dgs/scene/gaussian_model.py:49) and `cov3D_precomp` is the precomputed [N,9] splat->pixel transform
 * (gs2dgs/gaussian_renderer/__init__.py:78-89).  Outputs: color[3,H,W], radii[N], allmap[7,H,W] with channels
 * 0 expected depth (un-normalised), 1 alpha, 2-4 view-space normal, 5 median depth, 6 depth distortion (:131-148).
 * The means2D gradient is the densification statistic the 2DGS model accumulates (gaussian_model.py:495).
 * Pair counting / overflow checks are shared: scorp_gs3d_num_pairs / scorp_gs3d_check_overflow work on this state. */
size_t scorp_gs2d_state_bytes(int32_t num_gaussians, int32_t image_width, int32_t image_height);
size_t scorp_gs2d_backward_scratch_bytes(int32_t num_gaussians);
int scorp_gs2d_preprocess(const ScorpGs3dInputs *in, int32_t *out_radii, void *state, size_t state_bytes,
                          scorp_stream_t stream);
int scorp_gs2d_render(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity, float *out_color,
                      float *out_allmap, scorp_stream_t stream);
/* As scorp_gs3d_render_image: the same images with nothing left behind for scorp_gs2d_backward. */
int scorp_gs2d_render_image(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity, float *out_color,
                            float *out_allmap, scorp_stream_t stream);
int scorp_gs2d_backward(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                        const float *dL_dcolor, const float *dL_dallmap, const ScorpGs3dGrads *grads, void *scratch,
                        size_t scratch_bytes, scorp_stream_t stream);
/* The same with the `flags` of scorp_gs3d_backward_ex.  The surfel backward reduces eight per-pixel values of a hit over
 * the block's pixels on the matrix cores; by default they travel as two fp16 terms each under a per-hit power of two (22
 * bits, exact products, fp32 accumulation).  SCORP_BACKWARD_EXACT_FP32: the values and the upstream gradients stay fp32
 * and the reduction runs on fp32 MFMAs - no operand narrower than the reference's fp32 arithmetic
 * (gs2dgs/gaussian_renderer/__init__.py:111-120 under train_2dgs.py:142-150).  SCORP_BACKWARD_DETERMINISTIC: no float
 * atomics - one plain 80-byte row per (8x8 block, hit) and an ordered per-surfel sum, as in 3-D; two runs give the same bits
 * (utils/mask.py:52-124 votes on signs of repeated backward passes).  SCORP_BACKWARD_SCRATCH_ZEROED is ignored.
 * scratch: scorp_gs2d_backward_scratch_bytes_ex(flags) (num_gaussians * 80 bytes, plus for the deterministic form
 * num_gaussians + 1 pair ordinals, 4 * capacity flag bytes and 4 * capacity rows of 80 bytes). */
size_t scorp_gs2d_backward_scratch_bytes_ex(int32_t num_gaussians, int32_t image_width, int32_t image_height, uint64_t capacity,
                                            uint32_t flags);
int scorp_gs2d_backward_ex(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity,
                           const float *dL_dcolor, const float *dL_dallmap, const ScorpGs3dGrads *grads, void *scratch,
                           size_t scratch_bytes, uint32_t flags, scorp_stream_t stream);
/* T[N,9], xy[N,2], depth[N], normal_opacity[N,4], rgb[N,3], rect[N,4]; any may be NULL (stage-level parity tests). */
int scorp_gs2d_debug_geom(const void *state, int32_t num_gaussians, int32_t image_width, int32_t image_height, float *T,
                          float *xy, float *depth, float *normal_opacity, float *rgb, int32_t *rect,
                          scorp_stream_t stream);

int scorp_gs2d_debug_tiles(const void *state, const void *pairs, uint64_t capacity, int32_t num_gaussians,
                           int32_t image_width, int32_t image_height, uint32_t *tile_start, uint32_t *point_list,
                           scorp_stream_t stream);

/* ---- what render() does to the rasterizer's outputs (gs3dgs/gaussian_renderer/__init__.py:113-120), one launch:
 * out_depth[HW] = nan_to_num(depth / alpha, nan=0, posinf=0); out_visible[N] (bytes, 0/1) = radii > 0.
 * The backward gives dL/ddepth, dL/dalpha for an upstream gradient on out_depth (zeros where alpha = 0). */
int scorp_gs3d_render_tail(const float *depth, const float *alpha, int64_t num_pixels, const int32_t *radii,
                           int32_t num_gaussians, float *out_depth, uint8_t *out_visible, scorp_stream_t stream);
int scorp_gs3d_render_tail_backward(const float *g_out_depth, const float *depth, const float *alpha, int64_t num_pixels,
                                    float *g_depth, float *g_alpha, scorp_stream_t stream);

/* ---- scoring one pose hypothesis against a target view (the render-and-compare form of the rotation sweep,
 * align_3dgs_clpe_9dof.py:80-111 / :336-368), straight from the rasterizer's raw outputs, one launch:
 * acc[0] += scale * sum over the pixels of |alpha - tgt_alpha| + |nan_to_num(depth / alpha, 0, 0) - tgt_depth|
 * (tgt_depth is a NORMALISED depth as scorp_gs3d_render_tail writes it).  One float atomic per workgroup: the order of
 * the partial sums is not fixed. */
int scorp_gs3d_pose_score_accumulate(const float *depth, const float *alpha, const float *tgt_depth, const float *tgt_alpha,
                                     int64_t num_pixels, float scale, float *acc, scorp_stream_t stream);

/* ---- per-pixel tail of the 2DGS render(): gs2dgs/gaussian_renderer/__init__.py:131-160 over
 * gs2dgs/utils/point_utils.py:9-40 (depths_to_points, depth_to_normal) ----
 * allmap[7,H,W] -> render_alpha[1,H,W], render_normal[3,H,W] (rotated to world space by viewmatrix[:3,:3]),
 * render_dist[1,H,W], surf_depth[1,H,W] = expected*(1-depth_ratio) + depth_ratio*median (both nan_to_num(.,0,0)),
 * surf_normal[3,H,W] = normalize(cross(dP/dy, dP/dx)) * alpha with P = surf_depth*rays_d + rays_o, zero on the
 * one-pixel border.  viewmatrix: the camera's world_view_transform (16 floats, as passed to the rasterizer);
 * rays_d[H*W,3] / rays_o[3]: the camera's per-pixel ray table (point_utils.py:9-22).  All pointers are device
 * pointers.  The backward takes the five upstream gradients (any may be NULL = zero) and writes g_allmap[7,H,W];
 * surf_depth is the forward's output.  Where PyTorch's chain yields 0/0 = NaN (empty pixels) it writes 0. */
int scorp_gs2d_maps_forward(int32_t image_width, int32_t image_height, const float *allmap, const float *viewmatrix,
                            const float *rays_d, const float *rays_o, float depth_ratio, float *render_alpha,
                            float *render_normal, float *render_dist, float *surf_depth, float *surf_normal,
                            scorp_stream_t stream);
int scorp_gs2d_maps_backward(int32_t image_width, int32_t image_height, const float *allmap, const float *viewmatrix,
                             const float *rays_d, const float *rays_o, float depth_ratio, const float *surf_depth,
                             const float *g_render_alpha, const float *g_render_normal, const float *g_render_dist,
                             const float *g_surf_depth, const float *g_surf_normal, float *g_allmap,
                             scorp_stream_t stream);

/* ---- the 2DGS regularisers of train_2dgs.py:142-150, fused over the same per-pixel tail ----
 * out2 (device) = { lambda_normal * mean(1 - render_normal . surf_normal), lambda_dist * mean(render_dist) } straight
 * from allmap (no maps are materialised); the backward writes g_allmap[7,H,W] for upstream gradients g_out2 (device,
 * two floats; NULL = ones).  workspace: scorp_gs2d_regularizers_workspace_bytes (per-workgroup partial sums; the
 * reduction order is fixed, so the value is deterministic). */
size_t scorp_gs2d_regularizers_workspace_bytes(int32_t image_width, int32_t image_height);
int scorp_gs2d_regularizers_forward(int32_t image_width, int32_t image_height, const float *allmap, const float *viewmatrix,
                                    const float *rays_d, const float *rays_o, float depth_ratio, float lambda_normal,
                                    float lambda_dist, float *out2, void *workspace, size_t workspace_bytes,
                                    scorp_stream_t stream);
int scorp_gs2d_regularizers_backward(int32_t image_width, int32_t image_height, const float *allmap,
                                     const float *viewmatrix, const float *rays_d, const float *rays_o, float depth_ratio,
                                     float lambda_normal, float lambda_dist, const float *g_out2, float *g_allmap,
                                     scorp_stream_t stream);

/* ---- fused photometric loss (rows a8/a9 of the hot path) ----
 * loss = (1-lambda) * mean|x-y| + lambda * (1 - mean SSIM(x,y)), x = img*mask, y = gt*mask (mask [H,W] or NULL):
 * train_3dgs.py:106-107, post_refine_gs.py:103-111 over gs3dgs/utils/loss_utils.py:17-73 (11x11 Gaussian window,
 * sigma 1.5, zero padding, C1=1e-4, C2=9e-4).  img/gt are [C,H,W].  out_loss3 (device) = {loss, l1, ssim}.
 * The workspace carries the forward's derivative maps to the backward (read-only there). */
size_t scorp_loss_workspace_bytes(int32_t channels, int32_t height, int32_t width);
int scorp_loss_l1_ssim_forward(const float *img, const float *gt, const float *mask, int32_t channels, int32_t height,
                               int32_t width, float lambda_dssim, float *out_loss3, void *workspace,
                               size_t workspace_bytes, int32_t need_backward, scorp_stream_t stream);
/* grad_img[C,H,W] = grad_out[0] * d loss / d img (grad_out: device scalar, NULL = 1). */
int scorp_loss_l1_ssim_backward(const float *img, const float *gt, const float *mask, int32_t channels, int32_t height,
                                int32_t width, float lambda_dssim, const void *workspace, const float *grad_out,
                                float *grad_img, scorp_stream_t stream);

/* ---- one training view in one call (train_3dgs.py:88-150 minus the optimizer: render, photometric loss, backward) ----
 * Enqueues, on `stream` and without any host synchronisation, exactly the sequence a caller of the entry points above
 * would issue for one iteration of the reference's training loop:
 *   scorp_gs3d_preprocess -> scorp_gs3d_render -> scorp_gs3d_render_tail -> scorp_loss_l1_ssim_forward ->
 *   scorp_loss_l1_ssim_backward (upstream gradient 1) -> scorp_gs3d_backward (dL_ddepth = dL_dalpha = NULL).
 * It exists because at ~1 ms of device work per view the host side (Python glue, autograd bookkeeping, a dozen FFI
 * calls) is of the same size: one call keeps the GPU the bottleneck whatever the host is doing.  Every buffer is the
 * caller's, sized as for the separate calls; `capacity` pairs must have been reserved (check with
 * scorp_gs3d_check_overflow afterwards, as for scorp_gs3d_render without scorp_gs3d_num_pairs).
 * out_depth_raw is the rasterizer's un-normalised depth, out_depth = nan_to_num(depth / alpha) as render() returns
 * it; out_depth / out_visible may both be NULL (the tail is skipped then). */
/* Optional optimizer step INSIDE the view (train_3dgs.py:180-193 for the iterations that neither densify nor reset): the
 * per-Gaussian backward kernel holds every Gaussian's whole gradient row in registers / LDS and applies
 * torch.optim.Adam's update there - the arithmetic of scorp_adam_step_guarded, bit for bit - together with the view's share of
 * the densification statistics (scorp_densification_stats' arithmetic).  The 248-byte gradient row never travels to HBM
 * and back and the parameters are not read a second time: 932 instead of 1652 bytes of optimizer traffic per Gaussian.
 * Needs the raw-leaf convention (shs = _features_dc, shs_rest = _features_rest, raw_params = 7, scales + rotations): the
 * arrays of `in` ARE the optimizer's parameters and are updated in place by the view's last kernel.  Leaves in the order
 * xyz, features_dc, features_rest, opacity, scaling, rotation; exp_avg[k] == NULL: that leaf is frozen.  Nothing is updated
 * (and *skipped_counter is incremented, if given) when the view overflowed its pair reservation (the overflow word of
 * out_header, or of the state header) - the guard of scorp_adam_step_guarded, without the host in the loop. */
typedef struct ScorpFusedAdam {
  float *exp_avg[6];
  float *exp_avg_sq[6];
  float lr[6];
  float _pad[2];
  double beta1, beta2, eps;
  int32_t step;                  /* 1-based step count of this update (bias corrections) */
  int32_t _pad2;
  uint32_t *skipped_counter;     /* device word or NULL */
  float *max_radii2D;            /* [N] statistics, all three or none: updated for the visible Gaussians */
  float *xyz_gradient_accum;     /* [N] += |(dL/dmeans2D.x, dL/dmeans2D.y)| */
  float *denom;                  /* [N] += 1 */
} ScorpFusedAdam;

typedef struct ScorpGs3dTrainView {
  const ScorpGs3dInputs *in;
  int32_t *out_radii;            /* [N] */
  void *state;                   /* scorp_gs3d_state_bytes(), 256-byte aligned */
  size_t state_bytes;
  void *pairs;                   /* scorp_gs3d_pairs_bytes(capacity) */
  uint64_t capacity;
  float *out_color;              /* [3,H,W] */
  float *out_depth_raw;          /* [H,W] */
  float *out_alpha;              /* [H,W] */
  float *out_depth;              /* [H,W] or NULL */
  uint8_t *out_visible;          /* [N] bytes or NULL */
  const float *gt;               /* [3,H,W] */
  const float *mask;             /* [H,W] or NULL */
  float lambda_dssim;
  uint32_t backward_flags;       /* flags of scorp_gs3d_backward_ex for the view's backward (SCORP_BACKWARD_EXACT_FP32, ...); 0 = default */
  float *out_loss3;              /* device: {loss, l1, ssim} */
  void *loss_workspace;          /* scorp_loss_workspace_bytes(3, H, W) */
  size_t loss_workspace_bytes;
  float *grad_color;             /* [3,H,W] scratch: d loss / d color */
  const ScorpGs3dGrads *grads;   /* gradients w.r.t. the inputs, as for scorp_gs3d_backward */
  void *backward_scratch;        /* scorp_gs3d_backward_scratch_bytes(N) (scorp_gs3d_backward_scratch_bytes_ex with backward_flags) */
  size_t backward_scratch_bytes;
  uint32_t *out_header;          /* optional, 4 device words {pairs needed, overflow, capacity, 0}: the view's overflow word
                                    survives the state blob without a copy launch (the scatter kernel writes it) */
  const ScorpFusedAdam *adam;    /* optional: the optimizer step and the statistics inside the view (see ScorpFusedAdam) */
} ScorpGs3dTrainView;
int scorp_gs3d_train_view(const ScorpGs3dTrainView *view, scorp_stream_t stream);

/* The 2DGS twin: one iteration of train_2dgs.py:95-150 for the plain photometric loss plus its two regularisers,
 *   scorp_gs2d_preprocess -> scorp_gs2d_render -> scorp_loss_l1_ssim_forward -> scorp_gs2d_regularizers_forward ->
 *   scorp_loss_l1_ssim_backward -> scorp_gs2d_regularizers_backward -> scorp_gs2d_backward,
 * enqueued by one call; total loss = out_loss3[0] + out_reg2[0] + out_reg2[1].  With lambda_normal = lambda_dist = 0
 * (iterations <= 3000, train_2dgs.py:142-143) the regulariser kernels are skipped and no allmap gradient is formed.
 * rays_d[H*W,3] / rays_o[3]: the camera's ray table (see scorp_gs2d_maps_forward).  Every buffer is the caller's. */
typedef struct ScorpGs2dTrainView {
  const ScorpGs3dInputs *in;
  int32_t *out_radii;            /* [N] */
  void *state;                   /* scorp_gs2d_state_bytes(), 256-byte aligned */
  size_t state_bytes;
  void *pairs;                   /* scorp_gs3d_pairs_bytes(capacity) */
  uint64_t capacity;
  float *out_color;              /* [3,H,W] */
  float *out_allmap;             /* [7,H,W] */
  const float *gt;               /* [3,H,W] */
  const float *mask;             /* [H,W] or NULL */
  const float *rays_d;           /* [H*W,3] */
  const float *rays_o;           /* [3] */
  float lambda_dssim, depth_ratio, lambda_normal, lambda_dist;
  float *out_loss3;              /* device: {photometric loss, l1, ssim} */
  float *out_reg2;               /* device: {normal loss, distortion loss} (zeros if both lambdas are 0) */
  void *loss_workspace;          /* scorp_loss_workspace_bytes(3, H, W) */
  size_t loss_workspace_bytes;
  void *reg_workspace;           /* scorp_gs2d_regularizers_workspace_bytes(W, H) */
  size_t reg_workspace_bytes;
  float *grad_color;             /* [3,H,W] scratch */
  float *grad_allmap;            /* [7,H,W] scratch */
  const ScorpGs3dGrads *grads;
  void *backward_scratch;        /* scorp_gs2d_backward_scratch_bytes(N) (scorp_gs2d_backward_scratch_bytes_ex with backward_flags) */
  size_t backward_scratch_bytes;
  uint32_t backward_flags;       /* flags of scorp_gs2d_backward_ex for the view's backward; 0 = default */
  const ScorpFusedAdam *adam;    /* optional: the optimizer step and the statistics inside the view (see ScorpFusedAdam; the
                                    scaling leaf is [N,2], the statistic norms the whole means2D-gradient row) */
} ScorpGs2dTrainView;
int scorp_gs2d_train_view(const ScorpGs2dTrainView *view, scorp_stream_t stream);

/* ---- rigid / scale transform of a whole model incl. its SH coefficients (utils/gaussians.py:12-108) ----
 * In place, one launch:  xyz <- ((xyz - c) R^T) * s + c + t;  rotation <- q (x) normalize(rotation) (w,x,y,z; q = the
 * quaternion of R);  scaling <- scaling + log(s) (log-space, `scale_dims` = 3, or 2 for surfels);  features_rest
 * [N, rest_coeffs, 3]: band l = 1..3 (coefficients l^2-1 .. (l+1)^2-2 of it) multiplied by the real Wigner-D block D_l.
 * params: 113 device floats, 16-byte aligned: R[9] (row-major) c[3] t[3] s[3] q[4] D1[9] D2[25] D3[49].
 * rotation / scaling / features_rest may each be NULL: that part of the model is then left untouched (a translation
 * passes only xyz; the reference's gaussians_translate / gaussians_scale never touch the quaternions). */
int scorp_gaussians_transform(float *xyz, float *rotation, float *scaling, float *features_rest, int32_t num_gaussians,
                              int32_t rest_coeffs, int32_t scale_dims, const float *params, scorp_stream_t stream);

/* ---- simple_knn replacement ----
 * out[i] = mean of the squared distances from point i to its 3 nearest other points, as
 * `simple_knn._C.distCUDA2(points)` (gs3dgs/scene/gaussian_model.py:177).  xyz[N,3], out[N]. */
int scorp_knn_dist2(const float *xyz, int32_t num_points, float *out, scorp_stream_t stream);

/* ---- fused multi-tensor Adam (row a10) ----
 * One launch for all parameter groups of torch.optim.Adam(lr per group, betas, eps=1e-15, no weight decay)
 * (gs3dgs/scene/gaussian_model.py:197-206): m,v moments and parameters updated in place. `step` is the 1-based
 * step count used for the bias corrections. */
#define SCORP_ADAM_MAX_TENSORS 8
typedef struct ScorpAdamTensor {
  float *param;
  const float *grad;
  float *exp_avg;
  float *exp_avg_sq;
  uint64_t numel;
  float lr;
  float _pad;
} ScorpAdamTensor;
int scorp_adam_step(const ScorpAdamTensor *tensors, int32_t num_tensors, double beta1, double beta2, double eps,
                    int32_t step, scorp_stream_t stream);
/* The same step made conditional ON THE DEVICE: nothing is updated if *skip_if_nonzero != 0 when the kernel runs (NULL =
 * unconditional).  For training loops that reserve the pair buffer instead of synchronising for the pair count: the word
 * is the view's overflow flag (second 32-bit word of the forward state), so a view rendered from truncated tile lists
 * cannot move the parameters or the Adam moments - without a host round trip per iteration. */
int scorp_adam_step_guarded(const ScorpAdamTensor *tensors, int32_t num_tensors, double beta1, double beta2, double eps,
                            int32_t step, const uint32_t *skip_if_nonzero, scorp_stream_t stream);

/* The same; when the step is skipped, *skipped_counter (device word, NULL = not counted) is incremented by one - a training loop
 * reads it at its next synchronisation point and takes the skipped steps out of its bias-correction counter.  Replicas of a
 * data-parallel run skip on the all-reduced overflow word, so their counters agree (FusedAdam.take_skipped). */
int scorp_adam_step_guarded_ex(const ScorpAdamTensor *tensors, int32_t num_tensors, double beta1, double beta2, double eps,
                               int32_t step, const uint32_t *skip_if_nonzero, uint32_t *skipped_counter, scorp_stream_t stream);

/* ---- per-view densification statistics (row a11; train_3dgs.py:180-181, train_2dgs.py:189-190,
 * gs3dgs/scene/gaussian_model.py:603-605) ----
 * For every Gaussian i with visible[i] != 0:
 *     max_radii2D[i] = max(max_radii2D[i], (float)radii[i]);
 *     xyz_gradient_accum[i] += sqrt(g[0]^2 + g[1]^2),  g = grad_means2D + i * grad_stride;   denom[i] += 1.
 * One launch instead of the reference's three boolean-mask indexings (each a compaction plus a host synchronisation,
 * ~0.6 ms per iteration at 1 M Gaussians).  Nothing is touched if *skip_if_nonzero != 0 when the kernel runs (NULL =
 * unconditional): the overflow word of a view rendered with a reserved pair buffer, as for scorp_adam_step_guarded. */
int scorp_densification_stats(int32_t num_gaussians, const int32_t *radii, const uint8_t *visible, const float *grad_means2D,
                              int32_t grad_stride, const uint32_t *skip_if_nonzero, float *max_radii2D,
                              float *xyz_gradient_accum, float *denom, scorp_stream_t stream);
/* The same with the gradient norm taken over `norm_components` (2 or 3) leading floats of a row: the 2DGS model's
 * add_densification_stats norms the WHOLE means2D gradient row (gs2dgs/scene/gaussian_model.py:494-495), the 3DGS one
 * only x, y (gs3dgs/scene/gaussian_model.py:603-605).  scorp_densification_stats is this with norm_components = 2. */
int scorp_densification_stats_ex(int32_t num_gaussians, const int32_t *radii, const uint8_t *visible, const float *grad_means2D,
                                 int32_t grad_stride, int32_t norm_components, const uint32_t *skip_if_nonzero,
                                 float *max_radii2D, float *xyz_gradient_accum, float *denom, scorp_stream_t stream);

/* ---- densify / prune compaction (row f2 of the hot-path scope; gs3dgs/scene/gaussian_model.py:412-601) ----
 * Re-indexes up to SCORP_ROWS_MAX_TENSORS row-major float tensors in one launch: dst row j = src row
 * (src_index[j] & 0x7fffffff).  Bit 31 of an index marks a FRESH row (a cloned or split Gaussian): tensors with
 * zero_if_fresh != 0 (the Adam moments) get zeros there instead of a copy.  src and dst must not overlap. */
#define SCORP_ROWS_MAX_TENSORS 24
typedef struct ScorpRowTensor {
  const float *src;
  float *dst;
  uint32_t row_floats;
  uint32_t zero_if_fresh;
} ScorpRowTensor;
int scorp_gather_rows(const ScorpRowTensor *tensors, int32_t num_tensors, const int32_t *src_index, uint64_t num_out_rows,
                      scorp_stream_t stream);

/* ---- in-library kernel timing: hipEvent pairs recorded on the launch stream around every kernel ---- */
/* Off by default. scorp_prof_enable(1) clears the accumulators and starts recording; collect() synchronises the
 * recorded events and returns, per kernel id, the summed duration in ms and the number of launches. */
int scorp_prof_enable(int on);
/* Restrict the bracketing to the kernels whose bit (1 << kernel_id) is set (default: all) — an event pair costs a few
 * microseconds of stream time, so a throughput run brackets only the kernel it reports. */
int scorp_prof_select(uint64_t kernel_mask);
int scorp_prof_num_kernels(void);
const char *scorp_prof_kernel_name(int kernel_id);
int scorp_prof_collect(double *total_ms, uint64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* SCORP_GS_H */
