// train_view.hip — scorp_gs3d_train_view: one training view (render, L1 + SSIM loss, backward) enqueued by ONE call.
// Host code only: it chains the library's own entry points, so the kernels, their order and their results are those
// of the separate calls (train_3dgs.py:88-150 minus the optimizer step).  See include/scorp_gs.h for the contract.
#include <math.h>

#include "common.hpp"

using namespace scorp;

// ScorpFusedAdam -> what the per-Gaussian kernels take (floats formed as adam_kernel forms them).  `skip`: the view's overflow word.
static int make_adam_epi(const ScorpFusedAdam *fa, const ScorpGs3dInputs *in, int scale_dims, const uint32_t *skip, AdamEpi *out) {
  AdamEpi ad;
  memset(&ad, 0, sizeof(ad));
  *out = ad;
  if (!fa) return SCORP_OK;
  (void)scale_dims;
  if (!in->shs || !in->shs_rest || in->raw_params != 7 || !in->scales || !in->rotations || in->sh_coeffs < 2 || fa->step < 1) {
    set_error("the fused optimizer step needs the raw-leaf convention (shs + shs_rest, raw_params = 7, scales + rotations) and step >= 1");
    return SCORP_ERR_INVALID;
  }
  const bool stats = fa->max_radii2D || fa->xyz_gradient_accum || fa->denom;
  if (stats && !(fa->max_radii2D && fa->xyz_gradient_accum && fa->denom)) {
    set_error("the fused optimizer step: give all three statistics arrays or none"); return SCORP_ERR_INVALID;
  }
  uintptr_t al = 0;
  for (int k = 0; k < 6; k++) {
    if ((fa->exp_avg[k] == nullptr) != (fa->exp_avg_sq[k] == nullptr)) { set_error("the fused optimizer step: exp_avg / exp_avg_sq of leaf %d", k); return SCORP_ERR_INVALID; }
    al |= (uintptr_t)fa->exp_avg[k] | (uintptr_t)fa->exp_avg_sq[k];
    ad.m[k] = fa->exp_avg[k]; ad.v[k] = fa->exp_avg_sq[k];
  }
  if (al & 15) { set_error("the fused optimizer step: Adam moments must be 16-byte aligned"); return SCORP_ERR_INVALID; }
  const double bc1 = 1.0 - pow(fa->beta1, (double)fa->step), bc2 = 1.0 - pow(fa->beta2, (double)fa->step);
  for (int k = 0; k < 6; k++) ad.step_size[k] = fa->lr[k] / (float)bc1;     // (float / float, as adam_kernel forms it)
  ad.on = 1;
  ad.omb1 = (float)(1.0 - fa->beta1); ad.beta2 = (float)fa->beta2; ad.omb2 = (float)(1.0 - fa->beta2); ad.eps = (float)fa->eps;
  ad.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  ad.skip = skip;
  ad.skipped_counter = fa->skipped_counter;
  ad.max_radii2D = fa->max_radii2D; ad.accum = fa->xyz_gradient_accum; ad.denom = fa->denom;
  *out = ad;
  return SCORP_OK;
}

extern "C" int scorp_gs3d_train_view(const ScorpGs3dTrainView *v, scorp_stream_t stream) {
  if (!v || !v->in || !v->grads) { set_error("NULL argument to scorp_gs3d_train_view"); return SCORP_ERR_INVALID; }
  if (!v->out_color || !v->out_depth_raw || !v->out_alpha || !v->out_radii || !v->gt || !v->out_loss3 || !v->grad_color) {
    set_error("scorp_gs3d_train_view: an output / ground-truth / scratch pointer is NULL");
    return SCORP_ERR_INVALID;
  }
  const ScorpGs3dInputs *in = v->in;
  const int W = in->image_width, H = in->image_height;
  // render()'s tail (normalised depth, visibility filter) comes out of the per-Gaussian kernel and the blend forward's
  // epilogue: same values as scorp_gs3d_render_tail, one launch less per view
  const bool tail = v->out_depth && v->out_visible;
  if (int e = preprocess3d_impl(in, v->out_radii, tail ? v->out_visible : nullptr, v->state, v->state_bytes, stream)) return e;
  // ... and so do the zeros of the backward's accumulator rows (N x 64 bytes at the head of its scratch)
  const size_t acc_bytes = (size_t)(in->num_gaussians > 0 ? in->num_gaussians : 0) * kAccStride * sizeof(float);
  const bool zero_here = v->backward_scratch && acc_bytes > 0 && acc_bytes <= v->backward_scratch_bytes &&
                         ((uintptr_t)v->backward_scratch & 15) == 0;
  if (int e = render3d_impl(in, v->state, v->pairs, v->capacity, v->out_color, v->out_depth_raw, v->out_alpha,
                            tail ? v->out_depth : nullptr, zero_here ? v->backward_scratch : nullptr, acc_bytes, stream, true,
                            v->out_header)) return e;
  // (the three loss values are written by the loss backward's launch: no finalize kernel between the two)
  if (int e = loss_forward_impl(v->out_color, v->gt, v->mask, 3, H, W, v->lambda_dssim, v->out_loss3, v->loss_workspace,
                                v->loss_workspace_bytes, 1, false, (hipStream_t)stream)) return e;
  if (int e = loss_backward_impl(v->out_color, v->gt, v->mask, 3, H, W, v->lambda_dssim, v->loss_workspace, nullptr,
                                 v->grad_color, v->out_loss3, (hipStream_t)stream)) return e;
  AdamEpi ad;
  if (int e = make_adam_epi(v->adam, in, 3, v->out_header ? v->out_header + 1 : reinterpret_cast<const uint32_t *>(v->state) + 1, &ad)) return e;
  return backward3d_impl(in, v->state, v->pairs, v->capacity, v->grad_color, nullptr, nullptr, v->grads,
                         v->backward_scratch, v->backward_scratch_bytes,
                         (v->backward_flags & ~SCORP_BACKWARD_SCRATCH_ZEROED) | (zero_here ? SCORP_BACKWARD_SCRATCH_ZEROED : 0u),
                         stream, v->adam ? &ad : nullptr);
}

extern "C" int scorp_gs2d_train_view(const ScorpGs2dTrainView *v, scorp_stream_t stream) {
  if (!v || !v->in || !v->grads) { set_error("NULL argument to scorp_gs2d_train_view"); return SCORP_ERR_INVALID; }
  if (!v->out_color || !v->out_allmap || !v->out_radii || !v->gt || !v->out_loss3 || !v->out_reg2 || !v->grad_color) {
    set_error("scorp_gs2d_train_view: an output / ground-truth / scratch pointer is NULL");
    return SCORP_ERR_INVALID;
  }
  const ScorpGs3dInputs *in = v->in;
  const int W = in->image_width, H = in->image_height;
  const bool reg = v->lambda_normal != 0.0f || v->lambda_dist != 0.0f;
  if (reg && (!v->rays_d || !v->rays_o || !v->grad_allmap || !v->reg_workspace)) {
    set_error("scorp_gs2d_train_view: the regularisers need rays_d, rays_o, grad_allmap and reg_workspace");
    return SCORP_ERR_INVALID;
  }
  hipStream_t hs = (hipStream_t)stream;
  if (int e = scorp_gs2d_preprocess(in, v->out_radii, v->state, v->state_bytes, stream)) return e;
  if (int e = scorp_gs2d_render(in, v->state, v->pairs, v->capacity, v->out_color, v->out_allmap, stream)) return e;
  if (int e = loss_forward_impl(v->out_color, v->gt, v->mask, 3, H, W, v->lambda_dssim, v->out_loss3, v->loss_workspace,
                                v->loss_workspace_bytes, 1, false, hs)) return e;
  if (reg) {
    if (int e = scorp_gs2d_regularizers_forward(W, H, v->out_allmap, in->viewmatrix, v->rays_d, v->rays_o, v->depth_ratio,
                                                v->lambda_normal, v->lambda_dist, v->out_reg2, v->reg_workspace,
                                                v->reg_workspace_bytes, stream)) return e;
  } else {
    SCORP_HIP_CHECK(hipMemsetAsync(v->out_reg2, 0, 2 * sizeof(float), hs));
  }
  if (int e = loss_backward_impl(v->out_color, v->gt, v->mask, 3, H, W, v->lambda_dssim, v->loss_workspace, nullptr,
                                 v->grad_color, v->out_loss3, hs)) return e;
  if (reg) {
    if (int e = scorp_gs2d_regularizers_backward(W, H, v->out_allmap, in->viewmatrix, v->rays_d, v->rays_o, v->depth_ratio,
                                                 v->lambda_normal, v->lambda_dist, nullptr, v->grad_allmap, stream)) return e;
  }
  AdamEpi ad;
  if (int e = make_adam_epi(v->adam, in, 2, reinterpret_cast<const uint32_t *>(v->state) + 1, &ad)) return e;
  return backward2d_impl(in, v->state, v->pairs, v->capacity, v->grad_color, reg ? v->grad_allmap : nullptr, v->grads,
                         v->backward_scratch, v->backward_scratch_bytes, v->backward_flags & ~SCORP_BACKWARD_SCRATCH_ZEROED,
                         stream, v->adam ? &ad : nullptr);
}
