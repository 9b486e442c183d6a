// train_view.hip — scorp_gs3d_train_view: one training view (render, L1 + SSIM loss, backward) enqueued by ONE call.
// Host code only: it chains the library's own entry points, so the kernels, their order and their results are those
// of the separate calls (train_3dgs.py:88-150 minus the optimizer step).  See include/scorp_gs.h for the contract.
#include "common.hpp"

using namespace scorp;

extern "C" int scorp_gs3d_train_view(const ScorpGs3dTrainView *v, scorp_stream_t stream) {
  if (!v || !v->in || !v->grads) { set_error("NULL argument to scorp_gs3d_train_view"); return SCORP_ERR_INVALID; }
  if (!v->out_color || !v->out_depth_raw || !v->out_alpha || !v->out_radii || !v->gt || !v->out_loss3 || !v->grad_color) {
    set_error("scorp_gs3d_train_view: an output / ground-truth / scratch pointer is NULL");
    return SCORP_ERR_INVALID;
  }
  const ScorpGs3dInputs *in = v->in;
  const int W = in->image_width, H = in->image_height;
  if (int e = scorp_gs3d_preprocess(in, v->out_radii, v->state, v->state_bytes, stream)) return e;
  if (int e = scorp_gs3d_render(in, v->state, v->pairs, v->capacity, v->out_color, v->out_depth_raw, v->out_alpha, stream)) return e;
  if (v->out_depth && v->out_visible) {
    if (int e = scorp_gs3d_render_tail(v->out_depth_raw, v->out_alpha, (int64_t)W * H, v->out_radii, in->num_gaussians,
                                       v->out_depth, v->out_visible, stream)) return e;
  }
  if (int e = scorp_loss_l1_ssim_forward(v->out_color, v->gt, v->mask, 3, H, W, v->lambda_dssim, v->out_loss3,
                                         v->loss_workspace, v->loss_workspace_bytes, 1, stream)) return e;
  if (int e = scorp_loss_l1_ssim_backward(v->out_color, v->gt, v->mask, 3, H, W, v->lambda_dssim, v->loss_workspace, nullptr,
                                          v->grad_color, stream)) return e;
  return scorp_gs3d_backward(in, v->state, v->pairs, v->capacity, v->grad_color, nullptr, nullptr, v->grads,
                             v->backward_scratch, v->backward_scratch_bytes, stream);
}
