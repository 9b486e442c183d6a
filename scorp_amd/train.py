"""The build's counterparts of the reference's optimisation loops, on synthetic or caller-supplied views:

  training_iteration(...)  one iteration of train_3dgs.py:56-193 (LR schedule, SH-degree schedule, random background,
                           render, 0.8*L1 + 0.2*(1-SSIM), the depth terms and the isotropic regulariser after
                           depth_from_iter, backward, densification bookkeeping, Adam step); `surfels=True` with
                           render_fn=scorp_amd.renderer2d.render is train_2dgs.py's iteration (normal-consistency
                           and distortion regularisers on their 7000 / 3000 iteration schedule)
  post_refine(...)         the 800-iteration appearance refinement of post_refine_gs.py:30-203: everything frozen
                           but the colours, masked L1 + SSIM

  train(..., data_parallel=True)   the same loop on G ranks of ONE scene: rank r renders view
                           r of every group of G views, the per-Gaussian gradients are averaged (bucketed
                           all-reduce, RCCL over xGMI), and the densification statistics are reduced before every
                           densify step so that all replicas clone / split / prune identically

Dataset I/O, tensorboard and checkpoint plumbing of the scripts are out of scope; the loops take
cameras and ground-truth tensors that are already resident on the GPU.
"""
import random

import torch
import torch.distributed as dist

from .fused_loss import fused_l1_ssim_loss
from .rasterizer3d import PairPolicy
from .loss import depth_losses, isotropic_loss, psnr
from .parallel import average_gradients, average_gradients_sparse, collective, world
from .renderer import render


class PipelineParams:
    """Defaults of gs3dgs/arguments/__init__.py:67-72 + this build's fused fast path."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    fused_activations = True


def _sync_densification_stats(gaussians):
    """Before a densify / prune decision every replica must see the same statistics: sums of the screen-space gradient
    norms and visit counts, max of the screen radii."""
    if collective():
        dist.all_reduce(gaussians.xyz_gradient_accum, op=dist.ReduceOp.SUM)
        dist.all_reduce(gaussians.denom, op=dist.ReduceOp.SUM)
        dist.all_reduce(gaussians.max_radii2D, op=dist.ReduceOp.MAX)


def _shared_overflow(pkg, data_parallel):
    """Data-parallel replicas must discard an iteration TOGETHER: the overflow word of a fused view is per rank, and a rank
    that skipped its Adam step alone would stop being bit-identical to the others (next densification: different N,
    mismatched all-reduce shapes).  One 4-byte all-reduce(MAX) on the device word, no host synchronisation; the reduced
    word then guards the optimizer step and masks the statistics on every rank."""
    if data_parallel and collective():
        # (a clone: pkg["overflow"] is a view of the 64-byte header PairPolicy.drain() reads for THIS rank's bookkeeping -
        # reduced in place, a rank that did not overflow could read overflow = 1 with num_pairs <= capacity there)
        ovf = pkg["overflow"].clone()
        dist.all_reduce(ovf, op=dist.ReduceOp.MAX)
        return ovf
    return pkg["overflow"]


def training_iteration(gaussians, cam, gt_image, opt, pipe, background, iteration, scene_extent=4.0, densify=True,
                       render_fn=render, loss_fn=fused_l1_ssim_loss, data_parallel=False, gt_depth=None, gt_depth_est=None,
                       surfels=False, fused_view=False, white_background=False, sparse_gradients=False, fused_step=True,
                       view_fn=None):
    """Returns (loss tensor, render package). Mirrors train_3dgs.py:74-193 for one camera.  With `data_parallel` the
    caller hands each rank a different camera; gradients are averaged over ranks before the optimizer step and the
    densification statistics are reduced before they are used, so the replicas stay bit-identical.
    `fused_step` (with `fused_view`): let the view apply the optimizer step itself where that is the same computation (see
    below).  `view_fn`: the one-call view (default train_view.train_view; the CPU tests of the loop's bookkeeping pass a
    stand-in)."""
    stepped_in_view = False
    gaussians.update_learning_rate(iteration)
    if iteration % 1000 == 0:
        gaussians.oneupSHdegree()
    bg = torch.rand(3, device=background.device) if opt.random_background else background
    extra_terms = iteration > getattr(opt, "depth_from_iter", 1 << 30) and (
        gt_depth is not None or gt_depth_est is not None or getattr(opt, "lambda_isotropic", 0.0) > 0)
    if (fused_view and surfels and not extra_terms and loss_fn is fused_l1_ssim_loss and getattr(pipe, "fused_activations", False)
            and hasattr(gaussians, "raw_leaves")):
        # the 2DGS iteration (train_2dgs.py:95-150) by ONE library call: render + loss + regularisers + backward
        from .train_view import train_view2d
        lambda_normal = opt.lambda_normal if iteration > 7000 else 0.0
        lambda_dist = opt.lambda_dist if iteration > 3000 else 0.0
        kw_view = {}
        if _step_in_view_ok(gaussians, opt, iteration, densify, data_parallel, white_background, fused_step):
            kw_view["optimizer"] = gaussians.optimizer
            if densify and iteration < opt.densify_until_iter and _stats_components(gaussians) == 3:
                kw_view["stats"] = (gaussians.max_radii2D, gaussians.xyz_gradient_accum, gaussians.denom)
        pkg = train_view2d(cam, gaussians, pipe, bg, gt_image, opt.lambda_dssim, lambda_normal, lambda_dist, **kw_view)
        loss = pkg["loss"]
        stepped_in_view = bool(pkg.get("optimizer_stepped"))
        ovf = _shared_overflow(pkg, data_parallel)
        if hasattr(gaussians.optimizer, "skip_flag") and not stepped_in_view:
            gaussians.optimizer.skip_flag = ovf
        pkg["visibility_filter"] = pkg["visibility_filter"] & (ovf == 0)
    elif (fused_view and not surfels and not extra_terms and render_fn is render and loss_fn is fused_l1_ssim_loss
            and getattr(pipe, "fused_activations", False)):
        # plain photometric iteration: render + loss + backward enqueued by ONE library call (train_view.py); same
        # kernels and results, no autograd graph.  The pair buffer is reserved, the caller drains (see train()).
        view_fn = view_fn or _default_view_fn()
        # The optimizer step itself inside the view (ScorpFusedAdam: the per-Gaussian backward kernel applies Adam and the
        # view's statistics while it holds the gradient row) on the iterations where the reference's optimizer.step() sees
        # this view's gradients as they are: one replica (no averaging first) and neither a densification nor an opacity
        # reset in between - after those the reference's leaves are fresh tensors without .grad and its step() passes them by
        # (train_3dgs.py:183-193), which the separate path below reproduces.
        step_in_view = _step_in_view_ok(gaussians, opt, iteration, densify, data_parallel, white_background, fused_step)
        kw_view = {}
        if data_parallel and not sparse_gradients and not step_in_view:
            # the view writes its gradients straight into the flat arena the collective reduces in place (parallel.GradArena)
            arena = _grad_arena(gaussians)
            if arena is not None:
                kw_view["grad_out"] = arena.views
        if step_in_view:
            kw_view["optimizer"] = gaussians.optimizer
            if densify and iteration < opt.densify_until_iter and _stats_components(gaussians) == 2:
                kw_view["stats"] = (gaussians.max_radii2D, gaussians.xyz_gradient_accum, gaussians.denom)
        pkg = view_fn(cam, gaussians, pipe, bg, gt_image, opt.lambda_dssim, **kw_view)
        loss = pkg["loss"]
        stepped_in_view = bool(pkg.get("optimizer_stepped"))
        # The pair buffer was reserved, not sized from this view's count.  If the view overflowed it (device word
        # pkg["overflow"]), its gradients come from truncated tile lists: the optimizer step is skipped on the device and
        # the densification statistics below are masked - no host synchronisation, nothing wrong is ever applied.
        ovf = _shared_overflow(pkg, data_parallel)      # BEFORE the gradients are averaged and the flag is used
        if hasattr(gaussians.optimizer, "skip_flag") and not stepped_in_view:
            gaussians.optimizer.skip_flag = ovf
        pkg["visibility_filter"] = pkg["visibility_filter"] & (ovf == 0)
    else:
        pkg = render_fn(cam, gaussians, pipe, bg)
        loss = loss_fn(pkg["render"], gt_image, opt.lambda_dssim)
        # depth terms and the isotropic regulariser start together (train_3dgs.py:109-150, train_2dgs.py:95-140)
        if iteration > getattr(opt, "depth_from_iter", 1 << 30):
            if gt_depth is not None or gt_depth_est is not None:
                loss = loss + depth_losses(pkg["render_depth"], iteration, opt, gt_depth, gt_depth_est)
            if getattr(opt, "lambda_isotropic", 0.0) > 0:
                loss = loss + opt.lambda_isotropic * isotropic_loss(gaussians.get_scaling)
        if surfels:   # train_2dgs.py:142-150: normal consistency after 7000 iterations, depth distortion after 3000
            from .renderer2d import fused_surfel_regularizers, surfel_regularizers
            lambda_normal = opt.lambda_normal if iteration > 7000 else 0.0
            lambda_dist = opt.lambda_dist if iteration > 3000 else 0.0
            if lambda_normal > 0 or lambda_dist > 0:
                reg = fused_surfel_regularizers if getattr(pkg, "allmap", None) is not None else surfel_regularizers
                normal_loss, dist_loss = reg(pkg, lambda_normal, lambda_dist)
                loss = loss + normal_loss + dist_loss
        loss.backward()
    with torch.no_grad():
        if data_parallel:
            ps = [g["params"][0] for g in gaussians.optimizer.param_groups]
            # Only the rows some rank rendered travel (parallel.average_gradients_sparse) - valid only while every loss term
            # flows through the rasterizer: the isotropic regulariser gives EVERY Gaussian a scaling gradient, a
            # caller-supplied render / loss function may do anything, and then the dense average is the right one.
            rasterizer_only = not extra_terms and loss_fn is fused_l1_ssim_loss and (render_fn is render or surfels)
            arena = getattr(gaussians, "_grad_arena_obj", None)
            if sparse_gradients and rasterizer_only:
                average_gradients_sparse(ps, pkg["radii"] > 0)    # (the view's own visibility: not masked by an overflow)
            elif arena is not None and fused_view and all(p.grad is None or (v is not None and p.grad.data_ptr() == v.data_ptr())
                                                          for p, v in zip(arena.params, arena.views)):
                arena.attach()        # (a leaf nothing was written for: zeros)
                arena.average()       # in place, two collectives back to back, one wait
            else:
                average_gradients(ps)
        if densify and iteration < opt.densify_until_iter:
            vis, radii = pkg["visibility_filter"], pkg["radii"]
            if not pkg.get("stats_accumulated"):     # (the fused step's kernel has done it)
                gaussians.accumulate_view_stats(pkg["viewspace_points"], vis, radii)   # train_3dgs.py:180-181, no mask indexing
            if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
                size_threshold = opt.max_screen_size if iteration > opt.opacity_reset_interval else None   # train_3dgs.py:184
                if data_parallel:
                    _sync_densification_stats(gaussians)
                    torch.manual_seed(1_000_003 * iteration)   # densify_and_split samples positions: same draw on every rank
                gaussians.densify_and_prune(opt.densify_grad_threshold, opt.opacity_cull, scene_extent, size_threshold)
                if fused_view:
                    # the reservation context keeps pairs PER GAUSSIAN, so the resized model starts from a scaled
                    # reservation; drain now so that it is also the freshest figure (one event wait per 100 iterations)
                    _drain_reservation(optimizer=gaussians.optimizer)
            if iteration % opt.opacity_reset_interval == 0 or (white_background and iteration == opt.densify_from_iter):
                gaussians.reset_opacity()   # train_3dgs.py:187-188
        if not stepped_in_view:
            gaussians.optimizer.step()
        gaussians.optimizer.zero_grad(set_to_none=True)
    return loss, pkg


def _stats_components(gaussians):
    """Over how many components of a means2D-gradient row the model's add_densification_stats norms - declared next to the
    method by the class that DEFINES it (GaussianModel.accumulate_view_stats uses the same rule); None: an override the fused
    statistics do not restate."""
    owner = next((c for c in type(gaussians).__mro__ if "add_densification_stats" in c.__dict__), None)
    return None if owner is None else owner.__dict__.get("_stats_norm_components")


def _step_in_view_ok(gaussians, opt, iteration, densify, data_parallel, white_background, fused_step):
    """May this iteration's optimizer step run inside the one-call view?  One replica (no averaging first), a FusedAdam, and
    neither a densification nor an opacity reset between backward() and step() (train_3dgs.py:183-193)."""
    will_densify = densify and iteration < opt.densify_until_iter and iteration > opt.densify_from_iter and \
        iteration % opt.densification_interval == 0
    will_reset = densify and iteration < opt.densify_until_iter and (
        iteration % opt.opacity_reset_interval == 0 or (white_background and iteration == opt.densify_from_iter))
    return bool(fused_step and not data_parallel and not will_densify and not will_reset and
                hasattr(gaussians.optimizer, "fused_view_pack"))


def _grad_arena(gaussians):
    """The model's persistent gradient arena (leaves in the one-call view's order), rebuilt when the model was resized."""
    from .parallel import GradArena
    if not hasattr(gaussians, "raw_leaves"):
        return None
    leaves = [gaussians.get_xyz] + list(gaussians.raw_leaves())
    arena = getattr(gaussians, "_grad_arena_obj", None)
    if arena is None or not arena.matches(leaves) or any(a is not b for a, b in zip(arena.params, leaves)):
        arena = GradArena(leaves)
        gaussians._grad_arena_obj = arena
    return arena


def _default_view_fn():
    from .train_view import train_view
    return train_view


def train(gaussians, cameras, gt_images, opt, pipe=None, iterations=None, background=None, seed=0, data_parallel=False,
          **kw):
    """Runs `iterations` training iterations over shuffled cameras; returns the list of per-iteration losses.
    `data_parallel`: every rank runs this with the same arguments and the same initial model; one iteration then
    consumes world_size views (rank r takes the r-th of each group), i.e. a batch of views per optimizer step."""
    pipe = pipe or PipelineParams()
    dev = gaussians.get_xyz.device
    background = torch.zeros(3, device=dev) if background is None else background
    if gaussians.optimizer is None:
        gaussians.training_setup(opt)
    rank, w = world() if data_parallel else (0, 1)
    rng = random.Random(seed)            # same seed on every rank: identical camera order
    stack, losses = [], []
    first_checked, retries = False, 0
    it = 1
    while it <= (iterations or opt.iterations):
        ks = []
        for _ in range(w):
            if not stack:
                stack = list(range(len(cameras)))
                rng.shuffle(stack)
            ks.append(stack.pop())
        k = ks[rank]
        loss, _ = training_iteration(gaussians, cameras[k], gt_images[k], opt, pipe, background, it,
                                     data_parallel=data_parallel and collective(), **kw)
        if kw.get("fused_view") and not first_checked:
            # The FIRST fused view of a run sizes the reservation: verified at once (one host synchronisation per run).
            # If it overflowed it was discarded on the device and the reservation has grown: the same views run again,
            # instead of a context's default being wrong for every iteration up to the first periodic drain.
            retries += 1
            # (data-parallel replicas take the same decision: the skipped-step count they read is that of the all-reduced word)
            first_checked = _drain_reservation(quiet=retries < 3, optimizer=gaussians.optimizer) or retries >= 3
            if not first_checked:
                stack.extend(reversed(ks))
                continue
        losses.append(loss.detach())
        if kw.get("fused_view") and it % 32 == 0:
            _drain_reservation(optimizer=gaussians.optimizer)
        it += 1
    if kw.get("fused_view"):
        _drain_reservation(optimizer=gaussians.optimizer)
    return _to_floats(losses)


def _to_floats(losses):
    """The per-iteration loss tensors as Python floats with ONE device-to-host copy (float(l) per element is a
    synchronisation per iteration: 20 us each, as much as a whole short refinement's launch overhead)."""
    if not losses:
        return []
    return torch.stack([l.reshape(()) for l in losses]).tolist()


def _drain_reservation(quiet=False, optimizer=None):
    """Fused views reserve their pair buffer instead of asking for the count.  A view that overflowed its reservation was
    blended from truncated tile lists (nothing is written out of bounds) and was DISCARDED on the device: its optimizer
    step was skipped and it did not enter the densification statistics (training_iteration).  drain() has grown the
    reservation; training goes on, the user is told how that happened.  Returns False if a view had overflowed.

    `optimizer`: FusedAdam counted a bias-correction step on the host for every view, the discarded ones included; they
    are taken back here (rollback_steps), so that the updates after a retry are scaled as those of a run that never
    overflowed (a first-view overflow is an expected event: the default reservation is four pairs per Gaussian)."""
    from .rasterizer3d import PairOverflow
    ok, msg = True, None
    try:
        PairPolicy.drain()
    except PairOverflow as e:      # (only the reservation's own report: any other error propagates)
        ok, msg = False, str(e)
    # How many steps to take back is read from the optimizer's OWN device counter (FusedAdam.take_skipped: incremented by the
    # guarded step where its guard word was set) - not from the rank-local list of pending views: it counts this optimizer's
    # steps and nothing else (evaluation renders, other models' views are not in it), and in a data-parallel run the guard
    # is the all-reduced overflow word, so every replica takes back the same number at the same drain and their bias
    # corrections stay equal (round 5 rolled back on the rank that overflowed only: replicas diverged by one step).
    if optimizer is not None and hasattr(optimizer, "take_skipped"):
        n = optimizer.take_skipped()
        if n:
            optimizer.rollback_steps(n)
            ok = False
            msg = msg or f"{n} optimizer step(s) were skipped on the device because a replica's view overflowed its pair reservation"
    elif optimizer is not None and hasattr(optimizer, "rollback_steps") and not ok:
        optimizer.rollback_steps(1)
    if not ok and not quiet:
        import warnings
        warnings.warn(f"train(fused_view=True): {msg}; the overflowed views were skipped (no optimizer step, no statistics)")
    return ok


def post_refine(gaussians, cameras, gt_images, gt_alphas, opt, iterations=800, pipe=None, background=None, seed=0,
                fused_view=None):
    """post_refine_gs.py: colours only (`_opacity/_rotation/_scaling/_xyz` frozen, :53-56), SH degree 0 (:47),
    loss on image*mask vs gt*mask (:103-111).  Returns per-iteration losses.

    `fused_view` (default: whenever the model lives on the GPU with raw leaves and the pipe has fused activations): an
    iteration is ONE library call - the masked one-call view (train_view.train_view(mask=...): render + masked L1 / SSIM
    + the colour-only replay of the backward) followed by the guarded FusedAdam step - instead of render() + fused loss
    + autograd: the same kernels, ~10 % less host and launch time per iteration."""
    assert gaussians.max_sh_degree == 0, "post-refinement runs on SH-0 objects (post_refine_gs.py:47)"
    pipe = pipe or PipelineParams()
    dev = gaussians.get_xyz.device
    background = torch.zeros(3, device=dev) if background is None else background
    if gaussians.optimizer is None:
        gaussians.training_setup(opt)
    for name in ("_opacity", "_rotation", "_scaling", "_xyz"):
        gaussians.set_freeze(name, True)
    if fused_view is None:
        fused_view = dev.type == "cuda" and hasattr(gaussians, "raw_leaves") and getattr(pipe, "fused_activations", False)
    rng = random.Random(seed)
    stack, losses = [], []
    sized, retries, it = False, 0, 1
    while it <= iterations:
        if not stack:
            stack = list(range(len(cameras)))
            rng.shuffle(stack)
        k = stack.pop()
        if fused_view:
            from .train_view import train_view
            pkg = train_view(cameras[k], gaussians, pipe, background, gt_images[k], opt.lambda_dssim, mask=gt_alphas[k])
            loss = pkg["loss"]
            if hasattr(gaussians.optimizer, "skip_flag"):      # a view that overflowed its pair reservation moves nothing
                gaussians.optimizer.skip_flag = pkg["overflow"]
        else:
            pkg = render(cameras[k], gaussians, pipe, background)
            loss = fused_l1_ssim_loss(pkg["render"], gt_images[k], opt.lambda_dssim, mask=gt_alphas[k])
            loss.backward()
        with torch.no_grad():
            gaussians.optimizer.step()
            gaussians.optimizer.zero_grad(set_to_none=True)
        if fused_view and not sized:
            # the first one-call view sizes the pair reservation (see train()): checked at once, run again if it overflowed
            # (objects that fill the screen need more than the default four pairs per Gaussian)
            retries += 1
            sized = _drain_reservation(quiet=retries < 3, optimizer=gaussians.optimizer) or retries >= 3
            if not sized:
                stack.append(k)
                continue
        losses.append(loss.detach())
        if fused_view and it % 32 == 0:
            _drain_reservation(optimizer=gaussians.optimizer)
        it += 1
    if fused_view:
        _drain_reservation(optimizer=gaussians.optimizer)
    return _to_floats(losses)


def post_refine_objects(objects, cameras, gt_images, object_alphas, opt, iterations=800, pipe=None, background=None, seed=0,
                        refine_fn=None):
    """BASELINE config #4, "4 objects in parallel on 4 GPUs": the 800-iteration appearance refinement with the OBJECT as
    the unit of sharding - object j -> rank j mod G, every rank refines its objects against the per-object
    masks (`masked_image_rgba/<prompt>/`, segmentation_2d.py:79), and ONE fixed-size all-gather hands every rank the
    refined `_features_dc` of every object (what post_refine_gs.py:197-202 writes out per object).

    This is NOT the reference's computation wherever objects overlap on screen: post_refine_gs.py:40-50,103-111 trains the
    concatenation of all objects against the UNION mask, so an object seen through or in front of another one receives
    gradient from the blend of both; refined separately, each object only explains its own mask.  The two agree exactly
    where the screen footprints are disjoint (tests/test_parallel_cpu.py); use `post_refine` on the merged model - one
    GPU - when they are not.

    `objects`: list of SH-0 GaussianModels (identical on every rank); `object_alphas[j][k]`: mask of object j in view k.
    Returns the per-iteration losses of this rank's objects ({j: [...]}); every object's `_features_dc` is updated in
    place on every rank."""
    from .parallel import gather_rows, shard_indices
    refine_fn = refine_fn or post_refine
    n = len(objects)
    mine = shard_indices(n)
    losses, err = {}, None
    try:
        for j in mine:
            losses[j] = refine_fn(objects[j], cameras, gt_images, object_alphas[j], opt, iterations=iterations, pipe=pipe,
                                  background=background, seed=seed + j)
    except Exception as e:   # noqa: BLE001   (raised on every rank below, not only here: nobody waits in the all-gather)
        err = e
    from .parallel import all_ok
    if not all_ok(err is None, objects[0]._features_dc.device):
        raise RuntimeError("post_refine_objects: refinement failed on " + ("this rank" if err is not None else "another rank")) from err
    widths = [int(o._features_dc.shape[0]) for o in objects]
    rows = {j: objects[j]._features_dc.detach().reshape(widths[j], -1) for j in mine}
    # (a rank without an object - more ranks than objects - still takes part in the collective)
    gathered = gather_rows(rows, n, widths, c=objects[0]._features_dc[0].numel(), device=objects[0]._features_dc.device)
    with torch.no_grad():
        for j in range(n):
            if j not in losses:
                objects[j]._features_dc.data.copy_(gathered[j].to(objects[j]._features_dc).view_as(objects[j]._features_dc))
    return losses


@torch.no_grad()
def evaluate_psnr(gaussians, cameras, gt_images, pipe=None, background=None):
    pipe = pipe or PipelineParams()
    background = torch.zeros(3, device=gaussians.get_xyz.device) if background is None else background
    vals = [psnr(render(c, gaussians, pipe, background)["render"].clamp(0, 1), g).mean() for c, g in zip(cameras, gt_images)]
    return float(torch.stack(vals).mean())


@torch.no_grad()
def render_views_gt(gaussians, cameras, with_alpha=False, pipe=None, background=None):
    """Ground truth for synthetic experiments: renders of a reference model (clamped to [0,1])."""
    pipe = pipe or PipelineParams()
    background = torch.zeros(3, device=gaussians.get_xyz.device) if background is None else background
    pk = [render(c, gaussians, pipe, background) for c in cameras]
    imgs = [p["render"].clamp(0, 1).clone() for p in pk]
    return (imgs, [p["render_alpha"].clone() for p in pk]) if with_alpha else imgs
