"""Pose-hypothesis scoring on the rasterizer — the GPU form of the rotation sweep of config #3.

The reference feeds its `rotations_N.npz` initialisations to Open3D ICP on the CPU
(align_3dgs_clpe_9dof.py:80-111) and keeps the arg-max fitness; rendering is only used afterwards.  Here each
hypothesis is scored by rendering the (SH0) object under the rotation from a handful of cameras, forward only, and
comparing alpha + depth with the cached target renders (the role of the original-scene renders at :336-368).
Hypotheses are independent, so they shard over ranks (scorp_amd.parallel.sweep) with one gather at the end.
"""
import contextlib
import copy
import logging

import torch

from .rasterizer3d import PairPolicy
from .renderer import render
from .transforms import gaussians_rotate


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    fused_activations = True


@torch.no_grad()
def render_views(model, cameras, bg, render_fn=render):
    return [render_fn(c, model, _Pipe(), bg) for c in cameras]


class _RawPipe(_Pipe):
    raw_outputs = True      # renderer.render returns the un-normalised depth and skips the per-view tail launch


@torch.no_grad()
def hypothesis_fitness(model, R, cameras, targets, bg, render_fn=render):
    """Higher is better: minus the mean |alpha - alpha*| + |depth - depth*| over the cameras.  With the HIP
    rasterizer the comparison of a view is ONE launch on the rasterizer's raw outputs (`scorp_gs3d_pose_score_accumulate`:
    depth normalisation, both differences, both means and the running sum) instead of a tail launch and ~11 torch
    kernels, which were 40 % of a 100k-Gaussian 800x800 view's time."""
    m = copy.copy(model)
    m._xyz, m._rotation = model._xyz.detach().clone(), model._rotation.detach().clone()
    m._features_rest = model._features_rest.detach().clone()
    gaussians_rotate(m, R, fix_center=True)
    if render_fn is render and m._xyz.is_cuda:
        import ctypes
        from . import _C
        from .rasterizer3d import _stream
        L = _C.lib()
        acc = torch.zeros(1, dtype=torch.float32, device=m._xyz.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        for cam, tgt in zip(cameras, targets):
            out = render(cam, m, _RawPipe(), bg)
            d, a = out["render_depth_raw"], out["render_alpha"]
            td, ta = tgt["render_depth"].contiguous(), tgt["render_alpha"].contiguous()
            n = a.numel()
            _C.check(L.scorp_gs3d_pose_score_accumulate(p(d), p(a), p(td), p(ta), n, 1.0 / (n * len(cameras)), p(acc), _stream()),
                     "scorp_gs3d_pose_score_accumulate")
        return -acc[0]
    err = 0.0
    for cam, tgt in zip(cameras, targets):
        out = render_fn(cam, m, _Pipe(), bg)
        err = err + (out["render_alpha"] - tgt["render_alpha"]).abs().mean() + \
            (out["render_depth"] - tgt["render_depth"]).abs().mean()
    return -(err / len(cameras))


log = logging.getLogger("scorp_amd.align")


def _stacked_ok(model, cameras, render_fn):
    """The camera-side form applies to SH-0 objects (view-independent colour: what align loads as `gaussian_generated`,
    align_3dgs_clpe_9dof.py:307-308) rendered by the HIP rasterizer from cameras of one resolution."""
    if render_fn is not render or not model._xyz.is_cuda or getattr(model, "max_sh_degree", 1) != 0:
        return False
    c0 = cameras[0]
    return all(tuple(c.resolution) == tuple(c0.resolution) and c.FoVx == c0.FoVx and c.FoVy == c0.FoVy for c in cameras) \
        and int(c0.resolution[1]) % 16 == 0


class StackedSweep:
    """The rotation sweep for an SH-0 object WITHOUT touching the object: a hypothesis x -> R (x - c) + c is scored by
    moving the V cameras the other way (multiview.ViewStack.moved; tests/test_aux_gpu.py proves that rotating object and
    camera together is invisible) and rendering all V views as ONE stacked image (ScorpGs3dInputs.num_views): per
    hypothesis one preprocess -> bin -> sort -> blend launch set of training-frame size plus ONE score launch, instead
    of a model clone + ~20 torch kernels + V x 7 latency-bound launches.  The camera matrices of every hypothesis are
    formed by three batched matmuls up front.

    `batch` hypotheses share a launch set (round 6): their batch x V views are one stacked image.  Measured before it was
    built (scripts/dev/time_sweep_batch.py, S4: 100 k Gaussians, 15 x 800x800): 567 us per hypothesis alone, 504 in pairs,
    577 in fours (beyond 8 192 cells the stacked image leaves the two-level binning) - the binning kernels gain 15 - 25 %
    from the doubled launch, the blend, which is 300 of the 567 us and bound by its own arithmetic, 4 %.  The sweep is not a
    chain of latency-bound launches any more; it costs what its 5.9 M (tile, splat) pairs per hypothesis cost.  With the
    scoring form of the blend (scorp_gs3d_render_score: no colour, no images, no score launch) a whole sweep of 128 runs at
    1 830 / 1 965 / 2 001 hypotheses per second for batch = 1 / 2 / 3 (scripts/dev/prof_sweep.py; round 5: 1 580)."""

    def __init__(self, model, cameras, targets, bg, batch=None):
        from .multiview import ViewStack
        self.model, self.bg = model, bg
        self.dev = model._xyz.device
        self.stack = ViewStack(cameras, self.dev)
        # targets in the stacked layout (normalised depth, alpha), 16-byte aligned rows of the score kernel
        self.t_depth = torch.cat([t["render_depth"].reshape(self.stack.H, self.stack.W) for t in targets]).contiguous()
        self.t_alpha = torch.cat([t["render_alpha"].reshape(self.stack.H, self.stack.W) for t in targets]).contiguous()
        self.centre = model._xyz.detach().mean(0)
        # hypotheses per launch set: as many as keep the stacked image on the two-level binning (common.hpp: kMaxCells = 8192
        # cells of 64 x 64 pixels), at most three (three were 2 % faster than two; four leave the two-level binning at 800x800)
        cells = -(-self.stack.W // 64) * -(-(self.stack.V * self.stack.H) // 64)
        self.batch = max(1, min(3, 8192 // max(cells, 1))) if batch is None else max(1, int(batch))
        self._stacks = {1: self.stack}
        # sizing pass (setup): the exact pair count of the unrotated object; the sweep reserves twice that per hypothesis
        from .multiview import render_stacked
        prev = PairPolicy.mode
        PairPolicy.mode = "exact"
        try:
            self.reserve = 2 * render_stacked(model, self.stack, bg)["num_pairs"] + 4096
        finally:
            PairPolicy.mode = prev

    def _stack_of(self, h):
        """The ViewStack of h hypotheses' views (h x V cameras; only its sizes are used, the matrices are replaced per call)."""
        if h not in self._stacks:
            import copy as _copy
            st = _copy.copy(self.stack)
            st.V = h * self.stack.V
            st.view, st.proj, st.campos = (t.repeat((h,) + (1,) * (t.dim() - 1)) for t in (self.stack.view, self.stack.proj, self.stack.campos))
            self._stacks[h] = st
        return self._stacks[h]

    def _score_all(self, view, proj, campos, acc):
        """Enqueue every hypothesis (self.batch per launch set); acc[k] += the mismatch of hypothesis k."""
        # One launch set per `batch` hypotheses, ending in the SCORING form of the blend (scorp_gs3d_render_score, round 6):
        # the comparison with the target happens where depth and alpha are formed - no image leaves the kernel, no colour is
        # accumulated for a score that does not look at it, and the score launch (it read 150 MB per hypothesis) is gone.
        from .multiview import score_stacked
        n = self.t_alpha.numel()
        nh, V = view.shape[0], self.stack.V
        k = 0
        while k < nh:
            h = min(self.batch, nh - k)
            score_stacked(self.model, self._stack_of(h), self.bg, view[k:k + h].reshape(h * V, 4, 4), proj[k:k + h].reshape(h * V, 4, 4),
                          campos[k:k + h].reshape(h * V, 3), self.t_depth, self.t_alpha, acc[k:k + h], V * self.stack.H, 1.0 / n)
            k += h

    def score(self, rotations, ids):
        import numpy as np
        if len(ids) == 0:
            return []
        R = torch.as_tensor(np.asarray([rotations[i] for i in ids]), dtype=torch.float32).to(self.dev)     # [n,3,3]
        d = self.centre - self.centre @ R.transpose(-1, -2)                                                     # c - R c
        view, proj, campos = self.stack.moved(R, d)
        acc = torch.zeros(len(ids), dtype=torch.float32, device=self.dev)
        prev, prev_reserve = PairPolicy.mode, PairPolicy.reserve
        PairPolicy.mode, PairPolicy.reserve = "reserve", max(prev_reserve, self.batch * self.reserve)
        try:
            self._score_all(view, proj, campos, acc)
            try:
                PairPolicy.drain()       # the sweep's one synchronisation: every launch set's overflow word
            except RuntimeError:
                # the reservation was too small for some hypothesis (drain has grown it): score again
                log.warning("StackedSweep: pair reservation grown during the sweep; scoring the hypotheses again")
                acc.zero_()
                self._score_all(view, proj, campos, acc)
                self.reserve = max(self.reserve, PairPolicy.drain() * 2 // self.batch)
        finally:
            PairPolicy.mode, PairPolicy.reserve = prev, prev_reserve
        return [(-acc[k]).reshape(1) for k in range(len(ids))]


class SweepPlan:
    """Everything of a rotation sweep that does not depend on the hypothesis, built once: on a GPU one hypothesis (rotate
    + len(cameras) renders + comparison) is captured as a HIP graph and `score()` replays it per rotation - the sweep's
    kernels are tens of microseconds each, so eager launches are CPU-bound (3.1k renders/s eager, 5.1k replayed at 100k
    Gaussians / 800x800).  The pair reservation is learned from an eager pass; every replay's overflow flags are folded
    into a device-side maximum and checked once per `score()`.  With `use_graph=False` (CPU stand-in renderers, capture
    failures) hypotheses are scored with eager launches, each verified by PairPolicy.drain()."""

    def __init__(self, model, cameras, targets, bg, use_graph=None, render_fn=render, stacked=None):
        self.model, self.cameras, self.targets, self.bg, self.render_fn = model, cameras, targets, bg, render_fn
        self.dev = model._xyz.device
        self.graph = None
        # SH-0 object on the GPU: cameras are moved, not the object, and the views render as one stacked image
        self.stacked = None
        if stacked is None:
            stacked = use_graph is None and _stacked_ok(model, cameras, render_fn)
        if stacked:
            if not _stacked_ok(model, cameras, render_fn):
                raise ValueError("the stacked sweep needs an SH-0 GPU model and cameras of one resolution (height % 16 == 0)")
            self.stacked = StackedSweep(model, cameras, targets, bg)
            self.reserve, self.fallback_reason = 0, None
            return
        self.reserve = 0        # this plan's pair-reservation floor: applied only inside its own calls (_reserved)
        self.fallback_reason = None
        if use_graph is None:
            use_graph = self.dev.type == "cuda" and render_fn is render
        if use_graph:
            try:
                self._capture()
            except Exception as e:   # capture unsupported: the eager path is always correct - but SAY so (1.5x slower)
                torch.cuda.synchronize()
                self.graph = None
                self.fallback_reason = f"capture failed: {type(e).__name__}: {e}"
                log.warning("SweepPlan: HIP-graph capture failed (%s); hypotheses will be scored with eager launches", self.fallback_reason)

    @contextlib.contextmanager
    def _reserved(self):
        """PairPolicy.reserve raised to this plan's floor for the duration of one of its calls, restored afterwards (the
        plan used to leave its small floor behind as the process-wide default of every later context)."""
        prev = PairPolicy.reserve
        PairPolicy.reserve = max(prev, self.reserve)
        try:
            yield
        finally:
            PairPolicy.reserve = prev

    def _capture(self):
        prev, pend_before, reserve_before = PairPolicy.mode, PairPolicy._pending, PairPolicy.reserve
        PairPolicy.mode = "reserve"
        try:
            PairPolicy._pending = []
            self.Rbuf = torch.eye(3, dtype=torch.float32, device=self.dev)
            hypothesis_fitness(self.model, self.Rbuf, self.cameras, self.targets, self.bg)
            worst = PairPolicy.drain()                       # sizes the reservation (raises if the default was too small)
            self.reserve = int(2.0 * worst) + 4096           # other rotations see other pair counts
            PairPolicy.reserve = max(reserve_before, self.reserve)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                hypothesis_fitness(self.model, self.Rbuf, self.cameras, self.targets, self.bg)
                PairPolicy.drain()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.fit = hypothesis_fitness(self.model, self.Rbuf, self.cameras, self.targets, self.bg)
                states, PairPolicy._pending = PairPolicy._pending, []
                # StateHeader: num_pairs u32 @0, overflow u32 @4
                self.flags = torch.stack([p.header[:8].view(torch.int32) for p in states]).amax(0)
            self.graph = graph
        finally:
            PairPolicy.mode = prev
            PairPolicy._pending = pend_before
            PairPolicy.reserve = reserve_before

    def score(self, rotations, ids):
        """Fitness (1-element float tensors) of the hypotheses `ids` of `rotations`."""
        if self.stacked is not None:
            return self.stacked.score(rotations, ids)
        if self.graph is not None:
            try:
                worst_flags = torch.zeros(2, dtype=torch.int32, device=self.dev)
                out = []
                for i in ids:
                    self.Rbuf.copy_(torch.as_tensor(rotations[i], dtype=torch.float32), non_blocking=True)
                    self.graph.replay()
                    out.append(self.fit.reshape(1).float().clone())
                    worst_flags = torch.maximum(worst_flags, self.flags)
                n_pairs, overflow = (int(v) for v in worst_flags.tolist())   # the sweep's one synchronisation
                if not overflow:
                    return out
                self.reserve = max(self.reserve, int(1.25 * n_pairs) + 4096)
                self.fallback_reason = f"a replay needed {n_pairs} pairs, more than the captured reservation"
            except Exception as e:
                torch.cuda.synchronize()
                self.fallback_reason = f"replay failed: {type(e).__name__}: {e}"
            self.graph = None   # reservation overflowed during replay / replay failed: eager from here on
            log.warning("SweepPlan: leaving the captured plan (%s); eager launches from here on", self.fallback_reason)
        with self._reserved():
            return self._score_eager(rotations, ids)

    def _score_eager(self, rotations, ids):
        """Renders issued without host synchronisation (`PairPolicy` "reserve"), verified once per hypothesis; a hypothesis
        whose reservation overflowed is re-scored with exact sizing."""
        out = []
        on_gpu = self.dev.type == "cuda"
        for i in ids:
            R = torch.as_tensor(rotations[i], dtype=torch.float32, device=self.dev)
            prev = PairPolicy.mode
            PairPolicy.mode = "reserve" if on_gpu else prev
            try:
                f = hypothesis_fitness(self.model, R, self.cameras, self.targets, self.bg, self.render_fn)
                if on_gpu:
                    PairPolicy.drain()
            except RuntimeError:
                PairPolicy.mode = "exact"
                f = hypothesis_fitness(self.model, R, self.cameras, self.targets, self.bg, self.render_fn)
            finally:
                PairPolicy.mode = prev
            out.append(f.reshape(1).float())
        return out


def _raise_together(any_failed, err, what):
    """A rank whose local part failed still enters the exchange it would otherwise have skipped (nobody waits for a rank that
    never arrives) and says so in the status row every rank sends through the same all-gather (parallel.gather_results,
    `failed=`): no extra collective and no extra host synchronisation on the success path; the flag - read where the result
    is read anyway - raises on EVERY rank.  (Round 5 marked a failure by NaN rows: a legitimately NaN score raised "another
    rank failed", and a failing rank with an empty shard had no row to put the NaN in.)"""
    if err is not None or bool(any_failed):
        raise RuntimeError(f"{what}: the local part failed on " + ("this rank" if err is not None else "another rank")) from err


def _to_device_or_reraise(rows, dev, err):
    """The (zero) rows a failed rank sends, built on the host: if even the copy to the device fails - the local error was a device
    fault - the original error is raised instead of a second one (the peers then see the collective's timeout)."""
    try:
        return rows.to(dev)
    except Exception:   # noqa: BLE001
        raise err


def align_objects(models, rotations, cameras, targets_per_object, bg, render_fn=render):
    """The rotation sweep with the OBJECT as the unit of sharding (align_3dgs_clpe_9dof.py:489-499 loops over objects
    serially; they share no mutable state): object j -> rank j mod G, every rank sweeps all hypotheses of its objects
    locally, and ONE fixed-size all-gather of (object, best id, best fitness) tells every rank every result.
    Returns [(best id, best fitness)] per object, identical on every rank."""
    from .parallel import gather_results, shard_indices
    dev = models[0]._xyz.device
    mine = shard_indices(len(models))
    rows, err = [], None
    try:
        for j in mine:
            _, fit, best = rotation_sweep(models[j], rotations, cameras, targets_per_object[j], bg, render_fn=render_fn, shard=False)
            rows.append(torch.stack([torch.tensor(float(best), device=fit.device), fit[best, 0].float()]))
    except Exception as e:   # noqa: BLE001   (raised below, on every rank, behind the exchange)
        err, rows = e, None
    if err is not None:
        v = _to_device_or_reraise(torch.zeros((len(mine), 2), dtype=torch.float32), dev, err)
    else:
        v = torch.stack(rows).to(dev) if rows else torch.zeros((0, 2), dtype=torch.float32, device=dev)
    _, vals, bad = gather_results(mine, v, n_total=len(models), failed=err is not None)
    vals = torch.cat([vals.reshape(-1), bad.reshape(1).to(vals.dtype)]).cpu()     # the one synchronisation of the call
    _raise_together(float(vals[-1]) != 0.0, err, "align_objects")
    vals = vals[:-1].view(len(models), 2)
    return [(int(vals[j, 0]), float(vals[j, 1])) for j in range(len(models))]


def rotation_sweep(model, rotations, cameras, targets, bg, use_graph=None, plan=None, render_fn=render, shard=True):
    """Score every rotation hypothesis (hypothesis j -> rank j mod world, one all-gather of (id, fitness) at the end);
    returns (ids, fitness[n,1], best id) on every rank.  `plan`: a SweepPlan built earlier (its graph capture is then
    outside the caller's timed region).  `shard=False`: this rank scores every hypothesis itself (align_objects)."""
    from .parallel import gather_results, shard_indices
    dev = model._xyz.device
    if not shard:
        if plan is None:
            plan = SweepPlan(model, cameras, targets, bg, use_graph=use_graph, render_fn=render_fn)
        ids = list(range(len(rotations)))
        scores = torch.stack(plan.score(rotations, ids)) if ids else torch.zeros((0, 1), dtype=torch.float32, device=dev)
        return torch.arange(len(ids), device=scores.device), scores, (int(torch.argmax(scores[:, 0])) if ids else -1)
    mine = shard_indices(len(rotations))
    err = None
    try:
        if plan is None:
            plan = SweepPlan(model, cameras, targets, bg, use_graph=use_graph, render_fn=render_fn)
        vals = plan.score(rotations, mine) if mine else []
    except Exception as e:   # noqa: BLE001   (raised below, on every rank, behind the exchange)
        err, vals = e, None
    if err is not None:
        v = _to_device_or_reraise(torch.zeros((len(mine), 1), dtype=torch.float32), dev, err)
    else:
        v = torch.stack(vals).to(dev) if vals else torch.zeros((0, 1), dtype=torch.float32, device=dev)
    # ONE fixed-size all-gather, no host sync; every rank's status row rides in it
    ids, scores, any_failed = gather_results(mine, v, n_total=len(rotations), failed=err is not None)
    if not ids.numel():
        _raise_together(float(any_failed) != 0.0, err, "rotation_sweep")
        return ids, scores, -1
    # (the flag comes back in the same device-to-host read that fetches the arg-max; a NaN score is a score, not a failure)
    best, bad = (int(x) for x in torch.stack([ids[torch.argmax(torch.nan_to_num(scores[:, 0], nan=-float("inf")))].long(),
                                              (any_failed != 0).long().to(ids.device)]).tolist())
    _raise_together(bad, err, "rotation_sweep")
    return ids, scores, best
