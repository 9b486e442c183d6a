"""`diff_gaussian_rasterization`-compatible front-end over libscorp_gs.so.

Same names, argument meaning and error behaviour as the module the reference imports at
gs3dgs/gaussian_renderer/__init__.py:15 and calls at :51-66,101-111:
`GaussianRasterizationSettings` (12 fields) and `GaussianRasterizer(raster_settings)(means3D, means2D, shs,
colors_precomp, opacities, scales, rotations, cov3D_precomp) -> (color[3,H,W], radii[N] int32, depth[1,H,W],
alpha[1,H,W])`, differentiable w.r.t. all eight arguments.  PyTorch supplies device memory, the current
stream and autograd bookkeeping; every kernel is in the HIP library, reached through ctypes.
"""
import contextlib
import ctypes
import threading
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class _Pending:
    """One forward issued without a host synchronisation: a copy of its 64-byte StateHeader (pair count, overflow flag),
    the event recorded behind that copy on the stream that produced it, the reservation context it belongs to and the
    number of Gaussians it rendered."""
    __slots__ = ("header", "event", "key", "n")

    def __init__(self, header, event, key, n=0):
        self.header, self.event, self.key, self.n = header, event, key, int(n)

    def numel(self):
        return self.header.numel()


class PairOverflow(RuntimeError):
    """Raised by PairPolicy.drain(): `count` pending views had overflowed their reserved pair buffer (each was discarded on
    the device: no optimizer step, no statistics)."""
    def __init__(self, msg, count):
        super().__init__(msg)
        self.count = int(count)


class PairPolicy:
    """How the (tile,splat) pair buffer is sized.

    "exact"   (default) one 8-byte D2H read per forward tells the exact pair count — what the CUDA original
              does too; always correct.
    "reserve" no host sync: the buffer is sized from the largest count seen so far IN THE SAME CONTEXT times `slack`;
              each forward is queued for `drain()`, which waits for the forward's own stream, raises if any view
              overflowed and grows that context's reservation.  For throughput loops that check once per N views.

    A context is (image height, image width, stream): the align loop changes resolution per render, `bench.py
    --streams` runs views on several streams — each gets its own reservation instead of one process-wide number.  What a
    context remembers is pairs PER GAUSSIAN at the model size it was learned on: when the model grows or shrinks
    (densify_and_prune, every 100 iterations) the reservation scales with it instead of starting again from a default
    (a scene that needs more than 4 pairs per Gaussian used to overflow after every densification).  `reserve` is only
    a caller-set FLOOR for every context (0 = none); `mode` / `slack` are configuration.  A context nobody has sized yet
    starts at max(4 N, 2^20) pairs.
    """
    mode = "exact"
    slack = 1.25
    reserve = 0          # pairs: floor applied to every context (callers that know their workload set it)
    _ctx = {}            # context key -> (reserved pairs, number of Gaussians they were learned on)
    _pending = []        # _Pending entries whose overflow flag has not been read yet
    _MAX_CTX = 64        # contexts kept (least recently learned dropped first)

    @classmethod
    def key(cls, N, H, W):
        return (int(H), int(W), _C.current_stream_ptr())

    @classmethod
    def capacity(cls, N, H, W):
        """Pairs to reserve for a forward of this context."""
        got, n0 = cls._ctx.get(cls.key(N, H, W), (0, 0))
        if got > 0 and n0 > 0 and int(N) != n0:
            got = cls._rescaled(got, n0, N, H, W)
        if got <= 0 and cls.reserve <= 0:
            got = max(4 * int(N), 1 << 20)
        return max(got, int(cls.reserve))

    @classmethod
    def _rescaled(cls, got, n0, N, H, W):
        """What a context learned on n0 Gaussians, for a model of N: the same pairs per Gaussian while the model is the SAME
        model resized (densify_and_prune: within 0.5x ... 2x); anything else is another scene that happens to share the
        resolution and the stream - a 10 k-Gaussian object that fills the screen (200 pairs per Gaussian, align's eager
        sweeps) says nothing about a multi-million-Gaussian scene - and starts from the default, as a context nobody has
        sized.  Never more than every Gaussian in every tile."""
        ratio = int(N) / n0
        if not 0.5 <= ratio <= 2.0:
            return 0
        tiles = ((int(W) + 15) // 16) * ((int(H) + 15) // 16)
        return min(max(int(got * ratio) + 1024, 1 << 16), max(tiles * int(N), 1 << 16))

    @classmethod
    def set_context(cls, N, H, W, pairs):
        """Seed / override the reservation of the current stream's (H, W) context (tests, callers that know their scene)."""
        cls._ctx[cls.key(N, H, W)] = (int(pairs), int(N))

    @classmethod
    def pend(cls, state, N, H, W, header=None):
        """Queue a forward for drain(): copies the header on the current stream and records an event behind the copy.
        `header`: a 64-byte tensor the library already filled with the first four header words (scorp_gs3d_train_view's
        out_header): then nothing is copied."""
        ev = None
        if header is not None:
            hdr = header
            if not torch.cuda.is_current_stream_capturing():
                ev = torch.cuda.Event()
                ev.record()
        elif torch.cuda.is_current_stream_capturing():
            # a captured replay is checked by its capturer (align.SweepPlan), which reads the live headers at the end of
            # the graph: a view of the state (kept alive by the graph's pool anyway) instead of a copy node per view
            hdr = state[:64]
        else:
            hdr = state[:64].clone()     # not the state itself, or every pending view would pin ~100 MB until the drain
            ev = torch.cuda.Event()
            ev.record()
        cls._pending.append(_Pending(hdr, ev, cls.key(N, H, W), N))
        return hdr

    @classmethod
    def drain(cls):
        """Verify every forward issued in "reserve" mode since the last drain (each on its own stream's event)."""
        pend, cls._pending = cls._pending, []
        worst, n_over = 0, 0
        err = None
        if not pend:
            return worst
        for p in pend:
            if p.event is not None:
                p.event.synchronize()     # the header was written on the stream that rendered the view
        # ONE device-to-host copy for all pending headers (first four words: pairs needed, overflow, capacity, -), per
        # device.  One scorp_gs3d_check_overflow per view was a copy and a stream synchronisation each - 32 of them per
        # drain, ~0.1 ms per training iteration of a 100k-Gaussian object.
        words = [None] * len(pend)
        by_dev = {}
        for i, p in enumerate(pend):
            by_dev.setdefault(p.header.device, []).append(i)
        for dev, idx in by_dev.items():
            rows = torch.stack([pend[i].header.reshape(-1).view(torch.uint8)[:16] for i in idx]).cpu().numpy().view("<u4")
            for i, r in zip(idx, rows):
                words[i] = r
        for p, w in zip(pend, words):
            n, overflow, capacity = int(w[0]), int(w[1]), int(w[2])
            worst = max(worst, n)
            if p.key is not None:
                need = int(n * cls.slack) + 1024
                got, n0 = cls._ctx.pop(p.key, (0, 0))
                if n0 > 0 and p.n > 0 and n0 != p.n:
                    got = int(got * (p.n / n0)) if 0.5 <= p.n / n0 <= 2.0 else 0   # what was learned, at this view's model size (_rescaled)
                cls._ctx[p.key] = (max(got, need), p.n if p.n > 0 else n0)   # (re-inserted last: most recently learned)
                while len(cls._ctx) > cls._MAX_CTX:
                    cls._ctx.pop(next(iter(cls._ctx)))
            if overflow:
                n_over += 1
                if err is None:
                    err = f"pair buffer overflow: {n} pairs needed, capacity {capacity}"
        if err:
            raise PairOverflow(f"pair reservation too small ({err}; {n_over} of {len(pend)} pending views); the context's "
                               "reservation has been grown, re-run the view(s)", n_over)
        return worst

    @classmethod
    def reset(cls):
        cls.mode, cls.reserve, cls._ctx, cls._pending = "exact", 0, {}, []


_tls = threading.local()
# SCORP_BACKWARD_DETERMINISTIC=1: every backward of this process runs the atomic-free form (include/scorp_gs.h)
_ENV_FLAGS = _C.BACKWARD_DETERMINISTIC if __import__("os").environ.get("SCORP_BACKWARD_DETERMINISTIC", "0") not in ("", "0") else 0


def _backward_flags():
    return int(getattr(_tls, "backward_flags", 0)) | _ENV_FLAGS


@contextlib.contextmanager
def backward_precision(mode):
    """Forwards issued inside `with backward_precision("exact_fp32"):` run their backward with fp32 MFMAs throughout
    (scorp_gs3d_backward_ex, SCORP_BACKWARD_EXACT_FP32) instead of the default two-term fp16 split of the pixel->splat
    reduction.  The choice is recorded per forward (thread-local while the block is active), so a backward that runs
    after the block still honours it.  Used by the parity tests to compare the two forms.
    "deterministic" / "exact_fp32_deterministic": the same reductions with the sums leaving as plain per-(block, hit) rows
    that a second kernel adds per Gaussian in a fixed order (SCORP_BACKWARD_DETERMINISTIC): no float atomics, two runs
    give the same bits - what a caller that votes on gradient SIGNS wants (utils/mask.py:52,65,89,124)."""
    assert mode in ("split", "exact_fp32", "deterministic", "exact_fp32_deterministic")
    prev = getattr(_tls, "backward_flags", 0)
    _tls.backward_flags = (_C.BACKWARD_EXACT_FP32 if mode.startswith("exact_fp32") else 0) | \
                          (_C.BACKWARD_DETERMINISTIC if mode.endswith("deterministic") else 0)
    try:
        yield
    finally:
        _tls.backward_flags = prev


KEEP_LAST_FORWARD = False   # diagnostics (bench.py): keep (state, N, W, H) of the most recent forward in LAST_FORWARD
LAST_FORWARD = None
LAST_NUM_PAIRS_LOG = []   # pair counts of the most recent "exact"-mode forwards (diagnostics / bench bookkeeping)


def _stream():
    return ctypes.c_void_p(_C.current_stream_ptr())


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _prep(t, name, shape_tail=None):
    """float32, contiguous, on the GPU; None stays None (maps to NULL in the C ABI)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (scorp_amd has no CPU path)")
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if t.data_ptr() % 16:      # an offset view: the kernels fetch 16-byte words
        t = t.clone()
    return t


def _inputs_struct(s, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D, keep, sh_rest=None, raw=0):
    """Fill the C struct; `keep` collects tensors that must outlive the call."""
    bg, vm, pm, cp = (_prep(s.bg, "bg"), _prep(s.viewmatrix, "viewmatrix"), _prep(s.projmatrix, "projmatrix"),
                      _prep(s.campos, "campos"))
    keep.extend([bg, vm, pm, cp])
    a = _C.ScorpGs3dInputs()
    a.num_gaussians = means3D.shape[0]
    a.sh_degree = int(s.sh_degree)
    a.sh_coeffs = 0 if sh is None else (sh.shape[1] if sh_rest is None else 1 + sh_rest.shape[1])
    a.shs_rest = None if sh_rest is None else sh_rest.data_ptr()
    a.raw_params = int(raw)
    a.image_width, a.image_height = int(s.image_width), int(s.image_height)
    a.tanfovx, a.tanfovy, a.scale_modifier = float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier)
    a.prefiltered, a.debug = int(bool(s.prefiltered)), int(bool(s.debug))
    a.bg, a.viewmatrix, a.projmatrix, a.campos = bg.data_ptr(), vm.data_ptr(), pm.data_ptr(), cp.data_ptr()
    a.means3D = means3D.data_ptr()
    a.shs = None if sh is None else sh.data_ptr()
    a.colors_precomp = None if colors_precomp is None else colors_precomp.data_ptr()
    a.opacities = opacities.data_ptr()
    a.scales = None if scales is None else scales.data_ptr()
    a.rotations = None if rotations is None else rotations.data_ptr()
    a.cov3D_precomp = None if cov3D is None else cov3D.data_ptr()
    return a


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings):
        dev = means3D.device
        means3D, sh, colors_precomp = _prep(means3D, "means3D"), _prep(sh, "shs"), _prep(colors_precomp, "colors_precomp")
        opacities, scales, rotations = _prep(opacities, "opacities"), _prep(scales, "scales"), _prep(rotations, "rotations")
        cov3Ds_precomp = _prep(cov3Ds_precomp, "cov3D_precomp")
        color, radii, depth, alpha, state, pairs, keep = _forward_common(
            ctx, settings, means3D, sh, None, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raw=0)
        ctx.has = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None)
        none = torch.empty(0, device=dev)
        ctx.save_for_backward(means3D, none if sh is None else sh, none if colors_precomp is None else colors_precomp,
                              opacities, none if scales is None else scales, none if rotations is None else rotations,
                              none if cov3Ds_precomp is None else cov3Ds_precomp, state, pairs, *keep)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        L = _C.lib()
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3D, state, pairs, bg, vm, pm, cp = ctx.saved_tensors
        has_sh, has_col, has_sr, has_cov = ctx.has
        sh = sh if has_sh else None
        colors_precomp = colors_precomp if has_col else None
        scales, rotations = (scales, rotations) if has_sr else (None, None)
        cov3D = cov3D if has_cov else None
        s = ctx.settings._replace(bg=bg, viewmatrix=vm, projmatrix=pm, campos=cp)
        keep = []
        args = _inputs_struct(s, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D, keep)
        N, dev = means3D.shape[0], means3D.device
        need = ctx.needs_input_grad  # means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, settings
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        g_means3D = new(N, 3) if need[0] else None
        g_means2D = new(N, 3) if need[1] else None
        g_sh = torch.empty_like(sh) if (need[2] and has_sh) else None
        g_col = new(N, 3) if (need[3] and has_col) else None
        g_op = torch.empty_like(opacities) if need[4] else None
        g_sc = new(N, 3) if (need[5] and has_sr) else None
        g_rot = new(N, 4) if (need[6] and has_sr) else None
        g_cov = new(N, 6) if (need[7] and has_cov) else None
        grads = _C.ScorpGs3dGrads()
        grads.means3D, grads.means2D, grads.shs, grads.colors_precomp = _ptr(g_means3D), _ptr(g_means2D), _ptr(g_sh), _ptr(g_col)
        grads.opacities, grads.scales, grads.rotations, grads.cov3D_precomp = _ptr(g_op), _ptr(g_sc), _ptr(g_rot), _ptr(g_cov)
        if grad_color is None:
            grad_color = torch.zeros((3, int(s.image_height), int(s.image_width)), dtype=torch.float32, device=dev)
        gc = _prep(grad_color, "grad_color")
        gd = _prep(grad_depth, "grad_depth") if grad_depth is not None else None
        ga = _prep(grad_alpha, "grad_alpha") if grad_alpha is not None else None
        scratch_bytes = L.scorp_gs3d_backward_scratch_bytes_ex(N, int(s.image_width), int(s.image_height), ctx.capacity, ctx.backward_flags)
        scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
        _C.check(L.scorp_gs3d_backward_ex(ctypes.byref(args), _ptr(state), _ptr(pairs), ctx.capacity, _ptr(gc), _ptr(gd),
                                          _ptr(ga), ctypes.byref(grads), _ptr(scratch), scratch_bytes, ctx.backward_flags,
                                          _stream()),
                 "scorp_gs3d_backward_ex")
        return g_means3D, g_means2D, g_sh, g_col, g_op, g_sc, g_rot, g_cov, None


def _forward_common(ctx, settings, means3D, sh, sh_rest, colors_precomp, opacities, scales, rotations, cov3D, raw):
    """Shared by the two autograd Functions: allocate outputs + workspaces, run preprocess / render."""
    L = _C.lib()
    dev = means3D.device
    N, H, W = means3D.shape[0], int(settings.image_height), int(settings.image_width)
    keep = []
    args = _inputs_struct(settings, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D, keep, sh_rest, raw)
    color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    alpha = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((N,), dtype=torch.int32, device=dev)
    state_bytes = L.scorp_gs3d_state_bytes(N, W, H)
    state = torch.empty(state_bytes, dtype=torch.uint8, device=dev)
    stream = _stream()
    _C.check(L.scorp_gs3d_preprocess(ctypes.byref(args), _ptr(radii), _ptr(state), state_bytes, stream), "scorp_gs3d_preprocess")
    if PairPolicy.mode == "exact":
        n = ctypes.c_uint64(0)
        _C.check(L.scorp_gs3d_num_pairs(_ptr(state), stream, ctypes.byref(n)), "scorp_gs3d_num_pairs")
        capacity = max(int(n.value), 1)
        LAST_NUM_PAIRS_LOG.append(int(n.value))
        del LAST_NUM_PAIRS_LOG[:-64]
    else:
        capacity = PairPolicy.capacity(N, H, W)
    pairs = torch.empty(L.scorp_gs3d_pairs_bytes(capacity), dtype=torch.uint8, device=dev)
    # nothing to differentiate (the calls the reference makes under torch.no_grad()): the image-only render, which
    # leaves no state for a backward pass
    fn = L.scorp_gs3d_render if want_backward(ctx) else L.scorp_gs3d_render_image
    _C.check(fn(ctypes.byref(args), _ptr(state), _ptr(pairs), capacity, _ptr(color), _ptr(depth), _ptr(alpha), stream),
             "scorp_gs3d_render")
    if PairPolicy.mode != "exact":
        PairPolicy.pend(state, N, H, W)   # what drain() will look at: a copy of the StateHeader the render just filled in
    if KEEP_LAST_FORWARD:
        global LAST_FORWARD
        LAST_FORWARD = (state, N, W, H)
    ctx.settings, ctx.capacity = settings, capacity
    ctx.backward_flags = _backward_flags()
    ctx.set_materialize_grads(False)   # unused outputs arrive as None in backward: the kernels skip those terms
    return color, radii, depth, alpha, state, pairs, keep


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Same rasterizer on the GaussianModel's RAW storage (logit opacity, log scale, un-normalised quaternion,
    _features_dc / _features_rest kept apart): the activations of gaussian_model.py:126-146 and the SH concat run
    inside the per-Gaussian kernels, and the gradients come back w.r.t. the raw leaves."""

    @staticmethod
    def forward(ctx, means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, settings):
        means3D, f_dc, f_rest = _prep(means3D, "means3D"), _prep(f_dc, "features_dc"), _prep(f_rest, "features_rest")
        opacity_raw, scaling_raw, rotation_raw = _prep(opacity_raw, "opacity"), _prep(scaling_raw, "scaling"), _prep(rotation_raw, "rotation")
        color, radii, depth, alpha, state, pairs, keep = _forward_common(
            ctx, settings, means3D, f_dc, f_rest, None, opacity_raw, scaling_raw, rotation_raw, None, raw=7)
        ctx.save_for_backward(means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, state, pairs, *keep)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        L = _C.lib()
        means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, state, pairs, bg, vm, pm, cp = ctx.saved_tensors
        s = ctx.settings._replace(bg=bg, viewmatrix=vm, projmatrix=pm, campos=cp)
        keep = []
        args = _inputs_struct(s, means3D, f_dc, None, opacity_raw, scaling_raw, rotation_raw, None, keep, f_rest, 7)
        N, dev = means3D.shape[0], means3D.device
        need = ctx.needs_input_grad
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        g_means3D = new(N, 3) if need[0] else None
        g_means2D = new(N, 3) if need[1] else None
        want_sh = need[2] or need[3]
        g_dc = torch.empty_like(f_dc) if want_sh else None
        g_rest = torch.empty_like(f_rest) if want_sh else None
        g_op = torch.empty_like(opacity_raw) if need[4] else None
        g_sc = new(N, 3) if need[5] else None
        g_rot = new(N, 4) if need[6] else None
        grads = _C.ScorpGs3dGrads()
        grads.means3D, grads.means2D, grads.shs, grads.shs_rest = _ptr(g_means3D), _ptr(g_means2D), _ptr(g_dc), _ptr(g_rest)
        grads.opacities, grads.scales, grads.rotations = _ptr(g_op), _ptr(g_sc), _ptr(g_rot)
        if grad_color is None:
            grad_color = torch.zeros((3, int(s.image_height), int(s.image_width)), dtype=torch.float32, device=dev)
        gc = _prep(grad_color, "grad_color")
        gd = _prep(grad_depth, "grad_depth") if grad_depth is not None else None
        ga = _prep(grad_alpha, "grad_alpha") if grad_alpha is not None else None
        scratch_bytes = L.scorp_gs3d_backward_scratch_bytes_ex(N, int(s.image_width), int(s.image_height), ctx.capacity, ctx.backward_flags)
        scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
        _C.check(L.scorp_gs3d_backward_ex(ctypes.byref(args), _ptr(state), _ptr(pairs), ctx.capacity, _ptr(gc), _ptr(gd),
                                          _ptr(ga), ctypes.byref(grads), _ptr(scratch), scratch_bytes, ctx.backward_flags,
                                          _stream()),
                 "scorp_gs3d_backward_ex")
        return (g_means3D, g_means2D, g_dc if need[2] else None, g_rest if need[3] else None, g_op, g_sc, g_rot, None)


def want_backward(ctx):
    """True if a backward pass can follow this forward.  `ctx.needs_input_grad` alone says yes for every leaf parameter
    even under torch.no_grad() (it ignores the grad mode, and inside Function.forward the mode is always off), so the
    wrappers below note the caller's grad mode before `apply`."""
    return bool(getattr(_tls, "grad_mode", True)) and any(ctx.needs_input_grad)


def rasterize_gaussians_raw(means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings):
    _tls.grad_mode = torch.is_grad_enabled()
    return _RasterizeGaussiansRaw.apply(means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    _tls.grad_mode = torch.is_grad_enabled()
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Frustum test of the upstream API (not called by the reference): view-space z > 0.2."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix
            z = positions @ vm[:3, 2] + vm[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)


class _RenderTail(torch.autograd.Function):
    """(render_depth, visibility_filter) from (depth, alpha, radii) in one launch: gaussian_renderer/__init__.py:113-120."""

    @staticmethod
    def forward(ctx, depth, alpha, radii):
        L = _C.lib()
        depth, alpha = _prep(depth, "depth"), _prep(alpha, "alpha")
        radii = radii.contiguous()
        out = torch.empty_like(depth)
        vis = torch.empty(radii.shape, dtype=torch.bool, device=radii.device)
        _C.check(L.scorp_gs3d_render_tail(_ptr(depth), _ptr(alpha), depth.numel(), _ptr(radii), radii.numel(), _ptr(out),
                                          _ptr(vis), _stream()), "scorp_gs3d_render_tail")
        ctx.save_for_backward(depth, alpha)
        ctx.mark_non_differentiable(vis)
        return out, vis

    @staticmethod
    def backward(ctx, g_out, _g_vis):
        L = _C.lib()
        depth, alpha = ctx.saved_tensors
        g_out = _prep(g_out, "grad")
        g_depth, g_alpha = torch.empty_like(depth), torch.empty_like(alpha)
        _C.check(L.scorp_gs3d_render_tail_backward(_ptr(g_out), _ptr(depth), _ptr(alpha), depth.numel(), _ptr(g_depth),
                                                   _ptr(g_alpha), _stream()), "scorp_gs3d_render_tail_backward")
        return g_depth, g_alpha, None


def render_tail(depth, alpha, radii):
    """render_depth = nan_to_num(depth / alpha, 0, 0) and visibility_filter = radii > 0, fused."""
    if not depth.is_cuda:
        raise RuntimeError("render_tail needs GPU tensors (scorp_amd has no CPU path)")
    return _RenderTail.apply(depth, alpha, radii.int())
