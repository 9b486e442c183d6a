"""Opt-in acceleration of the REFERENCE's own Python around the rasterizer - for a SCORP checkout that is run with no edit
at all (only the three extension modules are replaced, so `render()`, `GaussianModel` and
`loss_utils.ssim` are the reference's).  In that setting 9.4 of a view's 11 ms at 1 M Gaussians, 1600x1200 are
`gs3dgs/utils/loss_utils.py:37-73`'s five depthwise convolutions.  With `SCORP_AMD_ACCELERATE=1` in the environment the
drop-in packages call `accelerate_reference()` when they are imported: the function object `loss_utils.ssim` of
`gs3dgs` / `gs2dgs` keeps its identity (names bound earlier by `from ... import ssim` stay valid) and gets a new body that
answers from the HIP loss kernels for the arguments `train_3dgs.py:107` passes and from the original code otherwise
(kept as `loss_utils.ssim_torch`).  Off by default: it edits another package's function at run time.
"""
import importlib
import os
import sys
import types

_MODULES = ("gs3dgs.utils.loss_utils", "gs2dgs.utils.loss_utils")


def patch_ssim(module):
    """Gives `module.ssim(img1, img2, window_size=11, size_average=True)` the dispatching body.  True if it was patched now."""
    orig = getattr(module, "ssim", None)
    if not isinstance(orig, types.FunctionType) or orig.__closure__ is not None or getattr(orig, "_scorp_patched", False):
        return False
    keep = types.FunctionType(orig.__code__, orig.__globals__, "ssim_torch", orig.__defaults__, None)
    keep.__kwdefaults__ = orig.__kwdefaults__

    def dispatch(img1, img2, window_size=11, size_average=True):
        from .loss import _hip_ssim_applies
        if _hip_ssim_applies(img1, img2, window_size, size_average):
            from .fused_loss import fused_l1_ssim_loss
            return 1.0 - fused_l1_ssim_loss(img1, img2, 1.0)
        return keep(img1, img2, window_size, size_average)

    def ssim(img1, img2, window_size=11, size_average=True):       # (closure-free: resolved in the patched module's globals)
        return _scorp_ssim_dispatch(img1, img2, window_size, size_average)   # noqa: F821

    orig.__globals__["_scorp_ssim_dispatch"] = dispatch
    orig.__globals__["ssim_torch"] = keep
    orig.__code__ = ssim.__code__
    orig.__defaults__ = (11, True)
    orig.__kwdefaults__ = None
    orig._scorp_patched = True
    return True


def accelerate_reference(force=False):
    """Patches the reference's `loss_utils.ssim` (both packages) if SCORP_AMD_ACCELERATE=1 or `force`; returns the names of
    the modules it patched.  A module that cannot be imported is skipped silently (the scripts of the other rasterizer)."""
    if not force and os.environ.get("SCORP_AMD_ACCELERATE", "0") in ("", "0"):
        return []
    done = []
    for name in _MODULES:
        mod = sys.modules.get(name)
        if mod is None:
            try:
                mod = importlib.import_module(name)
            except Exception:      # noqa: BLE001   (not this checkout's package, or its own imports are missing)
                continue
        if patch_ssim(mod):
            done.append(name)
    return done
