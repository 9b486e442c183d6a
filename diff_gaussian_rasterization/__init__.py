"""Drop-in for the `diff_gaussian_rasterization` module the reference imports
(gs3dgs/gaussian_renderer/__init__.py:15): same two names, backed by the gfx950 HIP library."""
from scorp_amd.rasterizer3d import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians  # noqa: F401

# opt-in (SCORP_AMD_ACCELERATE=1): the reference's own loss_utils.ssim answers from the HIP loss kernels (scorp_amd/hostpatch.py)
from scorp_amd.hostpatch import accelerate_reference as _accelerate_reference  # noqa: E402
_accelerate_reference()
