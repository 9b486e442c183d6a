"""Drop-in for the `diff_surfel_rasterization` module the reference imports
(gs2dgs/gaussian_renderer/__init__.py:14): same two names, backed by the gfx950 HIP library."""
from scorp_amd.rasterizer2d import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

# opt-in (SCORP_AMD_ACCELERATE=1): the reference's own loss_utils.ssim answers from the HIP loss kernels (scorp_amd/hostpatch.py)
from scorp_amd.hostpatch import accelerate_reference as _accelerate_reference  # noqa: E402
_accelerate_reference()
