"""Drop-in for the `diff_gaussian_rasterization` module the reference imports
(gs3dgs/gaussian_renderer/__init__.py:15): same two names, backed by the gfx950 HIP library."""
from scorp_amd.rasterizer3d import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians  # noqa: F401
