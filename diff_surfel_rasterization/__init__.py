"""Drop-in for the `diff_surfel_rasterization` module the reference imports
(gs2dgs/gaussian_renderer/__init__.py:14): same two names, backed by the gfx950 HIP library."""
from scorp_amd.rasterizer2d import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401
