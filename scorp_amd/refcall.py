"""The rasterizers called the way the reference calls them (gs3dgs/gaussian_renderer/__init__.py:51-66,101-111 and
gs2dgs/gaussian_renderer/__init__.py:51-67,111-120): activated inputs as leaf tensors, a zero `means2D` gradient sink,
the 12-field settings record, one call.  `kw` is the argument dictionary the parity harnesses share with the CPU checker
(numpy float32 arrays + scalars: means3D, opacities, shs | colors_precomp, scales + rotations | cov3D_precomp, W, H,
tanfovx, tanfovy, view, proj, campos, bg, sh_degree, scale_modifier).  Used by bench.py's parity leg and by tests/.
"""
import torch


def _leaves(kw, dev, requires_grad, names):
    T = lambda a, rg=False: None if a is None else torch.tensor(a, device=dev, requires_grad=rg and requires_grad)
    N = kw["means3D"].shape[0]
    t = {n: T(kw.get(n), True) for n in names}
    t["opacities"] = T(kw["opacities"].reshape(N, 1), True)
    means2D = torch.zeros(N, 3, device=dev, requires_grad=requires_grad)
    return T, t, means2D


def _settings(cls, kw, T, debug):
    return cls(image_height=kw["H"], image_width=kw["W"], tanfovx=kw["tanfovx"], tanfovy=kw["tanfovy"], bg=T(kw["bg"]),
               scale_modifier=kw.get("scale_modifier", 1.0), viewmatrix=T(kw["view"]), projmatrix=T(kw["proj"]),
               sh_degree=kw.get("sh_degree", 0), campos=T(kw["campos"]), prefiltered=False, debug=debug)


def render3d_reference_call(kw, dev, requires_grad=True, debug=False):
    """-> ((color, radii, depth, alpha), leaves) through diff_gaussian_rasterization.GaussianRasterizer."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    T, t, means2D = _leaves(kw, dev, requires_grad, ("means3D", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"))
    out = GaussianRasterizer(raster_settings=_settings(GaussianRasterizationSettings, kw, T, debug))(
        means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["shs"], colors_precomp=t["colors_precomp"],
        scales=t["scales"], rotations=t["rotations"], cov3D_precomp=t["cov3D_precomp"])
    t["means2D"] = means2D
    return out, t


def render2d_reference_call(kw, dev, requires_grad=True):
    """-> ((color, radii, allmap), leaves) through diff_surfel_rasterization.GaussianRasterizer."""
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    T, t, means2D = _leaves(kw, dev, requires_grad, ("means3D", "shs", "colors_precomp", "scales", "rotations", "transmat_precomp"))
    out = GaussianRasterizer(raster_settings=_settings(GaussianRasterizationSettings, kw, T, False))(
        means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["shs"], colors_precomp=t["colors_precomp"],
        scales=t["scales"], rotations=t["rotations"], cov3D_precomp=t["transmat_precomp"])   # (the reference's name for the [N, 9] transforms)
    t["means2D"] = means2D
    return out, t
