"""Spherical-harmonics helpers of the python colour branch (gs3dgs/utils/sh_utils.py:26-118), torch, any device."""
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


# degree 4: (constant, monomials in x, y, z and their exponents) of the nine real harmonics of band l = 4, m = -4 .. 4.  Each
# entry is `constant * sum_k coefficient_k * x^a_k y^b_k z^c_k`; the polynomials are the published ones (x^2 + y^2 + z^2 = 1 used
# where the reference's table uses it: 7 z^2 - 1, 7 z^2 - 3, 35 z^4 - 30 z^2 + 3).
_BAND4 = (
    (2.5033429417967046, ((1.0, 3, 1, 0), (-1.0, 1, 3, 0))),                      # x y (x^2 - y^2)
    (-1.7701307697799304, ((3.0, 2, 1, 1), (-1.0, 0, 3, 1))),                     # y z (3 x^2 - y^2)
    (0.9461746957575601, ((7.0, 1, 1, 2), (-1.0, 1, 1, 0))),                      # x y (7 z^2 - 1)
    (-0.6690465435572892, ((7.0, 0, 1, 3), (-3.0, 0, 1, 1))),                     # y z (7 z^2 - 3)
    (0.10578554691520431, ((35.0, 0, 0, 4), (-30.0, 0, 0, 2), (3.0, 0, 0, 0))),   # 35 z^4 - 30 z^2 + 3
    (-0.6690465435572892, ((7.0, 1, 0, 3), (-3.0, 1, 0, 1))),                     # x z (7 z^2 - 3)
    (0.47308734787878004, ((7.0, 2, 0, 2), (-1.0, 2, 0, 0), (-7.0, 0, 2, 2), (1.0, 0, 2, 0))),   # (x^2 - y^2)(7 z^2 - 1)
    (-1.7701307697799304, ((1.0, 3, 0, 1), (-3.0, 1, 2, 1))),                     # x z (x^2 - 3 y^2)
    (0.6258357354491761, ((1.0, 4, 0, 0), (-6.0, 2, 2, 0), (1.0, 0, 4, 0))),      # x^4 - 6 x^2 y^2 + y^4
)


def _band4(sh, x, y, z):
    """Contribution of the band l = 4 (coefficients 16 .. 24) - gs3dgs/utils/sh_utils.py:101-112 as a table of monomials."""
    out = 0.0
    for k, (c, terms) in enumerate(_BAND4):
        poly = 0.0
        for coef, a, b, cz in terms:
            poly = poly + coef * (x ** a) * (y ** b) * (z ** cz)
        out = out + c * poly * sh[..., 16 + k]
    return out


def eval_sh(deg, sh, dirs):
    """sh[..., C, (deg+1)^2], unit dirs[..., 3] -> [..., C]; degrees 0..4 (the rasterizers take 0..3; the fourth band is the
    python branch's only: sh_utils.py:101-112)."""
    assert 0 <= deg <= 4 and sh.shape[-1] >= (deg + 1) ** 2
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                      + C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10]
                          + C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                          + C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14]
                          + C3[6] * x * (xx - 3 * yy) * sh[..., 15])
                if deg > 3:
                    result = result + _band4(sh, x, y, z)
    return result


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
