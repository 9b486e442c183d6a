"""Dense per-pixel PyTorch formulation of the 3DGS render, differentiated by torch.autograd.

TEST INFRASTRUCTURE ONLY.  Its single job is to cross-check the analytic backward of oracle/gs3d_oracle.c
(SURVEY.md §7 step 0): every pixel evaluates every (depth-sorted) Gaussian, so it is O(H*W*N) memory and only
usable for N <= a few thousand at <= 64x64.  Same constants and same semantics as the C oracle, including the
three places where the published algorithm's backward is NOT the derivative of its forward:
  * alpha = min(0.99, o*G) passes the gradient straight through the clamp;
  * the FoV guard clamps t.x/t.z in the EWA Jacobian and treats the clamped value as a constant;
  * the conic backward divides by det^2 + 1e-7 (not mirrored here; relative effect ~1e-8).
"""
import math

import torch

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, sh, dirs):
    """sh[N,K,3], dirs[N,3] unit -> rgb[N,3]; restates gs3dgs/utils/sh_utils.py:57-112 in [N,K,3] layout."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
             + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
             + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
             + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def render_dense(means3D, opacities, view, proj, campos, bg, W, H, tanfovx, tanfovy, shs=None, sh_degree=0,
                 colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0,
                 means2D=None):
    """Returns color[3,H,W], radii[N], depth[1,H,W], alpha[1,H,W] (depth un-normalised), all differentiable."""
    dt = means3D.dtype
    N = means3D.shape[0]
    view = view.reshape(4, 4).to(dt)
    proj = proj.reshape(4, 4).to(dt)
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    t = (ph @ view)[:, :3]
    hom = ph @ proj
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]
    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).view(-1, 3, 3)
    else:
        L = quat_to_rot(rotations) * (scale_modifier * scales)[:, None, :]
        Sigma = L @ L.transpose(1, 2)
    tz = t[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    txc = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz).detach(), t[:, 0])
    tyc = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz).detach(), t[:, 1])
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * txc / (tz * tz), zero, fy / tz, -fy * tyc / (tz * tz)], 1).view(-1, 2, 3)
    Wrot = view[:3, :3].T
    M = J @ Wrot
    cov2 = M @ Sigma @ M.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c_ = cov2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    with torch.no_grad():
        mid = 0.5 * (a + c_)
        disc = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + disc, mid - disc)))
    px = ((ndc[:, 0] + 1) * W - 1) * 0.5
    py = ((ndc[:, 1] + 1) * H - 1) * 0.5
    tiles_x, tiles_y = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    with torch.no_grad():
        trunc = lambda v: torch.trunc(v).to(torch.int64)
        x0 = trunc((px - radius) / TILE).clamp(0, tiles_x)
        y0 = trunc((py - radius) / TILE).clamp(0, tiles_y)
        x1 = trunc((px + radius + TILE - 1) / TILE).clamp(0, tiles_x)
        y1 = trunc((py + radius + TILE - 1) / TILE).clamp(0, tiles_y)
        valid = (tz > 0.2) & (det != 0) & ((x1 - x0) * (y1 - y0) > 0)
    if colors_precomp is None:
        d = means3D - campos.to(dt)[None]
        rgb = torch.clamp_min(sh_to_rgb(sh_degree, shs, d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    else:
        rgb = colors_precomp
    radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)

    idx = torch.nonzero(valid).squeeze(1)
    # (depth, index) order: stable sort of index-ordered candidates by depth
    order = idx[torch.sort(tz[idx].detach(), stable=True).indices]
    A, B, C = (c_ / det)[order], (-b / det)[order], (a / det)[order]
    o = opacities.reshape(-1)[order]
    gx, gy, gz, grgb = px[order], py[order], tz[order], rgb[order]
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxs, pys = xs.reshape(-1, 1).to(dt), ys.reshape(-1, 1).to(dt)
    tix, tiy = (xs.reshape(-1, 1) // TILE), (ys.reshape(-1, 1) // TILE)
    in_tile = (tix >= x0[order][None]) & (tix < x1[order][None]) & (tiy >= y0[order][None]) & (tiy < y1[order][None])
    dx, dy = gx[None] - pxs, gy[None] - pys
    power = -0.5 * (A[None] * dx * dx + C[None] * dy * dy) - B[None] * dx * dy
    G = torch.exp(power)
    araw = o[None] * G
    alpha = araw + (torch.clamp_max(araw, 0.99) - araw).detach()
    contrib = in_tile & (power <= 0) & (alpha >= 1.0 / 255.0)
    alpha = torch.where(contrib, alpha, torch.zeros_like(alpha))
    one_m = 1 - alpha
    Tincl = torch.cumprod(one_m, dim=1)
    Texcl = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], 1)
    with torch.no_grad():
        stop = contrib & (Tincl < 0.0001)
        alive = torch.cumsum(stop.to(torch.int64), 1) == 0
    w = alpha * Texcl * alive
    T_final = torch.prod(1 - alpha * alive, dim=1)
    color = (w @ grgb) + T_final[:, None] * bg.to(dt)[None]
    depth = w @ gz
    acc = w.sum(1)
    return color.T.reshape(3, H, W), radii, depth.reshape(1, H, W), acc.reshape(1, H, W)
