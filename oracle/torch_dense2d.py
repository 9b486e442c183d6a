"""Dense per-pixel PyTorch formulation of the 2DGS (surfel) render, differentiated by torch.autograd.

TEST INFRASTRUCTURE ONLY: cross-checks the analytic backward of oracle/gs2d_oracle.c on tiny cases.  Same constants
and semantics as the C oracle; the places where the published backward is not the derivative of the forward are
mirrored explicitly (alpha clamp passes the gradient straight through; the screen-centre used by the low-pass branch
is differentiated through the 3-sigma box formula; the means2D output is a densification statistic, not a
gradient, and is not compared here).
"""
import torch

from .torch_dense import quat_to_rot, sh_to_rgb

TILE = 16
NEAR, FAR = 0.2, 100.0


def render_dense_2d(means3D, opacities, view, proj, campos, bg, W, H, shs=None, sh_degree=0, colors_precomp=None,
                    scales=None, rotations=None, scale_modifier=1.0):
    dt = means3D.dtype
    N = means3D.shape[0]
    view, proj = view.reshape(4, 4).to(dt), proj.reshape(4, 4).to(dt)
    Pm = proj.T                                             # maths matrix
    Q = torch.stack([0.5 * W * Pm[0] + 0.5 * (W - 1) * Pm[3], 0.5 * H * Pm[1] + 0.5 * (H - 1) * Pm[3], Pm[3]])  # [3,4]
    R = quat_to_rot(rotations)                              # [N,3,3]
    s = scale_modifier * scales
    ones = torch.ones(N, 1, dtype=dt)
    zeros = torch.zeros(N, 1, dtype=dt)
    h0 = torch.cat([R[:, :, 0] * s[:, 0:1], zeros], 1)
    h1 = torch.cat([R[:, :, 1] * s[:, 1:2], zeros], 1)
    h2 = torch.cat([means3D, ones], 1)
    Hm = torch.stack([h0, h1, h2], 2)                       # [N,4,3]
    T = Q[None] @ Hm                                        # [N,3,3] rows Tu,Tv,Tw
    Tu, Tv, Tw = T[:, 0], T[:, 1], T[:, 2]
    pv = (torch.cat([means3D, ones], 1) @ view)[:, :3]
    nv = R[:, :, 2] @ view[:3, :3]                          # Wrot @ tn with Wrot = view[:3,:3].T
    cosv = -(pv * nv).sum(1)
    mult = torch.where(cosv > 0, 1.0, -1.0).to(dt).detach()
    nrm = nv * mult[:, None]
    t = torch.tensor([9.0, 9.0, -1.0], dtype=dt)
    dd = (t * Tw * Tw).sum(1)
    f = t[None] / dd[:, None]
    cx, cy = (f * Tu * Tw).sum(1), (f * Tv * Tw).sum(1)
    with torch.no_grad():
        ex = torch.sqrt(torch.clamp_min(cx * cx - (f * Tu * Tu).sum(1), 1e-4))
        ey = torch.sqrt(torch.clamp_min(cy * cy - (f * Tv * Tv).sum(1), 1e-4))
        radius = torch.ceil(torch.maximum(torch.maximum(ex, ey), torch.tensor(3.0 * 0.707106, dtype=dt)))
        tiles_x, tiles_y = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
        tr = lambda v: torch.trunc(v).to(torch.int64)
        x0 = tr((cx - radius) / TILE).clamp(0, tiles_x); y0 = tr((cy - radius) / TILE).clamp(0, tiles_y)
        x1 = tr((cx + radius + TILE - 1) / TILE).clamp(0, tiles_x); y1 = tr((cy + radius + TILE - 1) / TILE).clamp(0, tiles_y)
        valid = (pv[:, 2] > NEAR) & (cosv != 0) & (dd != 0) & ((x1 - x0) * (y1 - y0) > 0)
    if colors_precomp is None:
        d = means3D - campos.to(dt)[None]
        rgb = torch.clamp_min(sh_to_rgb(sh_degree, shs, d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    else:
        rgb = colors_precomp
    radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)
    idx = torch.nonzero(valid).squeeze(1)
    order = idx[torch.sort(pv[idx, 2].detach(), stable=True).indices]
    Tu, Tv, Tw, o, nrm_s, rgb_s = Tu[order], Tv[order], Tw[order], opacities.reshape(-1)[order], nrm[order], rgb[order]
    cxs, cys = cx[order], cy[order]
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxs, pys = xs.reshape(-1, 1, 1).to(dt), ys.reshape(-1, 1, 1).to(dt)           # [P,1,1]
    tix, tiy = xs.reshape(-1, 1) // TILE, ys.reshape(-1, 1) // TILE
    in_tile = (tix >= x0[order][None]) & (tix < x1[order][None]) & (tiy >= y0[order][None]) & (tiy < y1[order][None])
    k = pxs * Tw[None] - Tu[None]                                                  # [P,n,3]
    l = pys * Tw[None] - Tv[None]
    p = torch.cross(k, l, dim=-1)
    pz_ok = p[..., 2] != 0
    pz = torch.where(pz_ok, p[..., 2], torch.ones_like(p[..., 2]))
    s0, s1 = p[..., 0] / pz, p[..., 1] / pz
    rho3d = s0 * s0 + s1 * s1
    dx, dy = cxs[None] - pxs[..., 0], cys[None] - pys[..., 0]
    rho2d = 2.0 * (dx * dx + dy * dy)
    use3d = rho3d <= rho2d
    rho = torch.where(use3d, rho3d, rho2d)
    depth = torch.where(use3d, s0 * Tw[None, :, 0] + s1 * Tw[None, :, 1] + Tw[None, :, 2], Tw[None, :, 2].expand_as(s0))
    power = -0.5 * rho
    G = torch.exp(power)
    araw = o[None] * G
    alpha = araw + (torch.clamp_max(araw, 0.99) - araw).detach()
    contrib = in_tile & pz_ok & (depth >= NEAR) & (power <= 0) & (alpha >= 1.0 / 255.0)
    alpha = torch.where(contrib, alpha, torch.zeros_like(alpha))
    depth = torch.where(contrib, depth, torch.ones_like(depth))
    Tincl = torch.cumprod(1 - alpha, dim=1)
    Texcl = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], 1)
    with torch.no_grad():
        stop = contrib & (Tincl < 0.0001)
        alive = torch.cumsum(stop.to(torch.int64), 1) == 0
    act = contrib & alive
    w = alpha * Texcl * alive
    T_final = torch.prod(1 - alpha * alive, dim=1)
    color = (w @ rgb_s) + T_final[:, None] * bg.to(dt)[None]
    normal = w @ nrm_s
    exp_depth = (w * depth).sum(1)
    acc = 1 - T_final
    m = FAR / (FAR - NEAR) * (1 - NEAR / depth)
    # distortion = sum_i w_i (m_i^2 A_i + M2_i - 2 m_i M1_i) with prefix sums over earlier contributors
    wm, wm2 = w * m, w * m * m
    M1 = torch.cumsum(wm, 1) - wm
    M2 = torch.cumsum(wm2, 1) - wm2
    A = 1 - Texcl
    dist = (w * (m * m * A + M2 - 2 * m * M1)).sum(1)
    # median depth: depth of the last contributor whose incoming T is > 0.5
    with torch.no_grad():
        cand = act & (Texcl > 0.5)
        pos = torch.arange(cand.shape[1])[None].expand_as(cand)
        last = torch.where(cand, pos, torch.full_like(pos, -1)).max(1).values
    med = torch.where(last >= 0, depth.gather(1, last.clamp_min(0)[:, None])[:, 0], torch.zeros_like(exp_depth))
    allmap = torch.stack([exp_depth, acc, normal[:, 0], normal[:, 1], normal[:, 2], med, dist], 0).reshape(7, H, W)
    return color.T.reshape(3, H, W), radii, allmap
