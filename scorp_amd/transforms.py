"""Rigid / scale transforms of a GaussianModel — the operations the alignment loop applies between renders
(utils/gaussians.py:12-108: translate, scale, rotate incl. SH rotation).  All on the model's device, no e3nn:
the real-SH rotation blocks are solved from the model's own SH basis (scorp_amd.sh.eval_sh), so they are consistent
with the renderer's conventions by construction.
"""
import torch

from .sh import eval_sh


def quat_multiply(a, b):
    """Hamilton product of (w,x,y,z) quaternions, broadcastable."""
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)


def matrix_to_quat(R):
    """One proper rotation matrix -> (w,x,y,z).  The four candidate formulas (largest of trace / diagonal entries) are
    all evaluated and selected with torch.where, so a device matrix never forces a host synchronisation (the sweep
    captures this in a HIP graph)."""
    R = R.double()
    m00, m11, m22 = R[0, 0], R[1, 1], R[2, 2]
    t = m00 + m11 + m22
    eps = 1e-300

    def cand(s2, w, x, y, z):
        s = torch.sqrt(torch.clamp_min(s2, eps)) * 2
        return torch.stack([w(s), x(s), y(s), z(s)])
    q0 = cand(t + 1.0, lambda s: 0.25 * s, lambda s: (R[2, 1] - R[1, 2]) / s, lambda s: (R[0, 2] - R[2, 0]) / s, lambda s: (R[1, 0] - R[0, 1]) / s)
    q1 = cand(1.0 + m00 - m11 - m22, lambda s: (R[2, 1] - R[1, 2]) / s, lambda s: 0.25 * s, lambda s: (R[0, 1] + R[1, 0]) / s, lambda s: (R[0, 2] + R[2, 0]) / s)
    q2 = cand(1.0 + m11 - m00 - m22, lambda s: (R[0, 2] - R[2, 0]) / s, lambda s: (R[0, 1] + R[1, 0]) / s, lambda s: 0.25 * s, lambda s: (R[1, 2] + R[2, 1]) / s)
    q3 = cand(1.0 + m22 - m00 - m11, lambda s: (R[1, 0] - R[0, 1]) / s, lambda s: (R[0, 2] + R[2, 0]) / s, lambda s: (R[1, 2] + R[2, 1]) / s, lambda s: 0.25 * s)
    q = torch.where(t > 0, q0, torch.where((m00 > m11) & (m00 > m22), q1, torch.where(m11 > m22, q2, q3)))
    return q.float()


def sh_rotation_blocks(R, max_degree=3):
    """[D_1 (3x3), D_2 (5x5), D_3 (7x7)] with  c'_l = D_l c_l  for the rotated function f'(d) = f(R^-1 d)."""
    R = R.detach().double().cpu()
    g = torch.Generator().manual_seed(0)
    d = torch.randn(64, 3, generator=g, dtype=torch.float64)
    d = d / d.norm(dim=1, keepdim=True)
    K = (max_degree + 1) ** 2
    eye = torch.eye(K, dtype=torch.float64)
    # basis values Y_k(d): evaluate with one-hot coefficient vectors
    Y = lambda dirs: torch.stack([eval_sh(max_degree, eye[k].expand(dirs.shape[0], 1, K), dirs)[:, 0] for k in range(K)], 1)
    Y0, Y1 = Y(d), Y(d @ R)            # rows: Y(d_i), Y(R^-1 d_i) (d @ R == (R^T d)^T)
    blocks = []
    for l in range(1, max_degree + 1):
        sl = slice(l * l, (l + 1) * (l + 1))
        M = torch.linalg.lstsq(Y0[:, sl], Y1[:, sl]).solution.T    # Y_l(R^-1 d) = M Y_l(d)
        blocks.append(M.T.float())                                  # c' = M^T c
    return blocks


def _use_kernel(g, *host_args):
    """The one-launch HIP transform serves GPU models whose transform parameters live on the host (what the align loop
    has: numpy rotations / RANSAC results); device-resident parameters (a captured sweep's rotation buffer) keep the
    torch formulation, which needs no host round trip."""
    return g._xyz.is_cuda and all(a is None or not (torch.is_tensor(a) and a.is_cuda) for a in host_args) \
        and not torch.cuda.is_current_stream_capturing() \
        and all(getattr(g, n).data.is_contiguous() and getattr(g, n).dtype == torch.float32
                for n in ("_xyz", "_rotation", "_scaling", "_features_rest"))   # (else: the torch formulation below)


@torch.no_grad()
def gaussians_transform(g, R=None, T=None, scale=None, fix_center=False, blocks=None):
    """x -> ((x - c) R^T) * scale + c + T on a GPU model in ONE launch (scorp_gaussians_transform): positions,
    quaternions, log-scales and the SH bands 1..3 (`blocks`: [D_1, D_2, D_3], default sh_rotation_blocks(R)); c = the
    model's centre if fix_center else 0.  Equal to gaussians_rotate -> gaussians_scale -> gaussians_translate."""
    import ctypes
    from . import _C
    dev = g._xyz.device
    Rh = torch.eye(3, dtype=torch.float64) if R is None else torch.as_tensor(R).detach().double().cpu()
    Th = torch.zeros(3, dtype=torch.float64) if T is None else torch.as_tensor(T).detach().double().cpu().reshape(3)
    Sh = torch.ones(3, dtype=torch.float64) if scale is None else torch.as_tensor(scale).detach().double().cpu().reshape(-1)
    if Sh.numel() == 1:
        Sh = Sh.repeat(3)
    k_rest = int(g._features_rest.shape[1])
    if blocks is None:
        blocks = sh_rotation_blocks(Rh, g.max_sh_degree) if (R is not None and k_rest > 0) else []
    D = [torch.eye(n, dtype=torch.float64) for n in (3, 5, 7)]
    for l, B in enumerate(blocks[:3]):
        D[l] = torch.as_tensor(B).double().cpu()
    q = matrix_to_quat(Rh).double() if R is not None else torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64)
    flat = torch.cat([Rh.reshape(-1), torch.zeros(3, dtype=torch.float64), Th, Sh[:3], q, D[0].reshape(-1), D[1].reshape(-1),
                      D[2].reshape(-1), torch.zeros(3, dtype=torch.float64)]).float()     # 113 floats + padding to 116
    params = flat.to(dev)
    if fix_center:
        params[9:12] = g._xyz.data.mean(0)
    L = _C.lib()
    for name in ("_xyz", "_rotation", "_scaling", "_features_rest"):
        t = getattr(g, name).data
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise RuntimeError(f"gaussians_transform: {name} must be a contiguous float32 tensor")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    # parts a transform does not have are not passed: a translation or a scale leaves the quaternions alone (the kernel
    # would rewrite them normalised), a rotation writes nothing into `_scaling` (which a shallow copy of the model shares)
    rotating, scaling = R is not None, scale is not None
    _C.check(L.scorp_gaussians_transform(p(g._xyz.data), p(g._rotation.data) if rotating else None,
                                         p(g._scaling.data) if scaling else None,
                                         p(g._features_rest.data) if (k_rest and rotating) else None, g._xyz.shape[0], k_rest,
                                         int(g._scaling.shape[1]), p(params), ctypes.c_void_p(_C.current_stream_ptr())),
             "scorp_gaussians_transform")


@torch.no_grad()
def gaussians_translate(g, T):
    if _use_kernel(g, T):
        return gaussians_transform(g, T=T)
    g._xyz.data = g._xyz.data + T[None].to(g._xyz)


@torch.no_grad()
def gaussians_scale(g, scale, fix_center=False):
    if _use_kernel(g, scale) and torch.as_tensor(scale).numel() in (1, 3) and g._scaling.shape[1] == 3:
        return gaussians_transform(g, scale=scale, fix_center=fix_center)
    scale = scale.to(g._xyz)
    if fix_center:
        c = g._xyz.data.mean(0)
        g._xyz.data = (g._xyz.data - c) * scale[None] + c
    else:
        g._xyz.data = g._xyz.data * scale[None]
    g._scaling.data = torch.log(torch.exp(g._scaling.data) * scale[None])


@torch.no_grad()
def gaussians_rotate(g, R, fix_center=False):
    if _use_kernel(g, R):
        return gaussians_transform(g, R=R, fix_center=fix_center)
    R = R.to(g._xyz)
    c = g._xyz.data.mean(0) if fix_center else torch.zeros(3, device=g._xyz.device)
    g._xyz.data = (g._xyz.data - c) @ R.T + c
    q = matrix_to_quat(R).to(g._xyz)
    rot = g._rotation.data
    g._rotation.data = quat_multiply(q[None], rot / rot.norm(dim=1, keepdim=True))
    if g.max_sh_degree > 0:
        blocks = sh_rotation_blocks(R, g.max_sh_degree)
        rest = g._features_rest.data
        for l, D in enumerate(blocks, start=1):
            sl = slice(l * l - 1, (l + 1) * (l + 1) - 1)            # _features_rest starts at coefficient 1
            rest[:, sl] = torch.einsum("ij,njc->nic", D.to(rest), rest[:, sl])
