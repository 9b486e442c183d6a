"""Closed-form rigid / similarity fits between corresponding 3-D point sets, on whatever device the points live on —
the solvers the alignment stage runs inside its RANSAC loop (utils/solution.py:8-86: `kabsch_algorithm_np`,
`umeyama_algorithm_np`).  Same contract: `target ~ s * R @ source + t`, returns (R[3,3], t[3], s).

Batched: inputs may carry leading batch dimensions ([..., n, 3]); the SVD of the 3x3 covariance runs in float64.
"""
import torch


def _fit(source, target, with_scale):
    if source.shape != target.shape or source.shape[-1] != 3:
        raise ValueError("Source and target points must have the same [..., n, 3] shape")
    if source.shape[-2] == 0:
        raise ValueError("Empty point sets")
    P, Q = source.double(), target.double()
    cp, cq = P.mean(-2, keepdim=True), Q.mean(-2, keepdim=True)
    Pc, Qc = P - cp, Q - cq
    H = Pc.transpose(-1, -2) @ Qc                                  # covariance, source^T target
    U, S, Vt = torch.linalg.svd(H)
    d = torch.sign(torch.linalg.det(U @ Vt))                       # reflection guard (right-handed result)
    d = torch.where(d == 0, torch.ones_like(d), d)
    D = torch.ones_like(S)
    D[..., -1] = d
    R = (Vt.transpose(-1, -2) * D.unsqueeze(-2)) @ U.transpose(-1, -2)
    if with_scale:
        s = (S * D).sum(-1) / (Pc ** 2).sum((-1, -2))
    else:
        s = torch.ones_like(S[..., 0])
    t = cq.squeeze(-2) - s.unsqueeze(-1) * (R @ cp.transpose(-1, -2)).squeeze(-1)
    return R, t, s


def kabsch(source, target):
    """Rotation + translation minimising the RMS deviation (utils/solution.py:8-39); scale is 1."""
    return _fit(source, target, with_scale=False)


def umeyama(source, target):
    """Rotation + translation + uniform scale (utils/solution.py:42-86)."""
    return _fit(source, target, with_scale=True)
