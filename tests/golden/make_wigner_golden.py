"""Golden table for the SH rotation: real Wigner-D blocks D_1, D_2, D_3 for eight rotations, generated
INDEPENDENTLY of scorp_amd.transforms.sh_rotation_blocks (a least-squares fit on sampled directions) by two methods that
must agree with each other to 1e-10 before the file is written:

  A. the Ivanic-Ruedenberg recurrence (J. Phys. Chem. 100 (1996) 6342, with the 1998 errata): D_l from D_{l-1} and the
     l = 1 block, which is the rotation matrix itself in the (y, z, x) ordering of the real harmonics m = -1, 0, 1;
  B. orthogonal projection by quadrature: D_ij = integral of Y_i(d) Y_j(R^-1 d) over the sphere (Gauss-Legendre in
     cos(theta) x uniform in phi, exact for these polynomial degrees), with Y the 3DGS basis polynomials written out here
     (gs3dgs/utils/sh_utils.py:24-55 constants; they carry a (-1)^m sign relative to the standard real harmonics).

Convention: coefficients of the rotated function f'(d) = f(R^-1 d) are c'_l = D_l c_l - what
utils/gaussians.py:64-108 obtains from e3nn (absent here) after its axis permutation P.  For l = 1 this is
D_1 = S (P R P^T) S with S = diag(-1, 1, -1): the permuted rotation itself, up to the basis signs.

    python tests/golden/make_wigner_golden.py   ->  tests/golden/wigner_d.npz
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def basis(d):
    """3DGS real SH basis, degrees 1..3, at unit directions d[n,3] -> [n,15] (sh_utils.py:68-100)."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    return np.stack([
        -C1 * y, C1 * z, -C1 * x,
        C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy),
        C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
        C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)], 1)


def by_quadrature(R, nt=24, nph=48):
    t, w = np.polynomial.legendre.leggauss(nt)
    ph = (np.arange(nph) + 0.5) * 2 * np.pi / nph
    st = np.sqrt(1 - t * t)
    d = np.stack([np.outer(st, np.cos(ph)), np.outer(st, np.sin(ph)), np.outer(t, np.ones(nph))], -1).reshape(-1, 3)
    wt = np.outer(w, np.full(nph, 2 * np.pi / nph)).reshape(-1)
    Y0, Y1 = basis(d), basis(d @ R)          # Y(d), Y(R^-1 d)   (d @ R = (R^T d)^T)
    out = []
    for l in (1, 2, 3):
        sl = slice(l * l - 1, (l + 1) * (l + 1) - 1)
        M = (Y1[:, sl] * wt[:, None]).T @ Y0[:, sl]       # M_ji = <Y_j(R^-1 .), Y_i> : Y_j(R^-1 d) = sum_i M_ji Y_i(d)
        out.append(M.T)                                    # c' = M^T c
    return out


def by_recurrence(R):
    """Standard real harmonics (no (-1)^m) by Ivanic-Ruedenberg, then the 3DGS signs."""
    perm = [1, 2, 0]                                        # m = -1, 0, 1  <->  y, z, x
    R1 = R[np.ix_(perm, perm)]
    mats = [None, R1]
    g1 = lambda i, j: R1[i + 1, j + 1]
    for l in (2, 3):
        prev = mats[l - 1]
        gp = lambda a, b: prev[a + l - 1, b + l - 1]

        def P(i, a, b):
            if b == l:
                return g1(i, 1) * gp(a, l - 1) - g1(i, -1) * gp(a, -l + 1)
            if b == -l:
                return g1(i, 1) * gp(a, -l + 1) + g1(i, -1) * gp(a, l - 1)
            return g1(i, 0) * gp(a, b)

        M = np.zeros((2 * l + 1, 2 * l + 1))
        for m in range(-l, l + 1):
            for n in range(-l, l + 1):
                d0 = 1.0 if m == 0 else 0.0
                den = (2.0 * l) * (2.0 * l - 1) if abs(n) == l else (l + n) * (l - n)
                u = np.sqrt((l + m) * (l - m) / den)
                v = 0.5 * np.sqrt((1 + d0) * (l + abs(m) - 1) * (l + abs(m)) / den) * (1 - 2 * d0)
                w = -0.5 * np.sqrt((l - abs(m) - 1) * (l - abs(m)) / den) * (1 - d0)
                U = P(0, m, n) if u != 0 else 0.0
                if v == 0:
                    V = 0.0
                elif m == 0:
                    V = P(1, 1, n) + P(-1, -1, n)
                elif m > 0:
                    d1 = 1.0 if m == 1 else 0.0
                    V = P(1, m - 1, n) * np.sqrt(1 + d1) - P(-1, -m + 1, n) * (1 - d1)
                else:
                    d1 = 1.0 if m == -1 else 0.0
                    V = P(1, m + 1, n) * (1 - d1) + P(-1, -m - 1, n) * np.sqrt(1 + d1)
                if w == 0:
                    Wv = 0.0
                elif m > 0:
                    Wv = P(1, m + 1, n) + P(-1, -m - 1, n)
                else:
                    Wv = P(1, m - 1, n) - P(-1, -m + 1, n)
                M[m + l, n + l] = u * U + v * V + w * Wv
        mats.append(M)
    out = []
    for l in (1, 2, 3):
        S = np.diag([(-1.0) ** m for m in range(-l, l + 1)])
        out.append(S @ mats[l] @ S)
    return out


def main():
    rots = np.load(os.path.join(HERE, "rotations_128.npz"))["rotations"].astype(np.float64)
    rng = np.random.default_rng(7)
    Rs = [rots[i] for i in (0, 5, 37, 77, 101, 127)]
    for _ in range(2):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        Rs.append(np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]))
    D = {1: [], 2: [], 3: []}
    for R in Rs:
        U_, _, Vt = np.linalg.svd(R)
        R = U_ @ Vt                                          # the npz rotations are float32: re-orthonormalise in float64
        a, b = by_recurrence(R), by_quadrature(R)
        for l in (1, 2, 3):
            assert np.abs(a[l - 1] - b[l - 1]).max() < 1e-10, (l, np.abs(a[l - 1] - b[l - 1]).max())
            assert np.abs(a[l - 1] @ a[l - 1].T - np.eye(2 * l + 1)).max() < 1e-10      # orthogonal
            D[l].append(a[l - 1])
    np.savez(os.path.join(HERE, "wigner_d.npz"), rotations=np.stack([np.linalg.svd(R)[0] @ np.linalg.svd(R)[2] for R in Rs]),
             D1=np.stack(D[1]), D2=np.stack(D[2]), D3=np.stack(D[3]))
    print("wrote wigner_d.npz:", len(Rs), "rotations")


if __name__ == "__main__":
    main()
