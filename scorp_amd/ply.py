"""Gaussian PLY codec — the on-disk interchange between every SCORP stage.

Writes / reads the exact vertex layout of `GaussianModel.save_ply` / `load_ply` / `load_multi_ply`
(gs3dgs/scene/gaussian_model.py:220-251, 287-410): binary little-endian, float32 properties in the order
x y z nx ny nz f_dc_0..2 f_rest_0..(3(K-1)-1) opacity scale_0.. rot_0..3, with f_dc / f_rest flattened channel-major
(`transpose(1, 2)`), i.e. files interchange with the reference (which uses the `plyfile` package — not required here).
"""
import os

import numpy as np

_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
              "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
              "uint": "u4", "uint32": "u4"}


def attribute_names(n_dc, n_rest, n_scale, n_rot=4):
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return names


def write_ply(path, xyz, f_dc, f_rest, opacity, scaling, rotation):
    """Arrays in the GaussianModel layout: xyz[N,3], f_dc[N,1,3], f_rest[N,K-1,3], opacity[N,1], scaling[N,S], rotation[N,4]."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    xyz = np.asarray(xyz, np.float32)
    n = xyz.shape[0]
    dc = np.asarray(f_dc, np.float32).transpose(0, 2, 1).reshape(n, -1)
    rest = np.asarray(f_rest, np.float32).transpose(0, 2, 1).reshape(n, -1)
    cols = np.concatenate([xyz, np.zeros_like(xyz), dc, rest, np.asarray(opacity, np.float32).reshape(n, 1),
                           np.asarray(scaling, np.float32), np.asarray(rotation, np.float32)], axis=1)
    names = attribute_names(dc.shape[1], rest.shape[1], np.asarray(scaling).shape[1], np.asarray(rotation).shape[1])
    assert cols.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {n}\n" + \
        "".join(f"property float {nm}\n" for nm in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(cols, dtype="<f4").tobytes())


def read_ply_vertices(path):
    """Structured array of the first element of a binary-little-endian or ascii PLY (all scalar properties)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex, seen_element = None, 0, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = not seen_element
                seen_element = True
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "binary_little_endian":
            dt = np.dtype([(n, "<" + t) for n, t in props])
            return np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            out = np.empty(count, dtype=[(n, t) for n, t in props])
            for k, (n, _) in enumerate(props):
                out[n] = data[:, k]
            return out
        raise ValueError(f"{path}: unsupported PLY format {fmt}")


def read_gaussian_ply(path, max_sh_degree=None):
    """-> dict(xyz, features_dc[N,1,3], features_rest[N,K-1,3], opacity[N,1], scaling[N,S], rotation[N,4]) as float32."""
    v = read_ply_vertices(path)
    names = v.dtype.names
    n = v.shape[0]
    col = lambda nm: np.asarray(v[nm], np.float32)
    xyz = np.stack([col("x"), col("y"), col("z")], 1)
    dc = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], 1).reshape(n, 3, 1)
    by_index = lambda prefix: sorted([nm for nm in names if nm.startswith(prefix)], key=lambda s: int(s.split("_")[-1]))
    rest_names = by_index("f_rest_")
    if max_sh_degree is not None:
        want = 3 * (max_sh_degree + 1) ** 2 - 3
        rest = np.zeros((n, want), np.float32)
        for k, nm in enumerate(rest_names[:want]):
            rest[:, k] = col(nm)
    else:
        rest = np.stack([col(nm) for nm in rest_names], 1) if rest_names else np.zeros((n, 0), np.float32)
    rest = rest.reshape(n, 3, -1)
    scaling = np.stack([col(nm) for nm in by_index("scale_")], 1)
    rotation = np.stack([col(nm) for nm in by_index("rot")], 1)
    return dict(xyz=xyz, features_dc=np.ascontiguousarray(dc.transpose(0, 2, 1)),
                features_rest=np.ascontiguousarray(rest.transpose(0, 2, 1)), opacity=col("opacity").reshape(n, 1),
                scaling=scaling, rotation=rotation)
