"""The C-ABI library loads and exports every symbol include/scorp_gs.h declares (no compute: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from scorp_amd import build
    return build.build()


def header_functions():
    txt = open(os.path.join(ROOT, "include", "scorp_gs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(scorp_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    fns = header_functions()
    for f in ("scorp_gs3d_preprocess", "scorp_gs3d_num_pairs", "scorp_gs3d_render", "scorp_gs3d_backward",
              "scorp_last_error", "scorp_version"):
        assert f in fns


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for f in header_functions():
        assert hasattr(lib, f), f"{f} declared in include/scorp_gs.h but not exported by libscorp_gs.so"


def test_binding_lists_every_declared_symbol(built_lib):
    from scorp_amd import _C
    assert sorted(_C.EXPORTS) == header_functions()
    L = _C.lib()
    assert L.scorp_version() >= 100
    # workspace sizing is pure host arithmetic
    s = L.scorp_gs3d_state_bytes(1_000_000, 1600, 1200)
    assert 64 * 1_000_000 <= s < 128 * 1_000_000   # records + bin + tile masks + pixel state + block histograms
    assert L.scorp_gs3d_pairs_bytes(1_500_000) >= 12 * 1_500_000
    assert L.scorp_gs3d_backward_scratch_bytes(1000) >= 48 * 1000


def test_struct_layout_matches_header():
    from scorp_amd import _C
    assert ctypes.sizeof(_C.ScorpGs3dInputs) == 40 + 12 * 8 + 8   # 10 ints/floats, 12 pointers, raw_params + padding
    assert ctypes.sizeof(_C.ScorpGs3dGrads) == 9 * 8
    assert ctypes.sizeof(_C.ScorpGs3dTrainView) == 13 * 8 + 2 * 4 + 9 * 8   # see the struct in include/scorp_gs.h
    assert ctypes.sizeof(_C.ScorpFusedAdam) == 12 * 8 + 8 * 4 + 3 * 8 + 2 * 4 + 4 * 8


def test_header_is_plain_c(tmp_path):
    """include/scorp_gs.h must be consumable from C (the boundary is a C ABI), and the structs there must have the
    sizes the ctypes mirror assumes."""
    import os, subprocess
    from scorp_amd import _C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include "scorp_gs.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", '
                   'sizeof(ScorpGs3dInputs), sizeof(ScorpGs3dGrads), sizeof(ScorpGs3dTrainView), sizeof(ScorpGs2dTrainView), '
                   'sizeof(ScorpFusedAdam)); return 0; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    a, b, c, d, e = (int(x) for x in subprocess.check_output([str(exe)]).split())
    assert (a, b, c, d, e) == (ctypes.sizeof(_C.ScorpGs3dInputs), ctypes.sizeof(_C.ScorpGs3dGrads), ctypes.sizeof(_C.ScorpGs3dTrainView),
                               ctypes.sizeof(_C.ScorpGs2dTrainView), ctypes.sizeof(_C.ScorpFusedAdam))


def test_shim_packages_expose_reference_names(built_lib):
    import diff_gaussian_rasterization as dgr
    fields = dgr.GaussianRasterizationSettings._fields
    assert fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                      "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    r = dgr.GaussianRasterizer(raster_settings=None)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=None, means2D=None, opacities=None, shs=None, colors_precomp=None, scales=1, rotations=1)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=None, means2D=None, opacities=None, shs=1, colors_precomp=None, scales=1, rotations=1, cov3D_precomp=1)
