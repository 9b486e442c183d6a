"""`simple_knn._C.distCUDA2` on the gfx950 HIP library (C ABI: scorp_knn_dist2)."""
import ctypes

import torch

from scorp_amd import _C as _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points[N,3] (GPU, fp32) -> [N] mean squared distance to the 3 nearest neighbours."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 needs a GPU tensor (scorp_amd has no CPU path)")
    pts = points.contiguous().float()
    out = torch.empty(pts.shape[0], dtype=torch.float32, device=pts.device)
    L = _lib.lib()
    _lib.check(L.scorp_knn_dist2(ctypes.c_void_p(pts.data_ptr()), pts.shape[0], ctypes.c_void_p(out.data_ptr()),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "scorp_knn_dist2")
    return out
