"""Generate tests/golden/ref_sh_deg4.npz by IMPORTING the reference's eval_sh (gs3dgs/utils/sh_utils.py:57-112) at degree 4.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden_sh4.py
Only inputs and outputs are stored.  (A file of its own, so that ref_helpers.npz stays the bytes round 1 committed.)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from gs3dgs.utils.sh_utils import eval_sh  # noqa: E402

rng = np.random.default_rng(20261004)
sh = rng.normal(0, 0.5, (48, 3, 25)).astype(np.float32)
dirs = rng.normal(0, 1, (48, 3)).astype(np.float32)
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
out = {"sh": sh, "dirs": dirs}
for deg in range(5):
    out[f"rgb_deg{deg}"] = eval_sh(deg, torch.tensor(sh), torch.tensor(dirs)).numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_sh_deg4.npz"), **out)
print("wrote ref_sh_deg4.npz", {k: v.shape for k, v in out.items()})
