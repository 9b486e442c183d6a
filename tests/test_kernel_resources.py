"""Register / LDS budgets of the blend kernels, checked at compile time (no GPU).

The four blend kernels are VALU-bound and run one wave per 8x8 block; how many waves a SIMD holds is set by a kernel's
registers (512 per lane per SIMD, granularity 8) and by its LDS (160 KB per CU, four SIMDs), and DESIGN.md sections 4.2,
4.3 and 4.7 record what each step of occupancy was worth (2DGS forward: three -> five waves per SIMD, 298 -> 272 us).  A
change that pushes a kernel over its budget costs that silently; this test says so.  It compiles the sources with the
library's own flags to assembly and reads the resource summary the compiler prints per kernel.
"""
import os
import re
import subprocess
import tempfile

import pytest

from scorp_amd.build import ARCH, CSRC, HIPCC, ROOT

# kernel-name fragment -> (max VGPRs, max scratch bytes, max LDS bytes, waves per SIMD that budget buys)
BUDGETS = {
    "gs3d_forward.hip": {"blend_forward_wave_kernelILb1E": (80, 16, 160 * 1024 // 24, 6), "blend_forward_wave_kernelILb0E": (80, 16, 160 * 1024 // 24, 6)},
    "gs3d_backward.hip": {"blend_backward_wave_kernel": (128, 32, 160 * 1024 // 16, 4)},
    "gs2d.hip": {"blend2d_forward_wave_kernel": (96, 16, 160 * 1024 // 20, 5), "blend2d_backward_wave_kernel": (128, 0, 160 * 1024 // 16, 4)},
    # (256-thread workgroups: the LDS figure is per workgroup; the unmasked backward is the one every training view runs)
    "loss.hip": {"ssim_l1_forward_strip_kernel": (128, 0, 5120, 4), "ssim_l1_backward_kernelILb0E": (80, 0, 13312, 6)},
}


def _resources(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fno-slp-vectorize", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
               "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        text = open(out).read()
    res, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^; (NumVgprs|ScratchSize|LDSByteSize|Occupancy): (\d+)", line)
        if m and cur:
            res.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", sorted(BUDGETS))
def test_blend_kernels_keep_their_occupancy(src):
    res = _resources(src)
    for frag, (vgprs, scratch, lds, waves) in BUDGETS[src].items():
        kernels = {k: v for k, v in res.items() if frag in k and not any(f2 != frag and frag in f2 and f2 in k for f2 in BUDGETS[src])}
        assert kernels, f"no kernel matching {frag} in {src}"
        for name, r in kernels.items():
            assert r["NumVgprs"] <= vgprs, f"{name}: {r['NumVgprs']} VGPRs > {vgprs} (fewer than {waves} waves per SIMD)"
            assert r["ScratchSize"] <= scratch, f"{name}: {r['ScratchSize']} bytes of scratch > {scratch} (spills in the blend loop)"
            assert r["LDSByteSize"] <= lds, f"{name}: {r['LDSByteSize']} bytes of LDS > {lds} (fewer than {4 * waves} waves per CU)"
            assert r["Occupancy"] >= waves, f"{name}: the compiler reports {r['Occupancy']} waves per SIMD, {waves} expected"
