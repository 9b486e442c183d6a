"""Timing-only probe for the per-Gaussian forward at five waves per SIMD: the LDS image of a block's SH rows
is HALVED - the 48 one-KiB chunks of the linear layout land at chunk (c % 24), rows are read modulo the half buffer - so the
kernel moves the same bytes and runs the same instructions with 25 KB of LDS per workgroup instead of 50 (occupancy 3 -> 5
waves per SIMD, its 82 registers allow no more).  The colours it produces are WRONG; only its duration means anything.
    python scripts/dev/build_pre_probe.py && gpurun -- 'bash scripts/ab_kernel.sh TAG preprocess_kernel default build/variants/libpre_probe.so'
Result (profiles/README.md, r05): 61.9 / 62.2 us -> 62.0 / 61.9 us: occupancy is not what bounds this kernel."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = os.path.join(ROOT, "scorp_amd", "csrc")
out = os.path.join(ROOT, "build", "variants", "probe")
os.makedirs(out, exist_ok=True)
src = open(os.path.join(C, "gs3d_pergaussian.hip")).read()
hpp = open(os.path.join(C, "pergaussian.hpp")).read()
h2 = hpp.replace("(LPtr)(l4 + c * 64), 16, 0, 0)", "(LPtr)(l4 + (c % 24) * 64), 16, 0, 0)")
h2 = h2.replace("(LPtr)(lr4 + (c - 3) * 64), 16, 0, 0)", "(LPtr)(l4 + (c % 24) * 64), 16, 0, 0)")
h2 = h2.replace("r.p0 = lds + 3 * t; r.pr = lds + kShLinearRest + 45 * t - 3;", "r.p0 = lds + (3 * t) % 6000; r.pr = lds + (kShLinearRest + 45 * t - 3) % 6000;")
assert h2.count("% 24") == 2 and "% 6000" in h2
a = "int32_t *__restrict__ radii, uint32_t *__restrict__ tile_count) {\n  __shared__ __attribute__((aligned(16))) float s_sh[256 * kShStride];"
assert src.count(a) == 1      # the forward kernel only (the backward's declaration differs)
open(os.path.join(out, "pergaussian.hpp"), "w").write(h2)
open(os.path.join(out, "gs3d_pergaussian.hip"), "w").write(src.replace(a, a.replace("256 * kShStride", "128 * kShStride")))
hipcc = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC"]
subprocess.check_call(hipcc + [f"-I{ROOT}/include", f"-I{out}", f"-I{C}", "-c", os.path.join(out, "gs3d_pergaussian.hip"), "-o", os.path.join(out, "pg_probe.o")])
others = [os.path.join(ROOT, "build", f) for f in sorted(os.listdir(os.path.join(ROOT, "build"))) if f.endswith(".hip.o") and "gs3d_pergaussian" not in f]
lib = os.path.join(ROOT, "build", "variants", "libpre_probe.so")
subprocess.check_call(hipcc + ["-shared", "-o", lib, os.path.join(out, "pg_probe.o")] + others)
print(lib)
