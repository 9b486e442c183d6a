"""Timing-only probe builds of blend_backward_wave_kernel (all-fp32 form; WRONG gradients are possible; never shipped): what binds it -
vector issue, the LDS pipe or the fp32 MFMAs?  python scripts/dev/make_probe3d.py -> build/variants/gs3d_backward_probe.hip;
bash scripts/build_file_variant.sh p3d_X build/variants/gs3d_backward_probe.hip gs3d_backward.hip "-DPROBE_X"
Flags: PROBE_EXTRA_LDS (the two row writes of a hit issued twice), PROBE_EXTRA_VALU / PROBE_EXTRA_SALU (four dependent v_add_f32 /
s_add_u32 more per hit), PROBE_HALF_LDS (only the v row written)."""
import os
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
s = open(os.path.join(root, "scorp_amd/csrc/gs3d_backward.hip")).read()
old = """            if constexpr (kExact) {
              xm[i8 * kXStride + lane] = __float_as_uint(v);
              xm[(8 + i8) * kXStride + lane] = __float_as_uint(w);"""
new = """            if constexpr (kExact) {
              xm[i8 * kXStride + lane] = __float_as_uint(v);
#ifndef PROBE_HALF_LDS
              xm[(8 + i8) * kXStride + lane] = __float_as_uint(w);
#else
              asm volatile("" ::"v"(w));
#endif
#ifdef PROBE_EXTRA_LDS
              { volatile uint32_t *x2 = xm; x2[i8 * kXStride + lane] = __float_as_uint(v); x2[(8 + i8) * kXStride + lane] = __float_as_uint(w); }
#endif
#ifdef PROBE_EXTRA_VALU
              { float z_ = v; asm volatile("v_add_f32 %0, %0, %0\\n\\tv_add_f32 %0, %0, %0\\n\\tv_add_f32 %0, %0, %0\\n\\tv_add_f32 %0, %0, %0" : "+v"(z_)); }
#endif
#ifdef PROBE_EXTRA_SALU
              { int z_ = i8; asm volatile("s_add_u32 %0, %0, 1\\n\\ts_add_u32 %0, %0, 1\\n\\ts_add_u32 %0, %0, 1\\n\\ts_add_u32 %0, %0, 1" : "+s"(z_) : : "scc"); }
#endif"""
assert old in s
s = s.replace(old, new, 1)
out = os.path.join(root, "build/variants/gs3d_backward_probe.hip")
open(out, "w").write(s)
print(out)
