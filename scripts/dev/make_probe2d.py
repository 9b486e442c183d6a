"""Timing-only probe builds of blend2d_backward_wave_kernel (WRONG gradients; never shipped): a copy of gs2d.hip with sections of the
hit loop switchable off by -D flags, to see what each section adds to the launch time at four waves per SIMD.

    python scripts/dev/make_probe2d.py            -> build/variants/gs2d_probe.hip
    bash scripts/build_file_variant.sh p2d_noatomic build/variants/gs2d_probe.hip gs2d.hip "-DPROBE_NO_ATOMIC"
    gpurun -- 'AB_BENCH_ARGS="--scene S6" bash scripts/ab_kernel.sh TAG blend2d_backward default build/variants/libp2d_noatomic.so ...'

Flags: PROBE_NO_ATOMIC (the read-out lanes keep their value, no atomic), PROBE_NO_READOUT (no result-tile write / read-out /
atomic), PROBE_NO_MFMA (no A-operand reads, no MFMAs), PROBE_NO_FLUSH (the pair is dropped), PROBE_NO_WMAX (the per-hit scale is a
constant: no wave maximum), PROBE_HALF_WRITES / PROBE_DOUBLE_WRITES (four / sixteen of the eight matrix-row writes per hit,
same arithmetic), PROBE_NO_SPLIT (raw bits written: the sixteen split instructions gone, same LDS traffic), PROBE_EXTRA_VALU / PROBE_EXTRA_SALU (eight dependent v_add_f32 / s_add_u32 more per live hit),
PROBE_EVAL_ONLY (every hit ends after its ballot).
"""
import os
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
src = open(os.path.join(root, "scorp_amd/csrc/gs2d.hip")).read()


def rep(old, new, count=1):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, count)


# the atomic
rep("        atomicAdd(acc + (size_t)q_id[sl] * kAcc2Stride + (lane & 31), v);",
    "#ifdef PROBE_NO_ATOMIC\n        asm volatile(\"\" ::\"v\"(v));\n#else\n"
    "        atomicAdd(acc + (size_t)q_id[sl] * kAcc2Stride + (lane & 31), v);\n#endif")
# the read-out (result tile write, read, atomic)
rep("    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\");\n    __builtin_amdgcn_wave_barrier();   // every lane has its A operands",
    "#ifdef PROBE_NO_READOUT\n    asm volatile(\"\" ::\"v\"(d[0]), \"v\"(d[1]), \"v\"(d[2]), \"v\"(d[3]));\n    pend = 0;\n    return;\n#endif\n"
    "    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\");\n    __builtin_amdgcn_wave_barrier();   // every lane has its A operands")
# the MFMAs and their operand reads
rep("    f32x4_2d d = {0.0f, 0.0f, 0.0f, 0.0f};\n    if constexpr (kExact) {",
    "    f32x4_2d d = {0.0f, 0.0f, 0.0f, 0.0f};\n#ifdef PROBE_NO_MFMA\n    if constexpr (false) {\n#else\n    if constexpr (kExact) {\n#endif")
rep("    } else {\n      Frag2 af[4];", "    } else if constexpr (\n#ifdef PROBE_NO_MFMA\n        false\n#else\n        true\n#endif\n    ) {\n      Frag2 af[4];")
# the whole flush
rep("  auto flush_pair = [&]() {\n", "  auto flush_pair = [&]() {\n#ifdef PROBE_NO_FLUSH\n    pend = 0;\n    return;\n#endif\n")
# the per-hit wave maximum
rep("        const uint32_t eb = wave_max_u32(__float_as_uint(fmaxf(fmaxf(fabsf(t), fabsf(dL_dz)), w))) >> 23;",
    "#ifdef PROBE_NO_WMAX\n        const uint32_t eb = 120u;\n#else\n"
    "        const uint32_t eb = wave_max_u32(__float_as_uint(fmaxf(fmaxf(fabsf(t), fabsf(dL_dz)), w))) >> 23;\n#endif")
# only half of the matrix rows written (all eight values still computed), no fp16 split (raw bits written), every row written twice
rep("        rowp[0] = term(dp0); rowp[k2XStride] = term(dp1);\n        rowp[2 * k2XStride] = term(dp2); rowp[3 * k2XStride] = term(zr);\n"
    "        rowp[4 * k2XStride] = term(z2); rowp[5 * k2XStride] = term(t2);\n        rowp[6 * k2XStride] = term(t); rowp[7 * k2XStride] = term(w * kWCarry);",
    "#if defined(PROBE_HALF_WRITES)\n"
    "        rowp[0] = term(dp0); rowp[k2XStride] = term(dp1); rowp[2 * k2XStride] = term(dp2); rowp[3 * k2XStride] = term(zr);\n"
    "        { uint32_t a_ = term(z2), b_ = term(t2), c_ = term(t), d_ = term(w * kWCarry); asm volatile(\"\" ::\"v\"(a_), \"v\"(b_), \"v\"(c_), \"v\"(d_)); }\n"
    "#elif defined(PROBE_DOUBLE_WRITES)\n"
    "        rowp[0] = term(dp0); rowp[k2XStride] = term(dp1); rowp[2 * k2XStride] = term(dp2); rowp[3 * k2XStride] = term(zr);\n"
    "        rowp[4 * k2XStride] = term(z2); rowp[5 * k2XStride] = term(t2); rowp[6 * k2XStride] = term(t); rowp[7 * k2XStride] = term(w * kWCarry);\n"
    "        { volatile uint32_t *r2 = rowp; r2[0] = term(dp0); r2[k2XStride] = term(dp1); r2[2 * k2XStride] = term(dp2); r2[3 * k2XStride] = term(zr);\n"
    "          r2[4 * k2XStride] = term(z2); r2[5 * k2XStride] = term(t2); r2[6 * k2XStride] = term(t); r2[7 * k2XStride] = term(w * kWCarry); }\n"
    "#else\n"
    "        rowp[0] = term(dp0); rowp[k2XStride] = term(dp1);\n        rowp[2 * k2XStride] = term(dp2); rowp[3 * k2XStride] = term(zr);\n"
    "        rowp[4 * k2XStride] = term(z2); rowp[5 * k2XStride] = term(t2);\n        rowp[6 * k2XStride] = term(t); rowp[7 * k2XStride] = term(w * kWCarry);\n"
    "#endif")
rep("        auto term = [](float x) { return kExact ? __float_as_uint(x) : split_one(x); };",
    "#ifdef PROBE_NO_SPLIT\n        auto term = [](float x) { return __float_as_uint(x); };\n#else\n"
    "        auto term = [](float x) { return kExact ? __float_as_uint(x) : split_one(x); };\n#endif")
# eight extra independent vector (or scalar) instructions per live hit
rep("      float inv_sg = 1.0f;\n      if constexpr (!kExact) {   // the hit's power-of-two scale",
    "#ifdef PROBE_EXTRA_VALU\n      { float z_ = qxb; asm volatile(\"v_add_f32 %0, %0, %0\\n\\tv_add_f32 %0, %0, %0\\n\\tv_add_f32 %0, %0, %0\\n\\tv_add_f32 %0, %0, %0\\n\\t\"\n"
    "                                 \"v_add_f32 %0, %0, %0\\n\\tv_add_f32 %0, %0, %0\\n\\tv_add_f32 %0, %0, %0\\n\\tv_add_f32 %0, %0, %0\" : \"+v\"(z_)); }\n#endif\n"
    "#ifdef PROBE_EXTRA_SALU\n      { int z_ = s_; asm volatile(\"s_add_u32 %0, %0, 1\\n\\ts_add_u32 %0, %0, 1\\n\\ts_add_u32 %0, %0, 1\\n\\ts_add_u32 %0, %0, 1\\n\\t\"\n"
    "                               \"s_add_u32 %0, %0, 1\\n\\ts_add_u32 %0, %0, 1\\n\\ts_add_u32 %0, %0, 1\\n\\ts_add_u32 %0, %0, 1\" : \"+s\"(z_) : : \"scc\"); }\n#endif\n"
    "      float inv_sg = 1.0f;\n      if constexpr (!kExact) {   // the hit's power-of-two scale")
# eval only
rep("      if (live == 0) continue;\n      // The per-pixel recurrence runs under `valid`",
    "      if (live == 0) continue;\n#ifdef PROBE_EVAL_ONLY\n      asm volatile(\"\" ::\"v\"(h.s0), \"v\"(h.s1), \"v\"(h.depth), \"v\"(h.Go), \"v\"(h.rdepth));\n      continue;\n#endif\n"
    "      // The per-pixel recurrence runs under `valid`")

out = os.path.join(root, "build/variants/gs2d_probe.hip")
os.makedirs(os.path.dirname(out), exist_ok=True)
open(out, "w").write(src)
print(out)
