#!/usr/bin/env python3
"""PARKED EXPERIMENT (round 5; not in the default build: gs3d_forward.hip includes the generated header only with -DSCORP_FWD_MERGE=1).

Generates scorp_amd/csrc/blend_group_asm.hpp: the blend forward's full clamp-free 16-hit group as ONE inline-assembly block in
which ADJACENT hits whose live pixels are disjoint share an iteration.

    python scripts/dev/gen_blend_group_asm.py && python scripts/build_ab.sh ... "-DSCORP_FWD_MERGE=1" gs3d_forward.hip

RESULT (one MI355X, S3, same box): images, hit lists, n_contrib bit-identical (129 / 129 tests of tests/test_gs3d_gpu.py); 27.7 % of the
iterations merged; SQ_INSTS_VALU 101.4 M -> 93.1 M per launch, SQ_INSTS_SALU 37.0 M -> 46.7 M (the pairing's bookkeeping and one
branch per slot), SQ_WAVE_CYCLES 459.6 M -> 462.3 M: blend_forward_wave_kernel<true> 161.1 -> 161.4 us, <false> 133.1 -> 129.4 us.
The kernel is bound by instructions ISSUED per wave (vector and scalar alike: ~11 cycles per instruction per wave at six waves
per SIMD), not by the vector datapath: trading eight vector instructions for ten scalar ones buys nothing.

Why assembly: written in C++ (scripts/dev/blend_forward_merged_pairs.hip.txt) the compiler gives the accumulators of the merged
and the unmerged path different registers and reconciles them with v_mov at every join - the instructions the merging saves come
back as copies, and at the kernel's 80-register cap it spills.  Here every value has one register for the whole group.

Per slot i (a label each; `pair` bit i = hit i takes hit i + 1 along):
    single : ds_read_b128 col | v_exp | al = live ? g : 0 | test_T = T - al T | ok = test_T >= 1e-4 | ae = ok ? al : 0 | w = ae T |
             [took = ok & live: kept bit, last contributor] | T = ok ? test_T : -|T| | 4 x fmac                      12 VALU
    merged : + the selects of exponent, ring slot and (backward bookkeeping) the second list position                15 VALU
The live masks are sixteen v_cmp against kExp2AlphaMinBits (common.hpp) whose results stay in SGPR pairs.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "scorp_amd", "csrc", "blend_group_asm.hpp")
COL = "v[76:79]"            # the ring slot's (r, g, b, depth): a fixed, clobbered tuple (an asm operand's sub-registers cannot be named)
CX, CY, CZ, CW = "v76", "v77", "v78", "v79"


def body(fb):
    L = []
    A = L.append
    m = lambda i: f"%[m{i}]"
    e = lambda i: f"%[e{i}]"
    # ---- live masks, overlaps, greedy pairing ----
    for i in range(15):
        A(f"v_cmp_le_f32_e64 {m(i)}, %[thr], {e(i)}")
    A("s_mov_b32 %[ov], 0")
    for i in range(14, -1, -1):          # bit i of ov = hits i and i + 1 share a live pixel
        A(f"s_and_b64 %[st0], {m(i)}, {m(i + 1)}")
        A("s_addc_u32 %[ov], %[ov], %[ov]")
    A("s_andn2_b32 %[ov], 0x7fff, %[ov]")                 # d: mergeable positions
    A("s_lshl_b32 %[t0], %[ov], 1")
    A("s_andn2_b32 %[t0], %[ov], %[t0]")                  # run starts
    A("s_and_b32 %[t0], %[t0], 0x5555")                   # ... at even positions
    A("s_add_u32 %[t0], %[ov], %[t0]")
    A("s_andn2_b32 %[t0], %[ov], %[t0]")                  # runs that start at an even position
    A("s_and_b32 %[t1], %[t0], 0x5555")
    A("s_andn2_b32 %[t0], %[ov], %[t0]")
    A("s_and_b32 %[t0], %[t0], 0xaaaa")
    A("s_or_b32 %[pair], %[t0], %[t1]")
    if fb:
        A("s_mov_b32 %[kept], 0")
    A("s_mov_b32 %[done], 0")

    def pipeline(live, i, merged):
        A(f"v_cndmask_b32_e64 %[g], 0, %[g], {live}")                     # al
        A("v_fma_f32 %[tt], -%[g], %[T], %[T]")                           # test_T
        A("v_cmp_le_f32_e32 vcc, %[tmin], %[tt]")                         # ok
        A("v_cndmask_b32_e32 %[g], 0, %[g], vcc")                         # ae
        A("v_mul_f32_e32 %[w], %[g], %[T]")
        if fb:
            if merged:
                A(f"s_and_b64 %[st1], vcc, {live}")                       # took
                A(f"v_cndmask_b32_e64 %[lastg], %[lastg], {i + 1}, %[st1]")
                A(f"s_and_b64 %[st1], %[st1], {m(i + 1)}")                # ... the second hit
                A(f"v_cndmask_b32_e64 %[lastg], %[lastg], {i + 2}, %[st1]")
                A(f"s_and_b64 %[st1], vcc, {m(i)}")
                A("s_addc_u32 %[kept], %[kept], %[kept]")
                A(f"s_and_b64 %[st1], vcc, {m(i + 1)}")
                A("s_addc_u32 %[kept], %[kept], %[kept]")
            else:
                A(f"s_and_b64 %[st1], vcc, {live}")
                A("s_addc_u32 %[kept], %[kept], %[kept]")
                A(f"v_cndmask_b32_e64 %[lastg], %[lastg], {i + 1}, %[st1]")
        A("v_cndmask_b32_e64 %[T], -|%[T]|, %[tt], vcc")
        A("s_waitcnt lgkmcnt(0)")
        A(f"v_fmac_f32_e32 %[C0], {CX}, %[w]")
        A(f"v_fmac_f32_e32 %[C1], {CY}, %[w]")
        A(f"v_fmac_f32_e32 %[C2], {CZ}, %[w]")
        A(f"v_fmac_f32_e32 %[Dp], {CW}, %[w]")

    for i in range(16):
        A(f"Lslot{i}_%=:")
        if i == 8:                                                        # every pixel saturated: the second half is not needed
            A("v_cmp_lt_f32_e32 vcc, 0, %[T]")
            A("s_cmp_eq_u64 vcc, 0")
            A("s_cbranch_scc1 Lhalf_%=")
        if i < 15:
            A(f"s_bitcmp1_b32 %[pair], {i}")
            A(f"s_cbranch_scc1 Lmerge{i}_%=")
        # (gfx950: a VALU instruction may not read a transcendental's result in the very next issue slot - the compiler inserts
        # the wait state itself, in inline assembly it is ours: the ring read sits between the v_exp and its first use)
        A(f"v_exp_f32_e32 %[g], {e(i)}")
        A(f"ds_read_b128 {COL}, %[gcb] offset:{16 * i}")
        pipeline(m(i), i, False)
        if i < 15:
            A(f"s_branch Lslot{i + 1}_%=")
    for i in range(15):
        A(f"Lmerge{i}_%=:")
        A(f"v_cndmask_b32_e64 %[addr], %[gcb], %[gcb16], {m(i + 1)}")
        A(f"ds_read_b128 {COL}, %[addr] offset:{16 * i}")
        A(f"v_cndmask_b32_e64 %[es], {e(i)}, {e(i + 1)}, {m(i + 1)}")
        A("v_exp_f32_e32 %[g], %[es]")
        A(f"s_or_b64 %[st0], {m(i)}, {m(i + 1)}")
        pipeline("%[st0]", i, True)
        A(f"s_branch Lslot{i + 2}_%=" if i + 2 < 16 else "s_branch Lend_%=")
    # (slot 15 falls through to here)
    L_slot15_end = None
    return L


def reorder(lines):
    """Single slots fall through to the next slot (no s_branch needed between consecutive slots): the generator above emits
    `s_branch Lslot{i+1}` right before `Lslot{i+1}:` for singles; drop those."""
    out = []
    for k, ln in enumerate(lines):
        if ln.startswith("s_branch Lslot") and k + 1 < len(lines) and lines[k + 1] == ln.replace("s_branch ", "") + ":":
            continue
        out.append(ln)
    return out


def emit_fn(fb):
    lines = reorder(body(fb))
    # the merged blocks must not be fallen into: after slot 15's pipeline jump over them
    idx = lines.index("Lmerge0_%=:")
    lines.insert(idx, "s_branch Lend_%=")
    lines.append("Lhalf_%=:")
    lines.append("s_mov_b32 %[done], 1")
    if fb:
        lines.append("s_brev_b32 %[kept], %[kept]")
        lines.append("s_lshr_b32 %[kept], %[kept], 24")
        lines.append("s_branch Lout_%=")
    lines.append("Lend_%=:")
    if fb:
        lines.append("s_brev_b32 %[kept], %[kept]")
        lines.append("s_lshr_b32 %[kept], %[kept], 16")
        lines.append("Lout_%=:")
    text = "\n".join(f'      "{ln}\\n"' for ln in lines)
    name = "blend_group_merged_asm_fb" if fb else "blend_group_merged_asm_img"
    outs = ['[T] "+v"(T)', '[C0] "+v"(C0)', '[C1] "+v"(C1)', '[C2] "+v"(C2)', '[Dp] "+v"(Dp)']
    if fb:
        outs += ['[lastg] "+v"(lastg)', '[kept] "=&s"(kept)']
    outs += ['[done] "=&s"(done)', '[es] "=&v"(es)', '[g] "=&v"(g)', '[tt] "=&v"(tt)', '[w] "=&v"(w)', '[addr] "=&v"(addr)']
    outs += [f'[m{i}] "=&s"(m[{i}])' for i in range(15)]
    outs += ['[st0] "=&s"(st0)', '[st1] "=&s"(st1)', '[ov] "=&s"(ov)', '[pair] "=&s"(pair)', '[t0] "=&s"(t0)', '[t1] "=&s"(t1)']
    ins = [f'[e{i}] "v"(e[{i}])' for i in range(16)]
    ins += ['[gcb] "v"(gcb)', '[gcb16] "v"(gcb16)', '[thr] "s"(thr)', '[tmin] "s"(tmin)', '[m15] "s"(m15)']
    sig_fb = ", uint32_t &lastg, uint32_t &kept16" if fb else ""
    return f'''// returns true if every pixel of the block was saturated after the first eight hits (the second half was skipped)
__device__ __forceinline__ bool {name}(const f32x16 &e, uint32_t gcb, float &T, float &C0, float &C1, float &C2, float &Dp{sig_fb}) {{
  const float thr = __uint_as_float(kExp2AlphaMinBits), tmin = kTMin;
  // (the first read of the exponent MFMAs' result is the compiler's: it owns the matrix-core -> VALU wait states)
  const uint64_t m15 = __builtin_amdgcn_ballot_w64(e[15] >= thr);
  const uint32_t gcb16 = gcb + 16u;
  uint64_t m[15], st0, st1;
  uint32_t ov, pair, t0, t1, done, addr{", kept" if fb else ""};
  float es, g, tt, w;
  asm volatile(
{text}
      : {", ".join(outs)}
      : {", ".join(ins)}
      : "v76", "v77", "v78", "v79", "vcc", "scc", "memory");
{"  kept16 = kept;" if fb else ""}
  return done != 0;
}}
'''


def main():
    hdr = '''// blend_group_asm.hpp - GENERATED by scripts/gen_blend_group_asm.py (edit the generator, not this file).
// The blend forward's full clamp-free group of sixteen hits as one inline-assembly block; see the generator's docstring and
// profiles/DESIGN_history_r01-r05.md section 8.0 (2).  gfx950 only.
#pragma once
#include "common.hpp"
#include "exp_mfma.hpp"

namespace scorp {
#ifdef __HIPCC__
'''
    txt = hdr + emit_fn(True) + "\n" + emit_fn(False) + "#endif\n}  // namespace scorp\n"
    txt = txt.replace("%[m15]", "%[m15]")
    open(OUT, "w").write(txt)
    print("wrote", OUT, len(txt.splitlines()), "lines")


if __name__ == "__main__":
    main()
