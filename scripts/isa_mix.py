"""Static instruction mix of the two blend kernels' hot blocks -> profiles/valu_mix.json.

    python scripts/isa_mix.py          (needs hipcc; no GPU)

Compiles gs3d_forward.hip / gs3d_backward.hip with -save-temps, takes each blend kernel's largest basic block (the
straight-line full group: 16 hits in both), counts its wave64 VALU instructions by class and
prices them with the per-class issue costs MEASURED on MI355X by scripts/mb_valu_peak.hip
(profiles/r02_mb_valu_peak.txt, 8 waves per SIMD, every SIMD busy, cycles at 2.4 GHz per wave-instruction per SIMD):
VOP2 2.8, VOP3 (three-source fma / mix) 3.3, v_cmp and v_cndmask 4.15, transcendental 8.5.  An MFMA holds the SIMD's
vector issue for 8 cycles (MI355X_MICROARCH.md, "vector-instruction ISSUE cost") and occupies the matrix pipe - which
runs beside the other waves' VALU work - for 16 (16x16x32) or 32 (32x32x16) cycles: `mfma_cycles_per_hit` is the issue
hold, `matrix_pipe_cycles_per_hit` the pipe occupancy (far below the VALU figure: not the limiter).  bench.py multiplies
the per-hit issue cost by the hits of the view: the time the kernel would take if the VALU issued back to back
("instruction-mix VALU roofline"), and reports measured / that.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scorp_amd.build import ARCH, CSRC, HIPCC, source_sha  # noqa: E402

COST = {"vop2": 2.8, "vop3": 3.3, "cmp": 4.15, "cnd": 4.15, "trans": 8.5}
TRANS = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")


def cat(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_cndmask"):
        return "cnd"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_"):
        return "vop2" if op.endswith("_e32") else "vop3"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def kernel_blocks(asm, symbol_re):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(symbol_re, l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], ["entry", []]
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
            continue
        t = l.strip()
        if t and not t.startswith((";", ".")):
            cur[1].append(t.split()[0])
    blocks.append(cur)
    return blocks


def mfma_pipe_cycles(op):
    return 32.0 if "32x32" in op else 16.0


def hot_block(src, symbol_re, hits, trans_per_hit, mfma_32x32=0):
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-c",
                               f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", os.path.join(CSRC, src), "-o",
                               os.path.join(td, "x.o"), "-save-temps=obj"], stderr=subprocess.DEVNULL)
        asm = open(next(os.path.join(td, f) for f in os.listdir(td) if f.endswith(f"{ARCH}.s"))).read()
    # The full group exists twice since round 3 (with and without the alpha clamp, see blend_group / process_group): of the
    # blocks within 40 % of the largest (round 4: the clamped form also carries the `power > 0` guard of indefinite conics and
    # is a quarter longer), the one with the fewest v_min_f32 is the clamp-free form nearly every group takes.
    blocks = kernel_blocks(asm, symbol_re)
    biggest = max(len(b[1]) for b in blocks)
    name, ops = min((b for b in blocks if len(b[1]) >= 0.6 * biggest), key=lambda b: sum(o.startswith("v_min_f32") for o in b[1]))
    c = collections.Counter(cat(o) for o in ops)
    # (the forward's mid-group "all pixels saturated?" test cuts its group in two blocks; the scheduler leaves the first
    # half's exponentials in the block in front, with the MFMAs: they are counted here so that a hit has all of its own)
    hoisted = max(0, trans_per_hit * hits - c["trans"])
    c["trans"] += hoisted
    if c["mfma"] == 0 and mfma_32x32:   # (... and so are the group's exponent MFMAs)
        ops = ops + ["v_mfma_f32_32x32x16_bf16"] * mfma_32x32
        c["mfma"] = mfma_32x32
    valu = sum(COST[k] * c[k] for k in COST)
    return {"block": name, "hits_per_block": hits, "transcendentals_counted_from_the_preceding_block": hoisted,
            "instructions": dict(sorted(c.items())),
            "valu_insts_per_hit": round(sum(c[k] for k in COST) / hits, 2),
            "valu_cycles_per_hit": round(valu / hits, 2), "mfma_cycles_per_hit": round(8.0 * c["mfma"] / hits, 2),
            "matrix_pipe_cycles_per_hit": round(sum(mfma_pipe_cycles(o) for o in ops if o.startswith("v_mfma")) / hits, 2)}


out = {"_source": "scripts/isa_mix.py; issue costs from profiles/r02_mb_valu_peak.txt (cycles at 2.4 GHz per wave-instruction per SIMD)",
       "source_sha": source_sha(), "costs": COST,
       "blend_forward": hot_block("gs3d_forward.hip", r"^_ZN5scorp12_GLOBAL__N_125blend_forward_wave_kernelILb1E.*:", 16, 1, 3),
       "blend_backward": hot_block("gs3d_backward.hip", r"^_ZN5scorp12_GLOBAL__N_126blend_backward_wave_kernelILb0ELb0E.*:", 16, 2)}
json.dump(out, open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
