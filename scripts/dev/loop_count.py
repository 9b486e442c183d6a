"""Static instruction count of a kernel's innermost (Depth=2) loop region: python scripts/dev/loop_count.py FILE.s SYMBOL_FRAGMENT..."""
import sys


def loop_stats(path, frag):
    lines = open(path).read().splitlines()
    st = next(i for i, l in enumerate(lines) if frag in l.split(":")[0] and l.startswith("_Z") and ":" in l)
    en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    body = lines[st:en]
    idx = [i for i, l in enumerate(body) if "Depth=2" in l]
    lo, j = min(idx), max(idx) + 1
    while j < len(body) and not body[j].startswith(".LBB"):
        j += 1
    c = {}
    for l in body[lo:j]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")):
            continue
        op = t[0]
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "nop" if op.startswith("s_nop") else "salu" if op.startswith("s_")
             else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global", "buffer", "flat")) else "other")
        c[k] = c.get(k, 0) + 1
    return c


if __name__ == "__main__":
    for frag in sys.argv[2:]:
        print(frag, loop_stats(sys.argv[1], frag))
