"""Randomised parity sweep (not part of the test suite): HIP vs the CPU oracle on random sizes / scales / degrees, 3DGS
and 2DGS, forward + backward.  `python scripts/fuzz_parity.py [n_cases] [seed] [3d|2d]` on a GPU box."""
import math
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gs3d_gpu import compare_forward, compare_grads, hip_render, oracle  # noqa: E402
from tests.util import image_weights, make_case  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    surfels = len(sys.argv) > 3 and sys.argv[3] == "2d"
    bad = 0
    for c in range(n_cases):
        if surfels:
            from tests import test_gs2d_gpu as T2
            case = dict(N=int(rng.integers(50, 6000)), W=int(rng.integers(9, 300)), H=int(rng.integers(9, 220)),
                        deg=int(rng.integers(0, 4)), seed=int(rng.integers(0, 1 << 30)),
                        log_scale=float(rng.uniform(math.log(0.004), math.log(0.3))))
            if rng.random() < 0.3:
                case["radius"] = float(rng.uniform(1.5, 6.0))
            if rng.random() < 0.3:
                case["scale_modifier"] = float(rng.uniform(0.5, 1.8))
            try:
                T2.CASES["fuzz"] = case
                T2.test_forward_backward_parity_2d("fuzz", dev)
                print(f"case {c:3d} ok   {case}", flush=True)
            except Exception as e:
                bad += 1
                print(f"case {c:3d} FAIL {case}\n{''.join(traceback.format_exception_only(type(e), e))}", flush=True)
            continue
        case = dict(N=int(rng.integers(1, 6000)), W=int(rng.integers(9, 300)), H=int(rng.integers(9, 220)),
                    deg=int(rng.integers(0, 4)), seed=int(rng.integers(0, 1 << 30)),
                    log_scale=float(rng.uniform(math.log(0.003), math.log(0.5))))
        if rng.random() < 0.3:
            case["radius"] = float(rng.uniform(0.5, 6.0))
        if rng.random() < 0.3:
            case["scale_modifier"] = float(rng.uniform(0.3, 2.0))
        if rng.random() < 0.3:
            case["bg"] = tuple(float(v) for v in rng.random(3))
        try:
            kw, _ = make_case(**case)
            o = oracle(kw)
            out, t = hip_render(kw, dev)
            compare_forward(out, o)
            wc, wd, wa = image_weights(kw["H"], kw["W"], case["seed"])
            color, _, depth, alpha = out
            loss = (color * torch.tensor(wc, device=dev)).sum() + (depth * torch.tensor(wd, device=dev)).sum() + \
                (alpha * torch.tensor(wa, device=dev)).sum()
            loss.backward()
            compare_grads(t, o.backward(wc, wd, wa))
            print(f"case {c:3d} ok   {case}", flush=True)
        except Exception as e:   # keep going: report every failing configuration
            bad += 1
            print(f"case {c:3d} FAIL {case}\n{''.join(traceback.format_exception_only(type(e), e))}", flush=True)
    print(f"{n_cases - bad}/{n_cases} cases passed")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
