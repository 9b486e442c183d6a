"""S6 (config #5) through bench.py's own secondary record, alone: both backward forms, the per-kernel table.
    python scripts/dev/time_s6.py [--parity]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    rec = bench.secondary_s6(dev, parity="--parity" in sys.argv)
    print(json.dumps(rec))
