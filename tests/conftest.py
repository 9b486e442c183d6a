import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_helpers.npz"))


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import gs_oracle
    gs_oracle.build()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Tally of tests.util.assert_grad_close: gradient tensors checked / tensors that needed the oracle's own band."""
    from tests.util import BAND_TALLY
    if BAND_TALLY["checked"]:
        cap = int(os.environ.get("SCORP_BAND_CAP", "14"))
        terminalreporter.write_line(f"assert_grad_close: {BAND_TALLY['checked']} gradient tensors checked, "
                                    f"{BAND_TALLY['fallback']} needed the oracle's own band in the default / split-form tests, "
                                    f"{BAND_TALLY['fallback_exact_fp32']} in the exact_fp32 ones (cap {cap} each)")
        for n in BAND_TALLY["names"]:
            terminalreporter.write_line("  band fallback: " + n)
        worst = {}
        for n, c in BAND_TALLY.get("beyond", []):
            worst[n] = max(worst.get(n, 0), c)
        if worst:
            terminalreporter.write_line("  elements beyond the max-norm tolerance after the band (largest count per tensor name): "
                                        + ", ".join(f"{n} {c}" for n, c in sorted(worst.items())))


def pytest_sessionfinish(session, exitstatus):
    from tests.util import BAND_TALLY
    cap = int(os.environ.get("SCORP_BAND_CAP", "14"))
    if max(BAND_TALLY["fallback"], BAND_TALLY["fallback_exact_fp32"]) > cap and session.exitstatus == 0:
        session.exitstatus = 1      # too many tensors passed only through the band: treated as a failed session
