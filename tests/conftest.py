import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a GPU")
    from cobevt_amd import lib
    lib.load()  # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The ten parity comparisons closest to their gates (tests/util.py GATE_RATIOS)."""
    try:
        from util import GATE_RATIOS
    except Exception:
        return
    if not GATE_RATIOS:
        return
    tr = terminalreporter
    tr.section("parity gate headroom: the 10 comparisons closest to their gates (measured / gate)")
    for ratio, norm, measured, gate, what, tid in sorted(GATE_RATIOS, key=lambda t: -t[0])[:10]:
        tr.write_line("%.3f  %s-rel %.3e / %.2e  %s  [%s]" % (ratio, norm, measured, gate, what, tid))
