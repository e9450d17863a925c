import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes tens of seconds on CPU")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def record_measurement(test, **values):
    """Append the measured parity figures of a GPU test to gpurun_out/parity_measured.jsonl (so that gates can be stated as a small
    multiple of what was measured, and DESIGN.md can quote them)."""
    import json
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_measured.jsonl"), "a") as f:
        f.write(json.dumps({"test": test, **values}) + "\n")
