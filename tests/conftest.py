import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (float64 PyTorch) is what most of the suite's time goes to.  A GPU box shows all of its hardware threads but
    # grants a cgroup quota of a few cores: PyTorch's default (one thread per visible CPU) then runs the oracle 15-20x slower than
    # the quota allows.  Size the intra-op pool to the cores this process may really use.
    try:
        import torch
        from mliis_amd.hostinfo import usable_cores
        torch.set_num_threads(usable_cores())
    except Exception:       # noqa: BLE001 -- never let a tuning knob break collection
        pass


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "host_logic.json")) as f:
        return json.load(f)
