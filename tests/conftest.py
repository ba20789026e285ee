"""Shared fixtures.  GPU tests are marked ``gpu``; everything else runs on a CPU-only box."""
import json
import os
import sys
from pathlib import Path

import pytest

# Some GPU tests put their inputs in HBM with torch.  PyTorch-ROCm brings its own copy of the HIP
# runtime: it has to be the first one loaded into the process (torch sees no GPU when libmercat_hip.so
# pulled in /opt/rocm's copy before it), so load it before any test touches the engine.
try:
    import torch  # noqa: F401
except ImportError:  # CPU-only environments without torch still run the host tests
    torch = None

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def expected():
    return json.loads((GOLDEN / "expected.json").read_text())


@pytest.fixture(scope="session")
def chunk_golden():
    return json.loads((GOLDEN / "chunks.json").read_text())


@pytest.fixture(scope="session")
def inputs_dir():
    return GOLDEN / "inputs"


def read_input(name: str) -> bytes:
    """Decompressed bytes of a golden input."""
    import gzip
    p = GOLDEN / "inputs" / name
    return gzip.open(p, "rb").read() if name.endswith(".gz") else p.read_bytes()
