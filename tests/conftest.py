"""pytest configuration: path setup + the `gpu` marker.

`python -m pytest tests -m "not gpu"` runs everywhere (oracle vs golden vectors, host logic, ABI surface);
`python -m pytest tests -m gpu` needs an MI355X and exercises the HIP path through the C ABI.
GPU tests never skip themselves: on a box without a GPU / without libyabpe.so they fail loudly.
"""
from __future__ import annotations

import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "yet-another-bpe_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return REPO / "tests" / "golden"
