"""CPU: the C-ABI shared library loads and exports every symbol include/yabpe.h declares; without a GPU the
entry points fail loudly instead of falling back to anything."""
from __future__ import annotations

import ctypes
import re
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "yet-another-bpe_amd" / "csrc"


@pytest.fixture(scope="module")
def lib_path():
    so = CSRC / "libyabpe.so"
    if not so.exists():
        subprocess.check_call(["make", "-C", str(CSRC)])
    return so


def declared_symbols() -> list[str]:
    text = (REPO / "include" / "yabpe.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yabpe_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib_path):
    from yet_another_bpe import _native

    names = declared_symbols()
    assert len(names) >= 18
    assert sorted(_native.SYMBOLS) == names  # the Python binding covers the whole header
    lib = ctypes.CDLL(str(lib_path))
    for n in names:
        assert hasattr(lib, n), n
    lib.yabpe_abi_version.restype = ctypes.c_int
    assert lib.yabpe_abi_version() == 2


def test_no_device_fails_loudly(lib_path):
    from yet_another_bpe import _native

    L = _native.lib()
    if L.yabpe_device_count() > 0:
        pytest.skip("a GPU is present; the no-device error path is checked on CPU-only hosts")
    with pytest.raises(_native.YabpeError) as e:
        _native.Context(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig

    with pytest.raises(_native.YabpeError):  # the product path does not silently compute on the CPU
        BBPETrainer(BBPETrainerConfig(vocab_size=300))._merge_loop([[65, 66], [65, 66]])


def test_product_package_never_imports_oracle():
    pkg = REPO / "yet-another-bpe_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        assert "oracle" not in p.read_text().lower().replace("# oracle-free", ""), p
