"""Build-time guard (no GPU needed: hipcc cross-compiles): the per-merge kernels must keep the register / LDS budget their
launch geometry assumes.  The sparse launch sizes its grid for 16 resident waves per CU (4 per SIMD: <= 128 VGPRs) and the
streaming launch for 4 workgroups per CU; one VGPR too many halves the resident grid (measured: +27 % time), and the
allocator sits exactly at the limit (DESIGN.md "What comes next" 0)."""
from __future__ import annotations

import re
import shutil
import subprocess
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parent.parent / "yet-another-bpe_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def resources():
    if not Path(HIPCC).exists():
        pytest.skip("no hipcc")
    flags = re.search(r"^CXXFLAGS \?= (.*)$", (CSRC / "Makefile").read_text(), re.M).group(1).split()
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", *flags, "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", "yabpe.hip"],
                         cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return res


def test_sparse_launch_keeps_four_waves_per_simd(resources):
    # k_scan_skip<WEIGHTED, NW>: the production forms are NW = 8 and 16 (flat <0,*>, pooled <1,*>)
    seen = 0
    for name, r in resources.items():
        m = re.match(r"_ZN2yb11k_scan_skipILb([01])ELi(8|16)E", name)
        if not m:
            continue
        seen += 1
        assert r["VGPRs"] <= 128 and r["Occupancy"] >= 4, (name, r)
        nw = int(m.group(2))
        assert r["LDS"] * (16 // nw) <= 160 * 1024, (name, r)  # 16 waves per CU resident
        assert r["ScratchSize"] == 0, (name, r)
    assert seen >= 4


def test_streaming_launch_fits_four_workgroups_per_cu(resources):
    seen = 0
    for name, r in resources.items():
        if not name.startswith("_ZN2yb7k_applyILb"):
            continue
        seen += 1
        assert r["Occupancy"] >= 4 and r["ScratchSize"] == 0, (name, r)
        assert r["LDS"] * 4 <= 160 * 1024, (name, r)
    assert seen >= 3
