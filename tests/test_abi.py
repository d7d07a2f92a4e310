"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/unimm_hip.h declares (no compute call is made without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from unimm_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "unimm_hip.h")).read()
    declared = set(re.findall(r"\b(unimm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 4
    so = ctypes.CDLL(built_lib)
    for name in sorted(declared):
        assert hasattr(so, name), f"{name} declared in include/unimm_hip.h but not exported"
    from unimm_amd import lib
    assert set(lib.SYMBOLS) == declared


def test_version_and_arch(built_lib):
    from unimm_amd import lib
    L = lib.lib()
    assert L.unimm_version() == lib.ABI_VERSION
    assert L.unimm_arch() == b"gfx950"
