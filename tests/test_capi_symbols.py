"""The C-ABI library loads without a GPU and exports every function include/mpnhip.h declares;
the ctypes table in mpntrackseg_amd/capi.py covers the same set.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

from mpntrackseg_amd import capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(REPO, "include", "mpnhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpnhip_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.lib_path()):
        subprocess.check_call(["make", "-C", REPO, "-j4"], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(capi.lib_path())


def test_header_functions_are_exported(lib):
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mpnhip.h but not exported"


def test_ctypes_table_matches_header():
    assert sorted(capi.SIGNATURES) == declared_functions()


def test_version_and_sizes_without_gpu(lib):
    l = capi.load()
    assert l.mpnhip_version().decode().startswith("mpnhip")
    # pure host arithmetic: buffer sizes grow with the graph
    a, b = l.mpnhip_graph_bytes(10, 100), l.mpnhip_graph_bytes(1000, 100000)
    assert 0 < a < b


def test_struct_layout_matches_c():
    # sizeof(mpnhip_mlp) = 2 ints + 8 ints + 4 * 8 pointers, padded to 8
    assert ctypes.sizeof(capi.Mlp) == 8 + 32 + 4 * 8 * 8
    # 6 ints, 7 MLPs, the trailing precision int padded to the struct's 8-byte alignment
    assert ctypes.sizeof(capi.Model) == 24 + 7 * ctypes.sizeof(capi.Mlp) + 8
    assert capi.Model.precision.offset == 24 + 7 * ctypes.sizeof(capi.Mlp)
