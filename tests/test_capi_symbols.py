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
    # 6 ints, 7 MLPs, then precision and weights_prepacked (two ints)
    assert ctypes.sizeof(capi.Model) == 24 + 7 * ctypes.sizeof(capi.Mlp) + 8
    assert capi.Model.precision.offset == 24 + 7 * ctypes.sizeof(capi.Mlp)
    assert capi.Model.weights_prepacked.offset == capi.Model.precision.offset + 4


def test_linear_bf16_args_layout_matches_c():
    # 15 pointers / int64 fields, int64 m, 7 ints, padded to 8
    assert ctypes.sizeof(capi.LinearBf16Args) == 15 * 8 + 8 + 7 * 4 + 4
    assert capi.LinearBf16Args.m.offset == 120 and capi.LinearBf16Args.n.offset == 128 and capi.LinearBf16Args.accumulate.offset == 152


def test_precision_codes_match_header():
    src = open(os.path.join(REPO, "include", "mpnhip.h")).read()
    codes = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"#define MPNHIP_PREC_([A-Z0-9_]+) (\d+)", src)}
    assert codes == capi.PRECISIONS


def test_new_entry_points_argument_checks_without_gpu():
    """Size queries and the argument checks of the f-3 / f-4 / optimizer entry points run on the host: no device work is
    reached for empty inputs, null pointers or undersized workspaces (error codes as include/mpnhip.h states)."""
    l = capi.load()
    assert l.mpnhip_knn_mask_workspace_bytes(1000, 0) > l.mpnhip_knn_mask_workspace_bytes(1000, 1) > 0
    assert l.mpnhip_time_valid_conn_workspace_bytes(500) >= 501 * 8
    assert l.mpnhip_compact_workspace_bytes(1000) > 0
    # empty inputs are successful no-ops
    assert l.mpnhip_knn_mask(None, None, 0, 0, 5, 1, 1, None, None, 0, None) == 0
    assert l.mpnhip_edge_features(None, 0, 0, None, ctypes.c_float(25.0), None, None, None, None, None, None) == 0
    assert l.mpnhip_pairwise_distance(None, 0, 0, None, 0, ctypes.c_float(1e-6), None, None) == 0
    assert l.mpnhip_gather_rows(None, 4, None, 0, 4, None, None) == 0
    assert l.mpnhip_average_preds(None, None, 0, None, None) == 0
    assert l.mpnhip_adam_step(None, None, None, None, 0, ctypes.c_float(1e-3), ctypes.c_float(0.9), ctypes.c_float(0.999),
                              ctypes.c_float(1e-8), ctypes.c_float(0.0), 1, None) == 0
    # null pointers / bad sizes are refused before anything is launched
    assert l.mpnhip_knn_mask(None, None, 10, 5, 5, 1, 1, None, None, 0, None) != 0
    assert b"knn_mask" in l.mpnhip_last_error()
    assert l.mpnhip_edge_features(None, 3, 2, None, ctypes.c_float(25.0), None, None, None, None, None, None) != 0
    assert l.mpnhip_adam_step(None, None, None, None, 5, ctypes.c_float(1e-3), ctypes.c_float(0.9), ctypes.c_float(0.999),
                              ctypes.c_float(1e-8), ctypes.c_float(0.0), 0, None) != 0
    assert l.mpnhip_window_accumulate(None, None, 3, None, 2, 0, None, None, None) != 0   # more kept than window edges
    # embedding-file selection (f-4): empty inputs succeed, null pointers are refused
    assert l.mpnhip_embedding_keep(None, 17, 0, None, 0, None, None) == 0
    assert l.mpnhip_embedding_keep(None, 17, 5, None, 0, None, None) != 0
    assert b"embedding_keep" in l.mpnhip_last_error()
    assert l.mpnhip_embedding_check(None, 17, None, 0, None, None, None) != 0   # the mismatch counter is always required
