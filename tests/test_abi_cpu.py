"""CPU-side checks of the product library: it loads, exports every symbol include/pcv_hip.h declares, and its
host logic (level table, argument validation that needs no device) agrees with the oracle. No compute calls."""
import os
import re

import numpy as np

import oracle_lib as O
import point_cloud_viewer_amd as pcv
from point_cloud_viewer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_declared_abi():
    lib = pcv.load_library()
    header = open(os.path.join(ROOT, "include", "pcv_hip.h")).read()
    declared = set(re.findall(r"\b(pcv_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libpcv_hip.so does not export {name}"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert lib.pcv_abi_version() == 2


def test_level_table_matches_oracle():
    cases = [((0, 0, 0), (283.0, 120.0, 20.0), 0.001), ((-5, -5, -5), (5, 5, 5), 1.0),
             ((1e6, 2e6, 3e6), (1e6 + 1000.5, 2e6 + 3, 3e6 + 999), 0.001), ((0, 0, 0), (20000.0, 1, 1), 0.001),
             ((0, 0, 0), (1, 1, 1), 2.0), ((0, 0, 0), (0.3, 0.2, 0.1), 1e-7)]
    for bmin, bmax, res in cases:
        ml_o, edge_o, enc_o = O.level_table(bmin, bmax, res, cap=40)
        ml, edge, enc = pcv.level_table(bmin, bmax, res, cap=40)
        assert ml == ml_o
        assert np.array_equal(edge, edge_o)
        assert np.array_equal(enc, enc_o)


def test_node_name_matches_reference_display():
    # src/octree/node.rs:73-86
    assert pcv.node_name(0, 0) == "r"
    assert pcv.node_name(2 << 56, 0o13) == "r13"
    assert pcv.node_name(3 << 56, 0o007) == "r007"
    hi, lo = O.node_id_from_str("r" + "5" * 21)
    assert pcv.node_name(hi, lo) == "r" + "5" * 21


def test_no_oracle_on_the_product_path():
    """The product package must not import, link or call anything under oracle/ (tier rule ③)."""
    pkg = os.path.join(ROOT, "point_cloud_viewer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "pcv_oracle" not in text and "libpcv_oracle" not in text, f
