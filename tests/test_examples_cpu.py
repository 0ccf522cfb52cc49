"""include/pcv_hip.h through a C compiler and a C++ compiler (examples/): the header is plain C, the struct layouts
the Rust #[repr(C)] / ctypes mirrors assume are what gcc lays out, and the binaries link against the in-tree library."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")


def test_examples_compile_as_c11_and_cxx17():
    subprocess.check_call(["make", "-s", "-C", EX])
    for exe in ("build_octree", "query_octree", "ingest_batches"):
        p = subprocess.run([os.path.join(EX, "bin", exe)], capture_output=True, text=True)
        assert p.returncode == 2 and "usage" in p.stderr  # loads libpcv_hip.so, parses no arguments


def test_header_compiles_standalone_in_both_languages(tmp_path):
    src = tmp_path / "only_header.c"
    src.write_text('#include "pcv_hip.h"\nint (*probe)(void) = pcv_abi_version;\nint main(void) { return probe == 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-I", inc, "-c", str(src), "-o", str(tmp_path / "a.o")])
    subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-Wall", "-Werror", "-pedantic", "-I", inc, "-c", str(src), "-o",
                           str(tmp_path / "b.o")])


def test_ctypes_mirrors_have_the_asserted_layouts():
    from point_cloud_viewer_amd import _lib as L
    assert C.sizeof(L.Points) == 64 and L.Points.mem.offset == 56 and L.Points.color_stride.offset == 40
    assert C.sizeof(L.BuildParams) == 64 and L.BuildParams.flags.offset == 60
    assert C.sizeof(L.NodeInfo) == 80 and L.NodeInfo.cube_edge.offset == 56 and L.NodeInfo.point_offset.offset == 72
    assert C.sizeof(L.Shape) == 264 and C.sizeof(L.TopStreams) == 584 and C.sizeof(L.TopLayout) == 360
    assert C.sizeof(L.RoutedPoints) == 48 and C.sizeof(L.RouteState) == 32 and C.sizeof(L.Plane) == 16
    assert C.sizeof(L.SplitNode) == 56 and L.SplitNode.is_leaf.offset == 48 and C.sizeof(L.PromoteNode) == 24
