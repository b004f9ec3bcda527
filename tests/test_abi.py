"""The C-ABI library loads and exports every symbol include/pt_hip.h declares (no compute)."""
import ctypes as C
import os
import re

from conftest import ROOT
from path_tracing_amd import native


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pth?_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_header():
    names = _declared("pt_hip.h")
    assert set(names) == set(native.HIP_SYMBOLS), set(names) ^ set(native.HIP_SYMBOLS)
    lib = C.CDLL(native.HIP_LIB)
    for n in names:
        assert hasattr(lib, n), n
    lib.pt_build_info.restype = C.c_char_p
    assert b"gfx950" in lib.pt_build_info()


def test_host_library_exports_header():
    lib = C.CDLL(native.HOST_LIB)
    for n in _declared("pt_host.h"):
        assert hasattr(lib, n), n


def test_record_sizes_match_header():
    assert C.sizeof(native.FlatNode) == 64
    assert native.NODE_DTYPE.itemsize == 64 and native.ISECT_DTYPE.itemsize == 48 and native.SHADE_DTYPE.itemsize == 64
    assert C.sizeof(native.Camera) == 17 * 4 and C.sizeof(native.Light) == 24 and C.sizeof(native.TilePass) == 24


def test_cpp_surface_matches_the_reference_headers():
    """tests/native/surface_probe.cpp names every public member and function that the reference's
    integrator.h, canvas.h, kdtree.h, utils.h and visualizer.h (without the SDL display) offer to a front
    end and asserts, at compile time, that include/pt_render.h declares each with the reference's type."""
    import subprocess
    res = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "tests", "native", "surface_probe.cpp")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-3000:]


def test_c_abi_headers_are_plain_c():
    """include/pt_hip.h and include/pt_host.h are what a cgo / JNI / ctypes binding reads: they compile as
    strict C99 on their own, and a C program that calls through them links against the libraries."""
    import subprocess
    import tempfile
    for header in ("pt_hip.h", "pt_host.h"):
        res = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-x", "c", "-"],
                             input='#include "%s"\n' % header, capture_output=True, text=True, timeout=60)
        assert res.returncode == 0, res.stderr[-2000:]
    prog = '#include "pt_hip.h"\n#include "pt_host.h"\nint main(void) { int n = 0; (void)pt_device_count(&n); return pt_last_error() == 0; }\n'
    with tempfile.TemporaryDirectory() as tmp:
        res = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-x", "c", "-", "-o", os.path.join(tmp, "c_caller"),
                              "-L" + os.path.dirname(native.HIP_LIB), "-lpt_hip", "-lpt_host", "-Wl,-rpath-link,/opt/rocm/lib"],
                             input=prog, capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stderr[-2000:]
