"""The C-ABI library loads and exports exactly what include/crt_amd.h declares (no compute)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, scene_path

import cudaraytracing_amd as crt
from cudaraytracing_amd import _capi as capi


def header_symbols():
    text = open(os.path.join(ROOT, "include", "crt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(capi.EXPORTED_SYMBOLS)


def test_library_exports_every_symbol():
    L = ctypes.CDLL(crt.LIB_PATH)
    for name in header_symbols():
        assert hasattr(L, name), name


def test_struct_sizes_match_the_c_side():
    # sizes the flat scene file format and the upload path rely on
    assert ctypes.sizeof(capi.Camera) == 108
    assert ctypes.sizeof(capi.Node) == 136
    assert ctypes.sizeof(capi.Material) == 36
    assert ctypes.sizeof(capi.Light) == 32
    assert ctypes.sizeof(capi.Stats) == 96 + 3 * 96 + 64 + 16 + 8


def test_environment_overrides_are_range_checked():
    """A typo in a CRT_* scheduler override must never reach the kernels: anything that is not a whole number inside
    the accepted range reads as the default."""
    k = capi.lib().crt_knob_value
    assert k(None, 12, 1, 64) == 12 and k(b"", 12, 1, 64) == 12
    assert k(b"16", 12, 1, 64) == 16 and k(b"1", 12, 1, 64) == 1 and k(b"64", 12, 1, 64) == 64
    for bad in (b"0", b"65", b"-3", b"abc", b"1e3", b" 8", b"8 ", b"99999999999999999999", b"4294967296"):
        assert k(bad, 12, 1, 64) == 12, bad
    assert k(b"0", 5, 0, 10) == 0


def test_error_paths_without_compute(tmp_path):
    L = capi.lib()
    h = ctypes.c_void_p()
    assert L.crt_scene_load_xml(b"/nonexistent/scene.xml", None, ctypes.byref(h)) == capi.E_IO
    assert b"nonexistent" in L.crt_host_last_error()
    assert L.crt_scene_load_file(b"/nonexistent/x.crtscene", ctypes.byref(h)) == capi.E_IO
    bad = tmp_path / "bad.crtscene"
    bad.write_bytes(b"not a scene")
    assert L.crt_scene_load_file(str(bad).encode(), ctypes.byref(h)) == capi.E_IO
    assert L.crt_scene_load_xml(None, None, ctypes.byref(h)) == capi.E_INVALID
    assert L.crt_create(0, None) == capi.E_INVALID
    assert L.crt_upload_scene(None, None) == capi.E_INVALID
    assert L.crt_render_pass(None, 0, 1, None, 0) == capi.E_INVALID


def test_no_device_is_reported_not_hidden():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(crt.CrtError) as e:
        crt.Context(0)
    assert e.value.code == capi.E_HIP


def test_product_does_not_reference_the_oracle():
    # the renderer must never load, link, include or import anything of the CPU oracle
    needles = ("liboracle", "oracle_lib", "oracle/", "oracle.cpp", "oracle_detmath", "import oracle", "from oracle")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cudaraytracing_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for n in needles:
                    assert n not in text, (os.path.join(dirpath, f), n)
