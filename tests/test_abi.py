"""CPU tier: libmrzgpu.so loads and exports every symbol include/*.h declares
(no compute calls -- there is no GPU here), and refuses to run without a device."""
import ctypes
import os
import re

import pytest

import modern_rzip_amd as m

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("mrzgpu.h", "mrzgpu_host.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for mm in re.finditer(r"\b(mrz_[a-z0-9_]+)\s*\(", src):
            names.add(mm.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    path = m.lib_path()
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.mrz_abi_version() == 4


def test_no_cpu_fallback_without_device():
    """On a machine without a HIP device mrz_open must fail with MRZ_E_NODEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = m.load_library()
    with pytest.raises(m.MrzError):
        m.RzipContext(lib=lib)


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(m.MrzError):
        m.load_library(str(tmp_path / "nope.so"))


def test_host_control_layout_matches_header():
    # mrz_control in include/mrzgpu_host.h <-> binding.Control
    assert [f[0] for f in m.Control._fields_] == [
        "rzip_compression_level", "compression_level", "window", "unlimited", "ramsize", "page_size", "hash_code",
        "device", "lz4_test", "threshold"]
    assert ctypes.sizeof(m.ChunkResult) == 8 + 8 + 4 + 4 + 8 + 8 + 56 + 24


def test_integration_snippet_compiles_and_matches_the_document():
    """INTEGRATION.md section 1 is real code: the copy in tests/c/integration_snippet.c passes gcc -fsyntax-only against
    include/mrzgpu.h and the reference prototypes it names, and the document holds the same text."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "c", "integration_snippet.c")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.join(root, "tests", "c"), src], check=True)
    text = open(src).read()
    body = text[text.index("*/\n", text.index("BEGIN SNIPPET")) + 3:text.index("/* END SNIPPET */")]
    assert body in open(os.path.join(root, "INTEGRATION.md")).read()
    # the C caller test is valid C99 against the public headers as well (it runs in the GPU tier)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.join(root, "oracle"), os.path.join(root, "tests", "c", "capi_test.c")], check=True)
