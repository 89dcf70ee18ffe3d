"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/cosyvoice_amd.h declares."""
import ctypes
import os
import re


def test_library_exports_header_symbols():
    from cosyvoice_amd import build, _lib
    path = build.build(verbose=False)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "cosyvoice_amd.h")).read()
    declared = set(re.findall(r"\b(cv_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    lib.cv_arch.restype = ctypes.c_char_p
    assert lib.cv_arch() == b"gfx950"
    _lib.lib()  # struct-size handshake
