"""The C-ABI library loads without a GPU and exports every symbol include/vidsitu_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

from vidsitu_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    hdr = open(os.path.join(ROOT, "include", "vidsitu_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", hdr))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert names == set(_lib.SIGNATURES), "python binding table out of sync with the header"


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.vs_version() >= 1
    assert isinstance(lib.vs_last_error_string(), bytes)


def test_conv_desc_layout_matches_header():
    assert ctypes.sizeof(_lib.ConvDesc) == 22 * 4
    hdr = open(os.path.join(ROOT, "include", "vidsitu_hip.h")).read()
    body = re.search(r"typedef struct vs_conv_desc \{(.*?)\} vs_conv_desc;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"int32_t ([^;]+);", body):
        fields += [f.strip() for f in decl.split(",")]
    assert fields == [f[0] for f in _lib.ConvDesc._fields_]


def test_argument_validation_without_gpu():
    """Bad descriptors are rejected on the host before any launch."""
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.N, d.Cin, d.Cout = 1, 3, 8  # Cin not a multiple of 8
    rc = lib.vs_conv_fwd(None, None, None, ctypes.byref(d), None, None, None, None, None, 0, None)
    assert rc == -1
    assert b"multiples of 8" in lib.vs_last_error_string()
