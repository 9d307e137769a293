"""C-ABI surface: libgpmi.so loads without a GPU, exports exactly what include/gpmi.h declares,
and refuses to run without a device (no CPU fallback)."""
import os
import re
import subprocess

import pytest

import gpmi355x
from gpmi355x import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "gpmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpmi_[a-z0-9_]+)\s*\(", src)))


def test_header_binding_and_library_agree():
    hdr = _header_symbols()
    assert hdr == sorted(_lib.SYMBOLS)
    lib = _lib.load()
    for s in hdr:
        assert hasattr(lib, s), f"libgpmi.so does not export {s}"
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r"\bT (gpmi_[a-z0-9_]+)$", nm, flags=re.M)))
    assert exported == hdr
    # ... and NOTHING else: no mangled C++ internals, no libstdc++ instantiations, no kernel stubs (-fvisibility=hidden + csrc/libgpmi.map)
    every = sorted(line.split()[-1] for line in nm.splitlines() if line.strip())
    assert every == hdr, [s for s in every if s not in hdr]


def test_version_string():
    assert b"gfx950" in _lib.load().gpmi_version()


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DeviceError):
        _lib.Context(0)
    with pytest.raises(_lib.DeviceError):
        gpmi355x.GP([[0.0, 1.0]], [0.0, 1.0], gpmi355x.MeanZero(), gpmi355x.SEIso(0.0, 0.0), -1.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gaussianprocesses.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in txt
