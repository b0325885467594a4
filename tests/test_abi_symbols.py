"""CPU-side checks of the drop-in boundary: the C-ABI libraries load and export every symbol the headers declare
(no compute calls: there is no GPU in the build container), and creating a context without a GPU fails cleanly."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_headers_and_binding_lists_agree():
    import idahip
    assert declared("ida_hip.h", "idahip_") == sorted(idahip.HIP_SYMBOLS)
    assert declared("ida_ensemble.h", "idaens_") == sorted(idahip.ENS_SYMBOLS)


def test_libraries_export_every_declared_symbol():
    import idahip
    H = C.CDLL(idahip.LIB_HIP, mode=C.RTLD_GLOBAL)
    E = C.CDLL(idahip.LIB_ENS)
    for s in declared("ida_hip.h", "idahip_"):
        assert hasattr(H, s), s
    for s in declared("ida_ensemble.h", "idaens_"):
        assert hasattr(E, s), s


def test_product_path_does_not_reference_the_oracle():
    """The shipped package must never import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "rust-ida_amd")
    for dp_, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp_, f)).read()
                assert "oracle_lib" not in txt and "libida_oracle" not in txt and "oracle/" not in txt.replace("oracle/problems.hpp", ""), f


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import idahip
    with pytest.raises(idahip.IdaHipError):
        idahip.Ctx("roberts", 3, 1)
