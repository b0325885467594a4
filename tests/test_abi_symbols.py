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


SWITCHES = ("IDAHIP_EXP_NOPRO", "IDAHIP_EXP_NOUPD", "IDAHIP_EXP_NOGATHER", "IDAHIP_EXP_NODIAG", "IDAHIP_EXP_NOSWEEP", "IDAHIP_STAMPS",
            "IDAHIP_TRAIL_PIPE", "IDAHIP_TRAIL_QUAD", "IDAHIP_WP_RING", "IDAHIP_US_PAD", "IDAHIP_SYS_UNR")


def test_shipped_library_is_not_a_timing_build():
    """The timing-build switches (kernels with a part removed: garbage results) live in csrc/exp_switches.hpp alone, are refused
    by the compiler without -DIDAHIP_TIMING_BUILD, the default `make` defines none of them, and the built library says so."""
    import idahip
    csrc = os.path.join(ROOT, "rust-ida_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "IDAHIP_TIMING_BUILD" not in mk and not any(s in mk for s in SWITCHES), "the default build must define no timing switch"
    guard = open(os.path.join(csrc, "exp_switches.hpp")).read()
    assert "#error" in guard and all("defined(%s)" % s in guard for s in SWITCHES)
    for f in sorted(os.listdir(csrc)):  # no kernel source tests a switch macro itself: they read the constants of namespace tb
        if f.endswith((".hpp", ".hip")) and f != "exp_switches.hpp":
            for line in open(os.path.join(csrc, f)):
                if line.lstrip().startswith("#") and any(s in line for s in SWITCHES):
                    # (idahip_debug_stamps, an extra export of stamp builds, is the one #ifdef left: it adds a symbol, changes no kernel)
                    assert f == "idahip.hip" and "IDAHIP_STAMPS" in line, (f, line)
            assert "__CUDACC__" not in open(os.path.join(csrc, f)).read(), f
    H = C.CDLL(idahip.LIB_HIP, mode=C.RTLD_GLOBAL)
    assert H.idahip_timing_build() == 0


def test_a_switch_without_the_timing_build_flag_does_not_compile(tmp_path):
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "t.cpp"
    src.write_text('#include "exp_switches.hpp"\nint main() { return idahip::tb::NOPRO ? 1 : 0; }\n')
    inc = os.path.join(ROOT, "rust-ida_amd", "csrc")
    bad = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", inc, "-DIDAHIP_EXP_NOPRO", str(src)], capture_output=True, text=True)
    assert bad.returncode != 0 and "timing-build" in bad.stderr
    ok = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", inc, "-DIDAHIP_EXP_NOPRO", "-DIDAHIP_TIMING_BUILD", str(src)], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    assert subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", inc, str(src)], capture_output=True).returncode == 0
