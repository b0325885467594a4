"""f-4 of SURVEY.md 8(f): the Rust crates a rust-ida maintainer adds (bindings/rust/). No Rust toolchain exists in the build
image, so they cannot be compiled here; what can be checked is that they cannot drift from the C ABI:
  * ida-hip-sys/src/lib.rs is exactly what tools/gen_rust_sys.py generates from the two headers (every prototype, every
    enumerator), and every `extern` item names a symbol the shared libraries export;
  * every `sys::` item the safe crate uses exists in the generated bindings;
  * both sources are lexically well formed as far as a bracket / string scanner can tell."""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS = os.path.join(ROOT, "bindings", "rust", "ida-hip-sys", "src", "lib.rs")
SAFE = os.path.join(ROOT, "bindings", "rust", "ida-hip", "src", "lib.rs")
SAFE_ALL = [os.path.join(ROOT, "bindings", "rust", "ida-hip", "src", f) for f in ("lib.rs", "nls.rs", "problem.rs")]


def test_sys_crate_is_the_generated_one():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"]).returncode == 0


def test_every_extern_item_is_an_exported_symbol_with_the_headers_arity():
    import idahip
    H = C.CDLL(idahip.LIB_HIP, mode=C.RTLD_GLOBAL)
    E = C.CDLL(idahip.LIB_ENS)
    txt = open(SYS).read()
    fns = re.findall(r"pub fn (\w+)\((.*?)\) -> ", txt)
    assert sorted(f for f, _ in fns) == sorted(idahip.HIP_SYMBOLS + idahip.ENS_SYMBOLS)
    for name, args in fns:
        assert hasattr(H if name.startswith("idahip_") else E, name), name
    # arity against the C prototypes
    for header in ("ida_hip.h", "ida_ensemble.h"):
        c = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        for name, cargs in re.findall(r"\b(ida(?:hip|ens)_\w+)\s*\(([^;{]*?)\)\s*;", c):
            rust = dict(fns)[name]
            assert len([a for a in cargs.split(",") if a.strip() and a.strip() != "void"]) == len([a for a in rust.split(",") if a.strip()]), name


def test_safe_crate_only_uses_items_of_the_sys_crate():
    sys_txt = open(SYS).read()
    items = set(re.findall(r"pub (?:fn|const|type|struct) (\w+)", sys_txt))
    used = set()
    for path in SAFE_ALL:
        used |= set(re.findall(r"\bsys::(\w+)", open(path).read()))
    assert used and used <= items, sorted(used - items)
    # the trait surface north_star names is there: LSolver, NLSolver, NLProblem, and the IdaProblem bridge
    txt = "".join(open(path).read() for path in SAFE_ALL)
    for needle in ("impl<D> LSolver<f64, D> for HipDense<D>", "impl<D> NLSolver<f64, D> for HipNewton<D>",
                   "impl<D> NLProblem<f64, D> for HipNlsProblem<D>", "impl<P> HostProblem for IdaProblemAdapter<P>"):
        assert needle in txt, needle


def _balanced(path):
    txt = open(path).read()
    txt = re.sub(r"//[^\n]*", "", txt)                      # line comments (incl. doc comments)
    txt = re.sub(r'"(?:\\.|[^"\\])*"', '""', txt)          # string literals
    txt = re.sub(r"'(?:\\.|[^'\\])'", "''", txt)           # char literals (lifetimes like '_ stay, harmless)
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    for ch in txt:
        if ch in "([{":
            stack.append(ch)
        elif ch in ")]}":
            assert stack and stack.pop() == pairs[ch], path
    assert not stack, path


def test_sources_are_lexically_well_formed():
    _balanced(SYS)
    for path in SAFE_ALL:
        _balanced(path)
    for p in ("Cargo.toml", os.path.join("ida-hip-sys", "Cargo.toml"), os.path.join("ida-hip-sys", "build.rs"),
              os.path.join("ida-hip", "Cargo.toml")):
        assert os.path.getsize(os.path.join(ROOT, "bindings", "rust", p)) > 0
