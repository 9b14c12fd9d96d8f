"""CPU: the C-ABI shared library loads and exports every symbol include/rrt_hip.h declares
(no compute calls -- there is no GPU here).  Also: the product never imports the oracle."""
import ctypes
import os
import re

import pytest

from rrtplanner_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "rrt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rrt_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_ffi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    _ffi.lib()  # argtypes bind


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _ffi.lib()


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "rrtplanner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, os.path.join(dirpath, f)


def test_integration_stub_structs_match_the_binding():
    """The ctypes structs in INTEGRATION.md's stub have the layout of the binding's (a field appended to rrt_query / rrt_result
    without the stub following would make rrt_plan write past the stub's struct).  The stub itself runs in the GPU suite."""
    import ctypes as C
    import re

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    ns = {}
    exec(compile(code.replace('C.CDLL("librrt_hip.so")', f"C.CDLL({_ffi.LIB_PATH!r})"), "INTEGRATION.md", "exec"), ns)
    for stub, ours in ((ns["_Query"], _ffi.Query), (ns["_Result"], _ffi.Result)):
        assert C.sizeof(stub) == C.sizeof(ours)
        assert [(n, getattr(stub, n).offset) for n, _ in stub._fields_] == [(n, getattr(ours, n).offset) for n, _ in ours._fields_]
