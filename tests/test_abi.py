"""CPU tests of the drop-in boundary: the shared library loads and exports every symbol include/pup_hip.h
declares, the Python binding covers exactly that set, and argument errors are reported without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pup_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pup_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(hip_lib):
    from coolpuppy_amd import _ffi
    declared = _declared()
    assert len(declared) >= 20
    assert declared == _ffi.declared_symbols(), "ctypes binding and header disagree"
    raw = C.CDLL(_ffi.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} is declared in include/pup_hip.h but not exported by libpup_hip.so"
    # ... and nothing else: the object is built with hidden visibility, its dynamic symbol table holds the C ABI only
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True)
    if out.returncode == 0:
        exported = sorted(line.split()[-1] for line in out.stdout.splitlines() if " T " in line)
        assert exported == declared, f"unexpected exports: {sorted(set(exported) - set(declared))}"


def test_version_and_null_handling(hip_lib):
    assert hip_lib.pup_version() >= 100
    assert isinstance(hip_lib.pup_last_error(None), bytes)
    # NULL context: every entry point must return PUP_EINVAL, never crash
    assert hip_lib.pup_sync(None) == -1
    assert hip_lib.pup_reset(None, 1, 1) == -1
    assert hip_lib.pup_create(0, None) == -1


def test_no_cpu_fallback_in_product():
    """The product package never imports the oracle (a fallback would void the parity claims)."""
    pkg = os.path.join(ROOT, "coolpuppy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} references the oracle"
                assert "liboracle" not in src, f


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from coolpuppy_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="not found"):
        _ffi.lib()


def test_every_source_under_csrc_makes_the_library_stale(monkeypatch):
    """coolpuppy_amd.build.is_stale(): a newer file ANYWHERE under csrc/ (or the public header) must trigger a rebuild — round 4's
    hand-kept header list missed pup_bin.hpp, so an edit of the binning alone left tests running yesterday's library."""
    import os
    from coolpuppy_amd import build
    csrc = os.path.join(ROOT, "coolpuppy_amd", "csrc")
    sources = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".hpp", ".h")))
    assert len(sources) >= 9 and any(f.endswith("pup_bin.hpp") for f in sources)
    for f in sources + [os.path.join(ROOT, "include", "pup_hip.h")]:
        assert f in build.DEPS, f
    real = os.path.getmtime
    lib_time = real(build.OUT) if os.path.exists(build.OUT) else 0.0
    for touched in sources:
        monkeypatch.setattr(os.path, "getmtime", lambda p, t=touched: lib_time + 10 if p == t else (lib_time if p == build.OUT else min(real(p), lib_time)))
        assert build.is_stale(), touched
    monkeypatch.setattr(os.path, "getmtime", lambda p: lib_time if p == build.OUT else min(real(p), lib_time))
    assert not build.is_stale() or not os.path.exists(build.OUT)
