"""CPU: the C-ABI library loads and exports every symbol include/tigar_hip.h declares; the
product path fails loudly without a GPU (no CPU fallback)."""
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tigar_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from tigar_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load(require_device=False)
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
        assert n in _lib.PROTOTYPES, "no ctypes prototype for %s" % n
    for n in _lib.PROTOTYPES:
        assert n in names, "%s bound in Python but not declared in the header" % n


def test_no_cpu_fallback_without_gpu():
    import shutil
    from tigar_amd import _lib
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.TigarHipError):
        _lib.load(require_device=True)
    from tigar_amd import device
    with pytest.raises(_lib.TigarHipError):
        device.DeviceVector(4)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "tigar_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("tg_oracle", ""), "%s mentions the oracle" % f
