"""The C-ABI library loads on a CPU-only machine and exports every symbol include/serl_hip.h declares
(no compute calls here); argument validation happens before any device access."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "serl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(serl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from serl_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/serl_hip.h but not exported"
    assert set(_lib.EXPORTS) == set(names), set(_lib.EXPORTS) ^ set(names)
    assert lib.serl_version() == 1


def test_argument_errors_map_to_python_exceptions():
    from serl_amd import ops
    with pytest.raises(ValueError, match="null pointer"):
        ops.gemm(None, None, None, None, 1, 1, 1, 1, a_ms=1, a_ks=1, b_ks=1, b_ns=1, c_ms=1)
    with pytest.raises(ValueError, match="ngroups"):
        ops.conv2d_fwd([1] * 9, [1] * 9, [1] * 9, [1] * 9, False, 1, 8, 8, 3, 3, 3, 1, 4, True, 0)
    with pytest.raises(ValueError, match="bad geometry"):
        ops.conv2d_fwd([1], [1], [1], [1], False, 1, 2, 2, 3, 3, 3, 1, 4, True, 0)
    with pytest.raises(ValueError):
        ops.crop_offsets((0, 1), 5, 3, 4, 4, 1, 0)          # shard outside the global batch
    assert ops.conv2d_wgrad_workspace_bytes(2, 256, 128, 128, 9, 3, 3, 2, 32) > 0


def test_missing_library_fails_loudly(monkeypatch):
    from serl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libserl_hip.so")
    with pytest.raises(_lib.SerlHipError, match="no CPU fallback"):
        _lib.load()
