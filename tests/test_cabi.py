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


def test_matmul_precision_switch():
    import serl_amd
    assert serl_amd.get_matmul_precision() in ("bf16_3x", "highest")
    serl_amd.set_matmul_precision("highest")
    assert serl_amd.get_matmul_precision() == "highest"
    serl_amd.set_matmul_precision("bf16_3x")
    assert serl_amd.get_matmul_precision() == "bf16_3x"
    with pytest.raises(ValueError, match="unknown precision"):
        serl_amd.set_matmul_precision("fp8")
    from serl_amd import _lib
    with pytest.raises(ValueError, match="mode must be"):
        _lib.call("serl_set_matmul_precision", 7)


def test_s_format_entry_points_validate_before_touching_the_device():
    from serl_amd import ops
    assert ops.convs_weight_bytes(True, 3, 3, 9, 32) == 4 * 96 * 32          # uint8 layer: K padded to KH*32
    assert ops.convs_weight_bytes(False, 3, 3, 32, 64) == 4 * 288 * 64
    with pytest.raises(ValueError, match="F % 8"):                           # F % 8
        ops.convs_fwd([16], [16], [16], [16], False, 1, 8, 8, 8, 3, 3, 2, 5, True, 0)
    with pytest.raises(ValueError, match="C % 8"):                           # C % 8 for S-format input
        ops.convs_fwd([16], [16], [16], [16], False, 1, 8, 8, 3, 3, 3, 2, 8, True, 0)
    with pytest.raises(ValueError, match="16-byte aligned"):
        ops.convs_fwd([16], [8], [16], [16], False, 1, 8, 8, 8, 3, 3, 2, 8, True, 0)
    with pytest.raises(ValueError, match="stride"):
        ops.convs_dgrad([16], [16], [16], [16], 1, 9, 9, 8, 3, 3, 3, 8, 0)
    with pytest.raises(ValueError, match="slabs need 2336 bytes, 4 available"):
        ops.convs_wgrad([16], [16], [16], [16], 16, 4, False, 1, 8, 8, 8, 3, 3, 2, 8, 0)
    with pytest.raises(ValueError, match="ngroups"):
        ops.sle_fwd([16] * 9, [16] * 9, [16] * 9, 8, 1, 4, 8, 8, 0)
