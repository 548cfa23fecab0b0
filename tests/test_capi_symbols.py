"""CPU-only: the C-ABI library builds, loads, and exports every function that
include/ndp.h declares (no compute calls here -- those need a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ndp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(ndp_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    from ndivplanning_amd import _build, _capi
    _build.build()
    return _capi.load()


def test_header_and_binding_agree(lib):
    from ndivplanning_amd import _capi
    declared = _declared_functions()
    assert declared, "no functions parsed from ndp.h"
    assert sorted(_capi.SIGNATURES) == declared


def test_every_declared_symbol_is_exported(lib):
    raw = ctypes.CDLL(lib._name)
    for name in _declared_functions():
        assert hasattr(raw, name), name


def test_host_only_entry_points(lib):
    from ndivplanning_amd import _capi
    assert lib.ndp_version() == _capi.EXPECTED_VERSION == 135
    assert lib.ndp_g_param_count(2) == 83780
    assert lib.ndp_d_param_count() == 58305           # row a4
    assert lib.ndp_g_param_count(16) == 83780 + 128 * 14
    assert lib.ndp_pad_rows(2688) == 2688 and lib.ndp_pad_rows(42) == 64
    assert lib.ndp_g_acts_floats(2688) == 2688 * 576
    assert lib.ndp_ndiv_partials(448, 6) >= 448 // 42


def test_argument_errors_are_reported_without_launching(lib):
    from ndivplanning_amd import _capi
    rc = lib.ndp_ndiv_fwd_bwd(None, 4, None, 2, 4, 6, 1.0, None, None, None, None)
    assert rc == 1
    assert b"null" in lib.ndp_last_error()
    cfg = _capi.StepConfig(noise_dim=2, num_sample=0, flat=10)
    assert lib.ndp_step_workspace_floats(ctypes.byref(cfg)) == 0
    cfg = _capi.StepConfig(noise_dim=2, num_sample=6, flat=448)
    assert lib.ndp_step_workspace_floats(ctypes.byref(cfg)) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ndivplanning_amd import _build, _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_build, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_capi.NdpError):
        _capi.load()
