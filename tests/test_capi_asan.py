"""Host side of the C ABI under AddressSanitizer ("optional -fsanitize=address host build of the C-ABI
shim").  The library is compiled once more with the sanitizer on the HOST half only (`-Xarch_host -fsanitize=address`: the
device half is what ships) and every exported entry point is called, in a child interpreter with the sanitizer's runtime
preloaded, with null pointers, zero / negative sizes and out-of-range indices (tests/asan_driver.py): each must answer with
an error code or a size, and the sanitizer must stay silent.  No GPU involved.  (~45 s: one extra compilation.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    from ndivplanning_amd import _build
    out = subprocess.run([_build._hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = out.stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_every_entry_point_rejects_null_arguments_under_the_address_sanitizer(tmp_path):
    from ndivplanning_amd import _build
    runtime = _asan_runtime()
    if runtime is None:
        pytest.skip("this toolchain has no shared AddressSanitizer runtime")
    lib = str(tmp_path / "libndp_hip_asan.so")
    _build.build(extra_flags=("-O1", "-g", "-Xarch_host", "-fsanitize=address"), out_path=lib)
    env = dict(os.environ, LD_PRELOAD=runtime, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86")
    env.pop("NDP_LIB_PATH", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py"), lib], env=env, capture_output=True,
                       text=True, timeout=600)
    assert "AddressSanitizer" not in p.stderr and "AddressSanitizer" not in p.stdout, p.stderr[-3000:]
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-1500:])
    assert "asan driver ok" in p.stdout
