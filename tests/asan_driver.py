"""Run by tests/test_capi_asan.py inside an interpreter that has the AddressSanitizer runtime preloaded: every exported
entry point of an ASan-instrumented host build of libndp_hip.so is called with null pointers / zero sizes / small bad
values.  Each must come back with an error code (or a size) -- no argument check may touch memory before it has looked at
its arguments.  No GPU is needed: a call that gets past its checks fails at the first HIP call, also with a code."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NDP_LIB_PATH"] = sys.argv[1]
from ndivplanning_amd import _capi  # noqa: E402

lib = _capi.load()
called = 0
# entry points for which all-null / all-zero arguments are a valid request (switches, queries, "nothing to close")
LENIENT = ("ndp_fm_side_stream", "ndp_timing_enable", "ndp_timing_collect", "ndp_fm_set_stat_sync", "ndp_fm_grad_buckets",
           "ndp_p2p_close", "ndp_p2p_region_free")
for name, (res, args) in sorted(_capi.SIGNATURES.items()):
    if name in ("ndp_last_error", "ndp_version"):
        continue
    for fill in (0, 1, -1, 7):
        vals = []
        for a in args:
            if a in (ctypes.c_void_p, ctypes.c_char_p) or issubclass(a, ctypes._Pointer):
                vals.append(None)
            elif a in (ctypes.c_float, ctypes.c_double):
                vals.append(float(fill))
            elif issubclass(a, ctypes._SimpleCData):
                vals.append(fill if "u" not in a._type_.lower() or fill >= 0 else 0)
            else:
                vals.append(a())                      # a structure passed by value: all zero
        rc = getattr(lib, name)(*vals)
        called += 1
        if res is ctypes.c_int and name not in LENIENT:
            assert rc != 0, "%s(%r) accepted null pointers" % (name, vals)
            assert lib.ndp_last_error(), name
# a few calls with real (host) buffers where the function is host-only
off = (ctypes.c_int64 * 16)()
cnt = (ctypes.c_int64 * 16)()
assert lib.ndp_fm_grad_buckets(off, cnt, 16) == 7 and sum(cnt[:7]) == lib.ndp_fm_param_floats()
assert lib.ndp_fm_grad_buckets(off, cnt, 6) != 7
o, dims = ctypes.c_int64(), (ctypes.c_int64 * 6)()
for what in range(6):
    for idx in range(-1, 16):
        lib.ndp_fm_layout(what, idx, ctypes.byref(o), dims)
print("asan driver ok: %d calls over %d entry points" % (called, len(_capi.SIGNATURES) - 2))
