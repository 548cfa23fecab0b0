"""Build libndp_hip.so (the C-ABI HIP library, include/ndp.h) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
.so is git-ignored but travels to the GPU box with the working tree.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
# NDP_LIB_PATH: load another build of the library (diagnostic builds with -DNDP_STAMPS / -DNDP_EXP_*; scripts/probe)
LIB_PATH = os.environ.get("NDP_LIB_PATH") or os.path.join(LIB_DIR, "libndp_hip.so")
SOURCES = ["ndp_kernels.hip"]
DEPS = ["ndp_kernels.hip", "ndp_device.h", "ndp_capi.inc", "ndp_encoder.inc", "ndp_forward_model.inc", os.path.join("..", "..", "include", "ndp.h")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")
    return exe


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > built for d in DEPS)


def build(force=False, verbose=False, extra_flags=(), out_path=None):
    """Compile the HIP sources into ndivplanning_amd/lib/libndp_hip.so (or a diagnostic variant: extra_flags,
    out_path)."""
    target = out_path or LIB_PATH
    if out_path is None and not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", "-Wno-pass-failed"] + list(extra_flags)
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", target + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
