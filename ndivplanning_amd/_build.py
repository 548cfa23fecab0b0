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
LIB_PATH = os.path.join(LIB_DIR, "libndp_hip.so")
SOURCES = ["ndp_kernels.hip"]
DEPS = ["ndp_kernels.hip", "ndp_device.h", "ndp_capi.inc", "ndp_encoder.inc", os.path.join("..", "..", "include", "ndp.h")]


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


def build(force=False, verbose=False):
    """Compile the HIP sources into ndivplanning_amd/lib/libndp_hip.so."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", "-Wno-pass-failed"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
