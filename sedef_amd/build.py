"""Builds sedef_amd/lib/libsedef_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsedef_hip.so")
# code-generation flags of the product build beyond -O3 (SDF_HIPCC_FLAGS in the environment adds to them: experiments)
EXTRA_FLAGS = []


def _sources():
    out = [os.path.join(_HERE, "..", "include", "sedef_hip.h")]
    for f in sorted(os.listdir(SRC_DIR)):
        out.append(os.path.join(SRC_DIR, f))
    return out


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    m = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > m for s in _sources())


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function"] + EXTRA_FLAGS + os.environ.get("SDF_HIPCC_FLAGS", "").split() + \
          ["-o", LIB_PATH, os.path.join(SRC_DIR, "sdf_unity.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH
