"""Builds lib/libcova_hip.so (gfx950) from csrc/*.hip with hipcc -- in-tree, no JIT cache."""
import glob
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "lib", "libcova_hip.so")
SOURCES = sorted(glob.glob(os.path.join(PKG_DIR, "csrc", "*.hip")))
HEADERS = sorted(glob.glob(os.path.join(PKG_DIR, "csrc", "*.h")))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
    if os.environ.get("COVA_ABLATE"):          # tools/conv_bench.py ablation study builds
        cmd.append("-DCOVA_ABLATE=1")
    cmd += SOURCES + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
