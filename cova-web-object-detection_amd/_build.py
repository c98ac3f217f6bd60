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


# per-file extra flags.  conv_wino: no SLP vectorisation -- on gfx950 packed f32 VALU ops cannot issue
# in the shadow of an MFMA (LLVM unpacks them again and leaves the shuffle moves behind).
FILE_FLAGS = {"conv_wino.hip": ["-fno-slp-vectorize"],
              # conv_wgrad4: the same (packed transform arithmetic cost 210 register moves per K-step of the input role)
              "conv_wgrad4.hip": ["-fno-slp-vectorize"]}


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    objdir = os.path.join(PKG_DIR, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
              "-Wall", "-Wno-unused-function"]
    common += os.environ.get("COVA_EXTRA_FLAGS", "").split()
    if os.environ.get("COVA_ABLATE"):          # tools/conv_bench.py ablation study builds
        common.append("-DCOVA_ABLATE=1")
    procs, objs = [], []
    for src in SOURCES:                        # one hipcc per translation unit, in parallel
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = common + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
