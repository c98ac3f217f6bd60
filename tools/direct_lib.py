"""Builds and loads tools/lib/libcova_direct.so: the direct-form 3x3 convolution kernels (tools only, not in the
product library; tools/csrc/conv3x3_direct.hip)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "conv3x3_direct.hip")
LIB = os.path.join(HERE, "lib", "libcova_direct.so")
HEADER = os.path.join(HERE, "include", "cova_direct.h")


def build():
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-fvisibility=hidden", "-shared", SRC, "-o", LIB]
    if os.environ.get("COVA_ABLATE"):
        cmd.insert(1, "-DCOVA_ABLATE=1")
    subprocess.check_call(cmd)
    return LIB


def load():
    import cova_amd  # noqa: F401
    from cova_web_object_detection_amd import _lib
    _lib.lib().load_extra(HEADER, build())


if __name__ == "__main__":
    print(build())
