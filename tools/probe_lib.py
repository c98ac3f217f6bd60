"""Builds and loads tools/lib/libcova_probe.so (micro-benchmark probes; tools only, not in the product library)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "probe.hip")
LIB = os.path.join(HERE, "lib", "libcova_probe.so")
HEADER = os.path.join(HERE, "include", "cova_probe.h")


def build():
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-fPIC", "-fvisibility=hidden", "-shared", SRC, "-o", LIB])
    return LIB


def load():
    import cova_amd  # noqa: F401
    from cova_web_object_detection_amd import _lib
    _lib.lib().load_extra(HEADER, build())


if __name__ == "__main__":
    print(build())
