"""Builds and loads tools/lib/libcova_f2x2.so: the F(2x2,3x3) Winograd kernels of rounds 1-3 (tools/csrc/conv_wino_f2x2.hip) as a
TEST-SUPPORT library -- the product library does not contain them; the F(4x4) kernel tests cross-check against them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "conv_wino_f2x2.hip")
LIB = os.path.join(HERE, "lib", "libcova_f2x2.so")
HEADER = os.path.join(HERE, "include", "cova_wino_f2x2.h")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-fvisibility=hidden", "-fno-slp-vectorize", "-shared", SRC, "-o", LIB]
    if os.environ.get("COVA_ABLATE"):
        cmd.insert(1, "-DCOVA_ABLATE=1")
    subprocess.check_call(cmd)
    return LIB


def load():
    """Register the F(2x2) entry points under _lib.call / _lib.query, and forward the test hooks of cova_set_option (2 = grid cap,
    6 = tile geometry) to this library's own cova_wino_f2x2_set_option as well."""
    import cova_amd  # noqa: F401
    from cova_web_object_detection_amd import _lib
    L = _lib.lib()
    if "cova_wino_f2x2_set_option" in L.fn:
        return
    if not os.path.exists(LIB):
        raise _lib.CovaHipError("%s not found: run __graft_entry__.build()" % LIB)
    L.load_extra(HEADER, LIB)
    product = L.fn["cova_set_option"]

    def set_option(key, value):
        if key in (5, 6):
            return L.fn["cova_wino_f2x2_set_option"](key, value)
        rc = product(key, value)
        if key == 2:
            L.fn["cova_wino_f2x2_set_option"](key, value)
        return rc
    L.fn["cova_set_option"] = set_option


if __name__ == "__main__":
    print(build(force=True))
