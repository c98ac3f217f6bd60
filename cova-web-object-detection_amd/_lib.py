"""ctypes binding of libcova_hip.so, generated from include/cova_hip.h (single source of truth).

The product path has NO fallback: if the HIP library is missing or a call fails, an exception
is raised (never a silent PyTorch/CPU substitute).
"""
import ctypes
import os
import re

import torch  # noqa: F401  -- must be imported first so that libamdhip64 is torch's copy

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(PKG_DIR, "..", "include", "cova_hip.h")
LIB_PATH = os.environ.get("COVA_HIP_LIB") or os.path.join(PKG_DIR, "lib", "libcova_hip.so")   # env: A/B builds

_CTYPES = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double,
    "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong,
}


def parse_header(path=HEADER):
    """-> {name: [ctype, ...]} for every `int cova_*(...)` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(cova_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(1), m.group(2)
        types = []
        for a in [a.strip() for a in args.split(",") if a.strip()]:
            if a == "void":                          # f(void)
                continue
            if "*" in a:
                types.append(ctypes.c_void_p)
                continue
            a = re.sub(r"\bconst\b", "", a).strip()
            base = " ".join(a.split()[:-1])          # drop the parameter name
            types.append(_CTYPES[base])
        protos[name] = types
    return protos


class CovaHipError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise CovaHipError(
                "libcova_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.fn = {}
        for name, types in self.protos.items():
            f = getattr(self.cdll, name)       # AttributeError => header/library mismatch
            f.argtypes = types
            f.restype = ctypes.c_int
            self.fn[name] = f
        if os.environ.get("COVA_CONV1_F32") == "1":      # A/B: conv1 on the f32-MFMA kernels instead of the bf16-split ones
            self.cdll.cova_set_option(7, 1)
        if os.environ.get("COVA_W4_F32") == "1":         # A/B: F(4x4,3x3) forward / data gradient on the f32-MFMA main loop
            self.cdll.cova_set_option(9, 1)

    def load_extra(self, header, lib_path):
        """Register the entry points of another C-ABI library (tools-only probes) under the same call()."""
        cdll = ctypes.CDLL(lib_path)
        for name, types in parse_header(header).items():
            f = getattr(cdll, name)
            f.argtypes = types
            f.restype = ctypes.c_int
            self.fn[name] = f


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def _arg(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


# Optional per-entry-point timing with HIP events on the launching stream (bench.py's roofline
# leg): PROFILE = {entry point name: [(start_event, end_event, integer arguments of the call, which arguments were
# non-NULL -- one bool per positional argument: the kernel variant a launch took follows from it), ...]}
PROFILE = None


def _device_of(name, args):
    """The one ROCm device all tensor arguments live on (mixed devices are an error)."""
    dev = None
    for a in args:
        if isinstance(a, torch.Tensor):
            if not a.is_cuda:
                raise CovaHipError("%s: got a %s tensor; the hot path runs on the ROCm device only "
                                   "(no CPU fallback)" % (name, a.device))
            if dev is None:
                dev = a.device
            elif a.device != dev:
                raise CovaHipError("%s: tensor arguments on different devices (%s, %s)" % (name, dev, a.device))
    return dev


_Tensor = torch.Tensor


def call(name, *args, stream=None):
    """Call a stream-taking entry point with tensors/None/scalars; raises on a non-zero status.

    The launch goes to the device the tensors live on, on THAT device's current stream: the
    reference picks ``cuda:<-d>`` (main.py:17, evaluate.py:92) and never calls set_device, so the
    process' current device may well be another GPU.

    One pass over the arguments (tensor -> pointer, device agreement): this function sits in front of every launch, and behind
    a host read (the reference's loop has two per step, train.py:54,57) the head's 5-20 us kernels wait for it."""
    L = _LIB if _LIB is not None else lib()
    dev = -1
    cargs = []
    for a in args:
        if isinstance(a, _Tensor):
            if not a.is_cuda:
                raise CovaHipError("%s: got a %s tensor; the hot path runs on the ROCm device only "
                                   "(no CPU fallback)" % (name, a.device))
            d = a.get_device()
            if dev < 0:
                dev = d
            elif d != dev:
                raise CovaHipError("%s: tensor arguments on different devices (cuda:%d, cuda:%d)" % (name, dev, d))
            cargs.append(a.data_ptr())
        else:
            cargs.append(a)
    if dev >= 0 and dev != torch.cuda.current_device():
        with torch.cuda.device(dev):
            return _launch(L, name, args, cargs, stream)
    return _launch(L, name, args, cargs, stream)


def _launch(L, name, args, cargs, stream):
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    prof = PROFILE.get(name) if PROFILE is not None else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = L.fn[name](*cargs, stream)
    if prof is not None:
        e1.record()
        prof.append((e0, e1, tuple(a for a in args if isinstance(a, int)), tuple(a is not None for a in args)))
    if rc != 0:
        raise CovaHipError("%s failed with status %d" % (name, rc))


def query(name, *args):
    """Call a pure host query (no stream, returns its int result)."""
    return lib().fn[name](*args)
