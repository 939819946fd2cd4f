"""ctypes binding of libsed_hip.so (the C ABI declared in include/sed_hip.h).

The product path is HIP-only: if the library is missing or a symbol the header declares is not exported, importing
this module raises -- there is no CPU / PyTorch fallback behind these ops.  Signatures are parsed
from the header itself so the binding cannot drift from the ABI.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsed_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sed_hip.h")

_CTYPES = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double, "hipStream_t": ctypes.c_void_p}


def parse_header(path=HEADER_PATH):
    """-> {name: [(ctype, argname), ...]} for every `int sed_*(...)` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(sed_\w+)\s*\(([^)]*)\)\s*;", src, flags=re.S):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if "*" in a:
                args.append((ctypes.c_void_p, a.split("*")[-1].strip()))
            else:
                ty, name = a.rsplit(" ", 1)
                args.append((_CTYPES[ty.replace("const ", "").strip()], name))
        protos[m.group(1)] = args
    return protos


class SedHipError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m transformer4sed_amd.build` "
                "(the MAT-SED hot path has no non-HIP fallback)")
        # torch must initialise ITS HIP runtime first: loading libsed_hip.so before torch pulls in a second copy of
        # libamdhip64 (system ROCm vs the one bundled with torch) and every launch then fails with hipErrorNoDevice.
        import torch  # noqa: F401
        # SED_HIP_LIB: developer A/B switch to another build of the same library (tools/ablate)
        self._dll = ctypes.CDLL(os.environ.get("SED_HIP_LIB") or LIB_PATH)
        self.protos = parse_header()
        for name, args in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise ImportError(f"libsed_hip.so does not export {name} declared in include/sed_hip.h") from e
            fn.restype = ctypes.c_int
            fn.argtypes = [t for t, _ in args]
            setattr(self, "_raw_" + name, fn)
        want = int(re.search(r"#define\s+SED_HIP_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read()).group(1))
        have = self._dll.sed_abi_version(0)
        if have != want:
            raise ImportError(f"libsed_hip.so implements ABI version {have}, include/sed_hip.h declares {want}: rebuild the library "
                              "(python -m transformer4sed_amd.build --force)")

    def call(self, name, *args):
        rc = getattr(self, "_raw_" + name)(*args)
        if rc > 0 and name.startswith("sed_debug_"):
            return rc           # (test aids return counts)
        if rc != 0:
            what = "bad argument" if rc == -1 else (f"HIP error {-rc - 1000}" if rc <= -1000 else "HIP launch failure")
            raise SedHipError(f"{name} failed with code {rc} ({what})")


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
