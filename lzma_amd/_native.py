"""ctypes binding of liblzma_amd.so (the C ABI declared in include/lzma_amd.h).

The library is the product: if it is missing or cannot be loaded this module
raises -- there is no CPU fallback anywhere in this package.
"""
import ctypes as C
from pathlib import Path

import os

# LZMA_AMD_LIB selects another build of the same library (diagnostic builds)
_SO = Path(os.environ.get("LZMA_AMD_LIB") or Path(__file__).resolve().parent / "liblzma_amd.so")

OK, ERR_HEADER, ERR_STREAM, ERR_TRUNCATED, ERR_OUT_CAP, ERR_ARG, ERR_NOMEM, ERR_HIP, \
    ERR_UNSUPPORTED = range(9)

STATUS_NAMES = {OK: "ok", ERR_HEADER: "header", ERR_STREAM: "stream", ERR_TRUNCATED: "truncated",
                ERR_OUT_CAP: "out_cap", ERR_ARG: "arg", ERR_NOMEM: "nomem", ERR_HIP: "hip",
                ERR_UNSUPPORTED: "unsupported"}

EXPORTS = [
    "lzma_amd_params_from_level", "lzma_amd_device_count", "lzma_amd_set_device",
    "lzma_amd_last_error", "lzma_amd_version", "lzma_amd_encode_bound", "lzma_amd_bt4_matches",
    "lzma_amd_bt4_matches_dev", "lzma_amd_encode_chunks", "lzma_amd_encode_chunks_dev",
    "lzma_amd_decode_batch", "lzma_amd_decode_batch_dev", "lzma_amd_parse_header",
    "lzma_amd_last_timing", "lzma_amd_last_coder_launches", "lzma_amd_last_coop_launches", "lzma_amd_release_workspace",
    "lzma_amd_decode_batch_dev_noalloc",
]


class Params(C.Structure):
    """lzma_amd_params: explicit form of the reference's compressionLevel."""
    _fields_ = [("dict_log", C.c_uint32), ("fb", C.c_uint32), ("lc", C.c_uint32),
                ("lp", C.c_uint32), ("pb", C.c_uint32), ("cut_value", C.c_uint32),
                ("eos", C.c_int32)]


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lzma_amd status {code} ({STATUS_NAMES.get(code, '?')}): {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not _SO.exists():
        raise ImportError(f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(str(_SO))
    vp, sz = C.c_void_p, C.c_size_t
    L.lzma_amd_params_from_level.argtypes = [C.c_int, C.c_int, C.POINTER(Params)]
    L.lzma_amd_device_count.restype = C.c_int
    L.lzma_amd_set_device.argtypes = [C.c_int]
    L.lzma_amd_last_error.restype = C.c_char_p
    L.lzma_amd_version.restype = C.c_char_p
    L.lzma_amd_encode_bound.restype = sz
    L.lzma_amd_encode_bound.argtypes = [sz]
    L.lzma_amd_bt4_matches.argtypes = [vp, sz, C.c_uint32, C.c_uint32, vp, vp, vp, sz,
                                       C.POINTER(sz)]
    L.lzma_amd_bt4_matches_dev.argtypes = [vp, sz, C.c_uint32, C.c_uint32, vp, vp, vp, sz, vp, vp]
    L.lzma_amd_encode_chunks.argtypes = [vp, vp, sz, C.POINTER(Params), vp, vp, vp, vp]
    L.lzma_amd_encode_chunks_dev.argtypes = [vp, vp, sz, sz, C.POINTER(Params), vp, vp, vp, vp, vp]
    L.lzma_amd_decode_batch.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    L.lzma_amd_decode_batch_dev.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp]
    L.lzma_amd_decode_batch_dev_noalloc.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp]
    L.lzma_amd_parse_header.argtypes = [vp, sz] + [C.POINTER(C.c_uint32)] * 4 + [C.POINTER(C.c_int64)]
    L.lzma_amd_last_timing.argtypes = [C.POINTER(C.c_float)] * 3
    L.lzma_amd_last_coder_launches.restype = C.c_int
    L.lzma_amd_last_coop_launches.restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise NativeError(rc, lib().lzma_amd_last_error().decode("utf-8", "replace"))


def release_workspace():
    """Give back the device memory the calling thread's encoder keeps between calls."""
    lib().lzma_amd_release_workspace()


def last_timing():
    a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
    lib().lzma_amd_last_timing(C.byref(a), C.byref(b), C.byref(c))
    return {"total_ms": a.value, "finder_ms": b.value, "coder_ms": c.value,
            "coder_launches": int(lib().lzma_amd_last_coder_launches()),
            "coop_launches": int(lib().lzma_amd_last_coop_launches())}
