"""ctypes loader for the CPU oracle (oracle/liblzma_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg -- never by the lzma_amd package.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "liblzma_oracle.so"

OK, ERR_HEADER, ERR_STREAM, ERR_TRUNCATED, ERR_OUT_CAP, ERR_ARG, ERR_NOMEM = range(7)


class Params(C.Structure):
    _fields_ = [("dict_log", C.c_uint32), ("fb", C.c_uint32), ("lc", C.c_uint32),
                ("lp", C.c_uint32), ("pb", C.c_uint32), ("cut_value", C.c_uint32),
                ("eos", C.c_int32)]


def build(force: bool = False) -> Path:
    src = _DIR / "lzma_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_DIR), "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _SO.exists():
            build()
        L = C.CDLL(str(_SO))
        u8p, u32p, szp = C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)
        L.lzo_params_from_level.argtypes = [C.c_int, C.c_int, C.POINTER(Params)]
        L.lzo_encode_bound.restype = C.c_size_t
        L.lzo_encode_bound.argtypes = [C.c_size_t]
        L.lzo_encode.argtypes = [u8p, C.c_size_t, C.POINTER(Params), u8p, C.c_size_t, szp]
        L.lzo_decode.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, szp, szp]
        L.lzo_bt4_matches.argtypes = [u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                      u32p, u32p, C.c_size_t, szp]
        _lib = L
    return _lib


def params(level=None, *, dict_log=None, fb=None, lc=3, lp=0, pb=2, cut=0, eos=True) -> Params:
    p = Params()
    if level is not None:
        rc = lib().lzo_params_from_level(level, int(eos), C.byref(p))
        if rc:
            raise ValueError(f"level out of range: {level}")
        if dict_log is not None:
            p.dict_log = dict_log
        if fb is not None:
            p.fb = fb
        p.cut_value = cut
        return p
    p.dict_log, p.fb, p.lc, p.lp, p.pb, p.cut_value, p.eos = dict_log, fb, lc, lp, pb, cut, int(eos)
    return p


def _ptr(a):
    return a.ctypes.data if a.size else None


def encode(raw: bytes, p: Params) -> bytes:
    src = np.frombuffer(bytes(raw), dtype=np.uint8)
    cap = lib().lzo_encode_bound(len(raw))
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().lzo_encode(_ptr(src), len(raw), C.byref(p), out.ctypes.data, cap, C.byref(n))
    if rc:
        raise RuntimeError(f"oracle encode failed: status {rc}")
    return out[: n.value].tobytes()


def decode(stream: bytes, out_cap: int = None):
    """Returns (status, bytes)."""
    src = np.frombuffer(bytes(stream), dtype=np.uint8)
    if out_cap is None:
        out_cap = 64 + 64 * len(stream)
        if len(stream) >= 13:
            sz = int.from_bytes(stream[5:13], "little", signed=True)
            if 0 <= sz < (1 << 34):
                out_cap = sz + 512
    out = np.empty(max(out_cap, 1), dtype=np.uint8)
    n = C.c_size_t(0)
    used = C.c_size_t(0)
    rc = lib().lzo_decode(_ptr(src), len(stream), out.ctypes.data, out_cap, C.byref(n), C.byref(used))
    return rc, out[: n.value].tobytes()


def bt4_matches(data: bytes, dict_size: int, fb: int, cut: int = 0):
    """Returns (counts[n] u32, pairs[total,2] u32) with pairs = (len, dist-1)."""
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    n = len(data)
    counts = np.zeros(max(n, 1), dtype=np.uint32)
    total = C.c_size_t(0)
    rc = lib().lzo_bt4_matches(_ptr(src), n, dict_size, fb, cut, counts.ctypes.data, None, 0,
                               C.byref(total))
    if rc:
        raise RuntimeError(f"oracle bt4 failed: status {rc}")
    pairs = np.zeros((max(total.value, 1), 2), dtype=np.uint32)
    rc = lib().lzo_bt4_matches(_ptr(src), n, dict_size, fb, cut, counts.ctypes.data,
                               pairs.ctypes.data, total.value * 2, C.byref(total))
    if rc:
        raise RuntimeError(f"oracle bt4 failed: status {rc}")
    return counts[:n], pairs[: total.value]
