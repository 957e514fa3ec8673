"""Batch codec calls over host buffers (numpy / bytes) through the C ABI."""
import ctypes as C

import numpy as np

from . import _native as N
from ._native import Params


def params_from_level(level: int, eos: bool = True) -> Params:
    """levels[level] of the reference (lzma_encoder.go:97-108)."""
    p = Params()
    N.check(N.lib().lzma_amd_params_from_level(int(level), int(bool(eos)), C.byref(p)))
    return p


def make_params(dict_log: int, fb: int, lc: int = 3, lp: int = 0, pb: int = 2,
                eos: bool = False) -> Params:
    p = Params()
    p.dict_log, p.fb, p.lc, p.lp, p.pb, p.cut_value, p.eos = dict_log, fb, lc, lp, pb, 0, int(eos)
    return p


def _offsets(sizes):
    off = np.zeros(len(sizes) + 1, dtype=np.uint64)
    np.cumsum(np.asarray(sizes, dtype=np.uint64), out=off[1:])
    return off


def _ptr(a):
    return a.ctypes.data if a.size else None


def header_sizes(streams):
    """Output capacity needed per stream: header size field when present."""
    caps = []
    for s in streams:
        cap = None
        if len(s) >= 13:
            sz = int.from_bytes(s[5:13], "little", signed=True)
            if 0 <= sz < (1 << 40):
                cap = sz
        caps.append(cap)
    return caps


def decode_batch(streams, out_caps=None):
    """Decode complete .lzma streams.  Returns (statuses[int32], [bytes])."""
    n = len(streams)
    if n == 0:
        return np.zeros(0, np.int32), []
    if out_caps is None:
        out_caps = [c if c is not None else 64 + 40 * len(s)
                    for c, s in zip(header_sizes(streams), streams)]
    in_off = _offsets([len(s) for s in streams])
    out_off = _offsets(out_caps)
    src = np.frombuffer(b"".join(bytes(s) for s in streams), dtype=np.uint8)
    out = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    N.check(N.lib().lzma_amd_decode_batch(_ptr(src), in_off.ctypes.data, n, out.ctypes.data,
                                          out_off.ctypes.data, sizes.ctypes.data,
                                          status.ctypes.data))
    res = [out[int(out_off[i]): int(out_off[i]) + int(sizes[i])].tobytes() for i in range(n)]
    return status, res


def encode_chunks(chunks, p: Params):
    """Encode each chunk as an independent .lzma stream.  Returns (statuses, [bytes])."""
    n = len(chunks)
    if n == 0:
        return np.zeros(0, np.int32), []
    L = N.lib()
    in_off = _offsets([len(c) for c in chunks])
    out_off = _offsets([L.lzma_amd_encode_bound(len(c)) for c in chunks])
    src = np.frombuffer(b"".join(bytes(c) for c in chunks), dtype=np.uint8)
    out = np.empty(int(out_off[-1]), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    N.check(L.lzma_amd_encode_chunks(_ptr(src), in_off.ctypes.data, n, C.byref(p), out.ctypes.data,
                                     out_off.ctypes.data, sizes.ctypes.data, status.ctypes.data))
    res = [out[int(out_off[i]): int(out_off[i]) + int(sizes[i])].tobytes() for i in range(n)]
    return status, res


def bt4_matches(data: bytes, dict_size: int, fb: int):
    """Exhaustive BT4 match lists.  Returns (counts[n] u32, pairs[total,2] u32 = (len, dist-1))."""
    n = len(data)
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    counts = np.zeros(max(n, 1), dtype=np.uint32)
    total = C.c_size_t(0)
    cap = max(16, 8 * n)
    L = N.lib()
    while True:
        pairs = np.zeros((cap, 2), dtype=np.uint32)
        rc = L.lzma_amd_bt4_matches(_ptr(src), n, dict_size, fb, counts.ctypes.data, None,
                                    pairs.ctypes.data, cap, C.byref(total))
        if rc == N.ERR_OUT_CAP and total.value > cap:
            cap = total.value
            continue
        N.check(rc)
        return counts[:n], pairs[: total.value]
