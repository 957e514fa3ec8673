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


# An LZMA stream cannot expand without bound: the cheapest symbol is a 273-byte rep0 match
# coded in 14 binary decisions, none cheaper than log2(2048/2017) bits (a model saturates at
# 2017/2048), i.e. fewer than 7100 output bytes per compressed byte.  Capacities derived
# from an untrusted header are clamped to this, so a forged size field cannot make a batch
# allocate terabytes.
MAX_EXPANSION = 7200
MAX_STREAM_OUT = (1 << 32) - 512  # what one stream may produce (include/lzma_amd.h)


def max_output(stream_len: int) -> int:
    return min(MAX_STREAM_OUT, 64 + MAX_EXPANSION * int(stream_len))


def header_sizes(streams):
    """Size field of each stream's header: None for -1 (end marker) or a short stream."""
    caps = []
    for s in streams:
        cap = None
        if len(s) >= 13:
            sz = int.from_bytes(s[5:13], "little", signed=True)
            if sz >= 0:
                cap = sz
        caps.append(cap)
    return caps


def _decode_once(streams, caps):
    n = len(streams)
    in_off = _offsets([len(s) for s in streams])
    out_off = _offsets(caps)
    src = np.frombuffer(b"".join(bytes(s) for s in streams), dtype=np.uint8)
    out = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    N.check(N.lib().lzma_amd_decode_batch(_ptr(src), in_off.ctypes.data, n, out.ctypes.data,
                                          out_off.ctypes.data, sizes.ctypes.data,
                                          status.ctypes.data))
    res = [out[int(out_off[i]): int(out_off[i]) + int(sizes[i])].tobytes() for i in range(n)]
    return status, res


def decode_batch(streams, out_caps=None, max_batch_bytes=8 << 30):
    """Decode complete .lzma streams.  Returns (statuses[int32], [bytes]).

    Output capacity per stream: the header's size field when it has one (clamped to what
    the compressed length can possibly expand to); for streams that end with a marker
    instead, 40x the compressed size at first, and every stream that reports ERR_OUT_CAP is
    submitted again with eight times the room until it fits or reaches that clamp.  Batches
    whose capacities add up to more than max_batch_bytes are decoded in several calls.
    Explicit out_caps are used as given, without retry."""
    n = len(streams)
    if n == 0:
        return np.zeros(0, np.int32), []
    status = np.zeros(n, dtype=np.int32)
    res = [b""] * n
    if out_caps is not None:
        caps = [int(c) for c in out_caps]
        grow = [False] * n
    else:
        hs = header_sizes(streams)
        caps = [min(h, max_output(len(s))) if h is not None else min(64 + 40 * len(s), max_output(len(s)))
                for h, s in zip(hs, streams)]
        grow = [h is None for h in hs]
    todo = list(range(n))
    while todo:
        # groups of streams whose capacities fit the per-call budget
        groups, cur, tot = [], [], 0
        for i in todo:
            if cur and tot + caps[i] > max_batch_bytes:
                groups.append(cur)
                cur, tot = [], 0
            cur.append(i)
            tot += caps[i]
        groups.append(cur)
        again = []
        for g in groups:
            st, out = _decode_once([streams[i] for i in g], [caps[i] for i in g])
            for k, i in enumerate(g):
                status[i], res[i] = st[k], out[k]
                if st[k] == N.ERR_OUT_CAP and grow[i] and caps[i] < max_output(len(streams[i])):
                    caps[i] = min(8 * caps[i], max_output(len(streams[i])))
                    again.append(i)
        todo = again
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
