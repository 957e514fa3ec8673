"""The allocation-free decode entry point against the allocating one, through the C ABI.

lzma_amd_decode_batch_dev_noalloc keeps what its kernels need beyond the caller's buffers
in a device array of the library's own.  Its statuses, sizes and bytes must equal those of
lzma_amd_decode_batch_dev on the same batch, on two calls in a row (the second with a
smaller batch: the array still holds the first call's contents), and a batch beyond its
limit must be refused with a clear error.  The call only enqueues work, so the test waits
for the device itself; around the second call the free device memory must not change.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(data_txt, n, seed):
    """Oracle-encoded ragged streams: several lc/lp/pb (incl. lc+lp > 3, which the entry point
    defers to its second launch), EOS and size-known, one truncated, one with a bad header."""
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    cfgs = [(3, 0, 2), (0, 0, 0), (4, 0, 2), (1, 2, 1), (8, 4, 4), (3, 1, 0), (2, 2, 2)]
    raws, streams = [], []
    for i in range(n):
        k = int(rng.integers(0, 40000)) if i % 7 else int(rng.integers(0, 6))
        s = int(rng.integers(0, len(data_txt) - k))
        raw = data_txt[s:s + k]
        lc, lp, pb = cfgs[i % len(cfgs)]
        p = O.params(dict_log=16 + i % 5, fb=32 + (i * 11) % 200, lc=lc, lp=lp, pb=pb, eos=bool(i & 1))
        raws.append(raw)
        streams.append(O.encode(raw, p))
    streams[1] = streams[1][: max(14, len(streams[1]) // 2)]  # truncated payload
    streams[2] = bytes([0xFE]) + streams[2][1:]               # props byte out of range
    return raws, streams


class _Hip:
    """Device buffers through the HIP runtime the library already uses (no torch: a second HIP
    runtime in the process would not see the GPU)."""

    def __init__(self):
        import ctypes as C
        from lzma_amd import _native as N
        N.lib()
        path = None
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
        assert path, "the HIP runtime is not loaded"
        self.C, self.h = C, C.CDLL(path)
        self.h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.h.hipFree.argtypes = [C.c_void_p]
        self.h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.bufs = []

    def free_bytes(self):
        a, b = self.C.c_size_t(0), self.C.c_size_t(0)
        assert self.h.hipMemGetInfo(self.C.byref(a), self.C.byref(b)) == 0
        return a.value

    def sync(self):
        assert self.h.hipDeviceSynchronize() == 0

    def alloc(self, nbytes, fill=None):
        p = self.C.c_void_p()
        assert self.h.hipMalloc(self.C.byref(p), max(int(nbytes), 16)) == 0
        self.bufs.append(p)
        if fill is not None:
            assert self.h.hipMemset(p, fill, max(int(nbytes), 16)) == 0
        return p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        if arr.nbytes:
            assert self.h.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        return p

    def download(self, p, dtype, count):
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            assert self.h.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0
        return out

    def free(self):
        for p in self.bufs:
            self.h.hipFree(p)
        self.bufs = []


def _run(entry, streams, caps):
    from lzma_amd import _native as N
    hip = _Hip()
    n = len(streams)
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in streams], out=in_off[1:])
    out_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    try:
        d_in = hip.upload(np.frombuffer(b"".join(streams), dtype=np.uint8))
        d_in_off, d_out_off = hip.upload(in_off), hip.upload(out_off)
        d_out = hip.alloc(int(out_off[-1]), fill=0)
        d_sizes, d_status = hip.alloc(8 * n, fill=0xFF), hip.alloc(4 * n, fill=0xFF)
        # every buffer exists before the call under test
        hip.sync()
        free0 = hip.free_bytes()
        N.check(entry(d_in, d_in_off, n, d_out, d_out_off, d_sizes, d_status, None))
        hip.sync()  # the allocation-free entry point returns before the device is done
        free1 = hip.free_bytes()
        sizes = hip.download(d_sizes, np.uint64, n)
        status = hip.download(d_status, np.int32, n)
        out = hip.download(d_out, np.uint8, int(out_off[-1]))
    finally:
        hip.free()
    outs = [out[int(out_off[i]): int(out_off[i]) + int(sizes[i])].tobytes() for i in range(n)]
    return status, sizes, outs, free0 - free1


def test_no_alloc_decode_equals_allocating_decode(data_txt):
    from lzma_amd import _native as N
    L = N.lib()
    assert L.lzma_amd_device_count() >= 1
    for n, seed in ((150, 1), (37, 2)):  # two calls in a row, the second smaller
        raws, streams = _batch(data_txt, n, seed)
        caps = [len(r) + 32 for r in raws]
        caps[5] = max(0, len(raws[5]) - 10)  # too small: ERR_OUT_CAP from both
        st_a, sz_a, out_a, _ = _run(L.lzma_amd_decode_batch_dev, streams, caps)
        st_b, sz_b, out_b, taken = _run(L.lzma_amd_decode_batch_dev_noalloc, streams, caps)
        if seed == 2:  # the array's address is known by now: the call must take no device memory
            assert taken == 0, f"free device memory changed by {taken} bytes across the call"
        assert np.array_equal(st_a, st_b), (st_a, st_b)
        assert np.array_equal(sz_a, sz_b)
        assert out_a == out_b
        assert st_b[1] == N.ERR_TRUNCATED and st_b[2] == N.ERR_HEADER and st_b[5] == N.ERR_OUT_CAP
        ok = [i for i in range(n) if i not in (1, 2, 5)]
        assert all(st_b[i] == 0 and out_b[i] == raws[i] for i in ok)
        assert any(st_b[i] == 0 and i % 7 == 4 for i in ok)  # an lc 8 / lp 4 stream went through the second launch


def test_no_alloc_decode_refuses_oversized_batches():
    from lzma_amd import _native as N
    L = N.lib()
    hip = _Hip()
    try:
        d = hip.alloc(512, fill=0)
        rc = L.lzma_amd_decode_batch_dev_noalloc(d, d, (1 << 20) + 1, d, d, d, d, None)
    finally:
        hip.free()
    assert rc == N.ERR_ARG
    assert b"at most" in L.lzma_amd_last_error()
