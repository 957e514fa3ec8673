"""The C ABI is callable from many threads at once (one goroutine per stream in the reference,
lzma_encoder.go:1120-1132): every thread keeps streams, staging buffers and workspaces of its own.
Four threads encode, decode and search concurrently; every result equals the one computed alone."""
import threading

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _work(k):
    import lzma_amd
    buf = synth.markov(1 << 20)
    chunks = [bytes(buf[(k * 7 + i) * 40000:(k * 7 + i) * 40000 + 30000 + 9000 * i]) for i in range(6)]
    p = lzma_amd.make_params(16 + 2 * k, [32, 64, 128, 273][k], lc=3 - (k & 1), lp=k & 1, pb=2, eos=bool(k & 1))
    st, outs = lzma_amd.encode_chunks(chunks, p)
    assert (st == 0).all()
    st2, dec = lzma_amd.decode_batch(outs)
    assert (st2 == 0).all() and dec == chunks
    c, pr = lzma_amd.bt4_matches(chunks[5], 1 << (14 + k), [8, 64, 128, 273][k])
    return outs, c.tobytes(), pr.tobytes()


def test_four_threads_equal_the_sequential_results():
    want = [_work(k) for k in range(4)]
    got, errs = [None] * 4, []

    def run(k):
        try:
            for _ in range(3):
                got[k] = _work(k)
                assert got[k] == want[k]
        except BaseException as e:  # reported by the main thread
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=run, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errs, errs
    assert all(not t.is_alive() for t in ts)
    assert got == want
