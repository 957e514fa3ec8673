"""GPU decoder parity: the HIP batched decoder against the reference's golden
streams and against oracle-encoded streams, through the C ABI.

Mirrors TestDecoder (lzma_decoder_test.go:14-32) and the byte-exact check of
BenchmarkDecoder (lzma_decoder_test.go:52).  Bit-exact is the bar.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import lzma_amd
    from lzma_amd import _native as N
    assert N.lib().lzma_amd_device_count() >= 1, "no HIP device visible"
    return lzma_amd


def test_reference_vectors(amd, vectors):
    from lzma_amd import _native as N
    streams = [bytes.fromhex(v["lzma_hex"]) for v in vectors]
    status, outs = amd.decode_batch(streams, out_caps=[4096] * len(streams))
    for v, st, o in zip(vectors, status, outs):
        if v["err"] == "headerError":
            assert st == N.ERR_HEADER, v["descr"]
        else:
            assert st == N.OK and o == bytes.fromhex(v["raw_hex"]), v["descr"]


def test_bench_file(amd, data_txt, data_lzma):
    status, outs = amd.decode_batch([data_lzma], out_caps=[len(data_txt) + 64])
    assert status[0] == 0 and outs[0] == data_txt


def test_api_reader(amd, data_txt, data_lzma):
    import io
    r = amd.NewReader(io.BytesIO(data_lzma))
    assert r.read() == data_txt
    with pytest.raises(amd.HeaderError):
        amd.NewReader(io.BytesIO(bytes([0xFE]) + data_lzma[1:100])).read()


def test_batch_of_oracle_streams(amd, data_txt):
    """Ragged batch: sizes 0..64 KiB, size-known and EOS, several lc/lp/pb, incl. lc+lp > 3."""
    from oracle import oracle as O
    rng = np.random.default_rng(7)
    raws, streams = [], []
    cfgs = [(3, 0, 2), (0, 0, 0), (4, 0, 2), (1, 2, 1), (8, 4, 4), (3, 1, 0)]
    for i in range(48):
        n = int(rng.integers(0, 65536)) if i % 5 else int(rng.integers(0, 8))
        start = int(rng.integers(0, len(data_txt) - n))
        raw = data_txt[start:start + n]
        lc, lp, pb = cfgs[i % len(cfgs)]
        p = O.params(dict_log=16 + i % 5, fb=32 + (i * 7) % 200, lc=lc, lp=lp, pb=pb, eos=bool(i & 1))
        raws.append(raw)
        streams.append(O.encode(raw, p))
    status, outs = amd.decode_batch(streams, out_caps=[len(r) + 16 for r in raws])
    assert (status == 0).all(), status
    for r, o in zip(raws, outs):
        assert r == o


def test_errors(amd, data_lzma, data_txt):
    from lzma_amd import _native as N
    bad = bytearray(data_lzma)
    for i in range(40, 60):
        bad[i] ^= 0xFF
    streams = [data_lzma[:5], data_lzma[:1000], bytes(bad), data_lzma, data_lzma]
    caps = [64, len(data_txt), len(data_txt), 1000, len(data_txt)]
    status, outs = amd.decode_batch(streams, out_caps=caps)
    assert status[0] == N.ERR_TRUNCATED
    assert status[1] == N.ERR_TRUNCATED
    from oracle import oracle as O
    rc, out = O.decode(bytes(bad), out_cap=len(data_txt))
    assert status[2] == rc and (rc != 0 or outs[2] == out)
    assert status[3] == N.ERR_OUT_CAP
    assert status[4] == N.OK and outs[4] == data_txt


def test_truncated_streams_match_oracle(amd, data_txt):
    """A stream cut at any byte: the status and the bytes delivered before the input ran out
    equal the oracle's, whether the cut falls in a literal, in a match or in the end marker,
    and whether or not the missing byte would only have been needed by the last normalisation."""
    from oracle import oracle as O
    raw = data_txt[5000:9000]
    cuts, streams, want = [], [], []
    for eos in (False, True):
        full = O.encode(raw, O.params(dict_log=16, fb=64, lc=3, lp=0, pb=2, eos=eos))
        points = list(range(13, 40)) + list(range(40, len(full) - 24, 53)) + list(range(len(full) - 24, len(full) + 1))
        for c in points:
            s = full[:c]
            cuts.append((eos, c, len(full)))
            streams.append(s)
            want.append(O.decode(s, out_cap=len(raw)))
    status, outs = amd.decode_batch(streams, out_caps=[len(raw)] * len(streams))
    for (eos, c, n), st, o, (rc, ref) in zip(cuts, status, outs, want):
        assert st == rc, (eos, c, n, int(st), rc)
        assert o == ref, (eos, c, n, len(o), len(ref))


def test_highly_compressible_eos_stream_is_retried(amd):
    """A stream that ends with a marker carries no size: the first guess of its output
    capacity (40x the compressed bytes) is far too small for 1 MiB of zeros (about 1 KiB
    compressed), and the batch call must come back with more room instead of failing --
    NewWriter (size -1) to NewReader round trip, as TestBoth does with text."""
    import io
    from oracle import oracle as O
    payloads = [bytes(1 << 20), b"ab" * 300000, b"x"]
    for payload in payloads:
        buf = io.BytesIO()
        w = amd.NewWriterLevel(buf, 3)
        w.write(payload)
        w.close()
        assert len(buf.getvalue()) * 40 + 64 < len(payload) or len(payload) < 100
        assert amd.NewReader(io.BytesIO(buf.getvalue())).read() == payload
    streams = [O.encode(p, O.params(3, eos=True)) for p in payloads]
    status, outs = amd.decode_batch(streams)
    assert (status == 0).all() and outs == payloads


def test_forged_size_field_does_not_size_the_batch(amd, data_txt):
    """A header that claims 2^39 bytes must not make the call allocate them: the capacity is
    clamped to what the compressed bytes can expand to, the other streams of the batch
    decode normally and the forged one ends with a per-stream status."""
    from lzma_amd import _native as N
    from lzma_amd import codec
    from oracle import oracle as O
    good = O.encode(data_txt[:20000], O.params(3, eos=False))
    forged = bytearray(good)
    forged[5:13] = (1 << 39).to_bytes(8, "little")
    assert codec.max_output(len(forged)) < (1 << 28)
    status, outs = amd.decode_batch([good, bytes(forged), good])
    assert status[0] == 0 and status[2] == 0 and outs[0] == outs[2] == data_txt[:20000]
    assert status[1] != 0 and outs[1][:20000] == data_txt[:20000]
