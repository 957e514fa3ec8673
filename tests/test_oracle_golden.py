"""Pins the CPU oracle to every known-answer vector the reference's tests hold.

Mirrors TestEncoder (lzma_encoder_test.go:93-124), TestDecoder
(lzma_decoder_test.go:14-32), the byte-exact checks inside BenchmarkEncoder /
BenchmarkDecoder (lzma_encoder_test.go:154, lzma_decoder_test.go:52) and the
round trips TestEmpty1/2, TestBoth1/2 (lzma_encoder_test.go:30-91).
"""
import lzma as pylzma

import pytest

from oracle import oracle as O


def test_encoder_vectors(vectors):
    n = 0
    for v in vectors:
        if v["err"]:
            continue
        raw, want = bytes.fromhex(v["raw_hex"]), bytes.fromhex(v["lzma_hex"])
        got = O.encode(raw, O.params(v["level"], eos=not v["size_known"]))
        assert got == want, v["descr"]
        n += 1
    assert n == 6


def test_decoder_vectors(vectors):
    for v in vectors:
        rc, out = O.decode(bytes.fromhex(v["lzma_hex"]))
        if v["err"] == "headerError":
            assert rc == O.ERR_HEADER, v["descr"]
        else:
            assert rc == O.OK and out == bytes.fromhex(v["raw_hex"]), v["descr"]


def test_bench_file_encode(data_txt, data_lzma):
    assert O.encode(data_txt, O.params(3, eos=True)) == data_lzma


def test_bench_file_decode(data_txt, data_lzma):
    rc, out = O.decode(data_lzma)
    assert rc == O.OK and out == data_txt


@pytest.mark.parametrize("payload", [b"", b"lzmalzmalzma"])
@pytest.mark.parametrize("eos", [False, True])
def test_round_trip_level3(payload, eos):
    enc = O.encode(payload, O.params(3, eos=eos))
    rc, out = O.decode(enc)
    assert rc == O.OK and out == payload


@pytest.mark.parametrize("level", range(1, 10))
def test_levels_decode_with_liblzma(level, data_txt):
    """Independent decoder (liblzma) accepts every level's output."""
    raw = data_txt[:60000]
    enc = O.encode(raw, O.params(level, eos=True))
    assert pylzma.decompress(enc, format=pylzma.FORMAT_ALONE) == raw
    rc, out = O.decode(enc)
    assert rc == O.OK and out == raw


def test_truncated_and_corrupt(data_lzma):
    rc, _ = O.decode(data_lzma[:5])
    assert rc == O.ERR_TRUNCATED
    rc, _ = O.decode(data_lzma[:1000])
    assert rc == O.ERR_TRUNCATED
    bad = bytearray(data_lzma)
    for i in range(40, 60):
        bad[i] ^= 0xFF
    rc, out = O.decode(bytes(bad))
    assert rc != O.OK or out != b""  # must not crash; corrupt data is detected or garbage


def test_bad_args():
    with pytest.raises(ValueError):
        O.params(0)
    with pytest.raises(RuntimeError):
        O.encode(b"x", O.params(dict_log=11, fb=64))
    with pytest.raises(RuntimeError):
        O.encode(b"x", O.params(dict_log=20, fb=4))
