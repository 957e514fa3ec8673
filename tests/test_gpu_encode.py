"""GPU encoder parity, through the C ABI.

The GPU encoder uses the exhaustive match finder, so its output must equal, byte
for byte, the oracle's encoder run with the cut value raised (the same code that
reproduces the reference's vectors at the default cut value).  Every stream must
also round-trip through the oracle's decoder and through liblzma, and its size
must be within 1 % of the oracle at the reference's default cut value.
Mirrors TestEncoder / TestEmpty / TestBoth (lzma_encoder_test.go:30-124).
"""
import io
import lzma as pylzma

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
CUT_INF = 1 << 30


@pytest.fixture(autouse=True, params=["one wave per chunk", "two waves per chunk", "three waves per chunk"])
def waves_per_chunk(request, monkeypatch):
    """Every test of this file runs under all three encoder kernels: the library reads LZMA_AMD_COOP
    at each call (0: one wave per chunk, 2 / 3: the cooperative kernels; unset: by chunk count)."""
    monkeypatch.setenv("LZMA_AMD_COOP", {"one": "0", "two": "2", "three": "3"}[request.param.split()[0]])


def _check(chunks, *, dict_log, fb, lc=3, lp=0, pb=2, eos=False, ratio_check=True):
    import lzma_amd
    from oracle import oracle as O
    p = lzma_amd.make_params(dict_log, fb, lc, lp, pb, eos)
    status, outs = lzma_amd.encode_chunks(chunks, p)
    assert (status == 0).all(), status
    tot_gpu = tot_ref = 0
    for raw, enc in zip(chunks, outs):
        raw = bytes(raw)
        want = O.encode(raw, O.params(dict_log=dict_log, fb=fb, lc=lc, lp=lp, pb=pb, cut=CUT_INF, eos=eos))
        assert enc == want, (len(raw), len(enc), len(want))
        rc, back = O.decode(enc)
        assert rc == 0 and back == raw
        if lc + lp <= 4:
            assert pylzma.decompress(enc, format=pylzma.FORMAT_ALONE) == raw
        if ratio_check:
            tot_gpu += len(enc)
            tot_ref += len(O.encode(raw, O.params(dict_log=dict_log, fb=fb, lc=lc, lp=lp, pb=pb, cut=0, eos=eos)))
    if ratio_check and tot_ref > 1000:
        assert tot_gpu <= 1.01 * tot_ref, (tot_gpu, tot_ref)
    return outs


def test_reference_vectors_level3(vectors):
    """Same inputs as TestEncoder; sizes within 1 % and, for the short ones, identical bytes."""
    import lzma_amd
    for v in vectors:
        if v["err"]:
            continue
        raw, want = bytes.fromhex(v["raw_hex"]), bytes.fromhex(v["lzma_hex"])
        p = lzma_amd.params_from_level(v["level"], eos=not v["size_known"])
        status, outs = lzma_amd.encode_chunks([raw], p)
        assert status[0] == 0
        assert len(outs[0]) <= len(want) * 1.01 + 1, v["descr"]
        if len(raw) < 100:
            assert outs[0] == want, v["descr"]


def test_data_txt_level3(data_txt, data_lzma):
    outs = _check([data_txt], dict_log=20, fb=64, eos=True)
    assert abs(len(outs[0]) - len(data_lzma)) <= 0.01 * len(data_lzma)


def test_ragged_batch(data_txt):
    rng = np.random.default_rng(3)
    chunks = [b"", b"a", b"ab", b"abc", b"abcd", b"lzmalzmalzma"]
    for i in range(20):
        n = int(rng.integers(1, 40000))
        s = int(rng.integers(0, len(data_txt) - n))
        chunks.append(data_txt[s:s + n])
    _check(chunks, dict_log=22, fb=128)
    _check(chunks, dict_log=20, fb=64, eos=True)


@pytest.mark.parametrize("cfg", [dict(dict_log=12, fb=5, lc=0, lp=0, pb=0), dict(dict_log=16, fb=273, lc=8, lp=4, pb=4),
                                 dict(dict_log=14, fb=32, lc=4, lp=1, pb=3), dict(dict_log=26, fb=273),
                                 dict(dict_log=13, fb=7, lc=1, lp=2, pb=1)])
def test_param_corners(cfg, data_txt):
    _check([data_txt[:50000], data_txt[100000:130000]], eos=True, **cfg)


def test_markov_chunks_level5_shape():
    buf = synth.markov(3 << 18)
    chunks = [bytes(buf[i:i + (1 << 18)]) for i in range(0, len(buf), 1 << 18)]
    _check(chunks, dict_log=22, fb=128)


def test_log_long_matches():
    buf = synth.logtext(1 << 19, block=1 << 19)
    _check([bytes(buf)], dict_log=26, fb=273)


def test_binary_and_runs():
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.integers(0, 256, 30000, dtype=np.uint8), np.zeros(50000, np.uint8),
                        rng.integers(0, 3, 40000, dtype=np.uint8)])
    _check([bytes(a)], dict_log=16, fb=64)


def test_api_writer_round_trip(data_txt):
    """TestBoth / TestEmpty through the NewWriter*/NewReader mirror."""
    import lzma_amd
    for payload in (b"", b"lzmalzmalzma", data_txt[:5000]):
        for size in (-1, len(payload)):
            buf = io.BytesIO()
            w = lzma_amd.NewWriterSizeLevel(buf, size, 3)
            w.write(payload)
            w.close()
            assert lzma_amd.NewReader(io.BytesIO(buf.getvalue())).read() == payload
    with pytest.raises(lzma_amd.ArgumentValueError):
        w = lzma_amd.NewWriterSizeLevel(io.BytesIO(), -1, 10)
        w.write(b"x")


def test_bad_params():
    import lzma_amd
    from lzma_amd import _native as N
    for kw in (dict(dict_log=11, fb=64), dict(dict_log=30, fb=64), dict(dict_log=20, fb=4),
               dict(dict_log=20, fb=274), dict(dict_log=20, fb=64, lc=9), dict(dict_log=20, fb=64, lp=5),
               dict(dict_log=20, fb=64, pb=5)):
        with pytest.raises(N.NativeError) as ei:
            lzma_amd.encode_chunks([b"abc"], lzma_amd.make_params(**kw))
        assert ei.value.code == N.ERR_ARG


def _mutated_repeats(n, block, seed):
    r = np.random.default_rng(seed)
    base = r.integers(97, 123, block, dtype=np.uint8)
    out = np.tile(base, n // block + 1)[:n].copy()
    hits = r.integers(0, n, max(1, n // 97))
    out[hits] = r.integers(97, 123, hits.size, dtype=np.uint8)
    return out


def test_random_parameters():
    """Forty draws of (data kind, length, dictionary, fast bytes, lc, lp, pb, end marker): bytes
    equal to the oracle's at the exhaustive cut value, and both chunks of the call decode back."""
    import lzma_amd
    from oracle import oracle as O
    rng = np.random.default_rng(2026)
    for it in range(40):
        n = int(rng.integers(1, 200000))
        kind = it % 4
        if kind == 0:
            data = synth.markov(1 << 20)[:n]
        elif kind == 1:
            data = synth.logtext(1 << 20, block=1 << 20)[:n]
        elif kind == 2:
            data = _mutated_repeats(n, int(rng.integers(20, 500)), it)
        else:
            data = rng.integers(0, 256, n, dtype=np.uint8) if it % 8 == 3 else np.zeros(n, np.uint8)
        data = bytes(np.asarray(data, dtype=np.uint8))
        dict_log, fb = int(rng.integers(12, 27)), int(rng.choice([5, 8, 16, 32, 64, 128, 255, 256, 273]))
        lc, lp, pb, eos = int(rng.integers(0, 5)), int(rng.integers(0, 3)), int(rng.integers(0, 5)), bool(it % 2)
        what = (it, n, kind, dict_log, fb, lc, lp, pb, eos)
        st, outs = lzma_amd.encode_chunks([data, data[: n // 3]], lzma_amd.make_params(dict_log, fb, lc, lp, pb, eos))
        assert (st == 0).all(), what
        want = O.encode(data, O.params(dict_log=dict_log, fb=fb, lc=lc, lp=lp, pb=pb, cut=1 << 30, eos=eos))
        assert outs[0] == want, what
        st2, dec = lzma_amd.decode_batch(outs)
        assert (st2 == 0).all() and dec == [data, data[: n // 3]], what
