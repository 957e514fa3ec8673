"""CPU-only checks: the C-ABI library loads and exports every declared symbol, the
host-side logic (parameters, header parsing, argument errors) behaves like the
reference, the host build of the kernel cores agrees with the oracle, and the
synthetic generators are deterministic.  No GPU compute is called here.
"""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from lzma_amd import _native as N
    lib = N.lib()
    header = (ROOT / "include" / "lzma_amd.h").read_text()
    declared = set(re.findall(r"\b(lzma_amd_[a-z0-9_]+)\s*\(", header))
    assert declared == set(N.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_levels_table_matches_reference():
    """levels[1..9] of lzma_encoder.go:97-108."""
    import lzma_amd
    want = {1: (16, 64), 2: (18, 64), 3: (20, 64), 4: (22, 128), 5: (23, 128), 6: (24, 128),
            7: (25, 256), 8: (26, 256), 9: (27, 256)}
    for lvl, (dl, fb) in want.items():
        p = lzma_amd.params_from_level(lvl)
        assert (p.dict_log, p.fb, p.lc, p.lp, p.pb) == (dl, fb, 3, 0, 2)
    from lzma_amd import _native as N
    for bad in (0, 10, -1):
        with pytest.raises(N.NativeError) as ei:
            lzma_amd.params_from_level(bad)
        assert ei.value.code == N.ERR_ARG
    assert (lzma_amd.BestSpeed, lzma_amd.BestCompression, lzma_amd.DefaultCompression) == (1, 9, 5)


def test_parse_header(data_lzma, vectors):
    from lzma_amd import _native as N
    lib = N.lib()
    lc, lp, pb, ds = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    us = C.c_int64()
    rc = lib.lzma_amd_parse_header(data_lzma, len(data_lzma), lc, lp, pb, ds, us)
    assert (rc, lc.value, lp.value, pb.value, ds.value, us.value) == (0, 3, 0, 2, 1 << 20, -1)
    bad = bytes.fromhex(vectors[-1]["lzma_hex"])
    assert lib.lzma_amd_parse_header(bad, len(bad), lc, lp, pb, ds, us) == N.ERR_HEADER
    assert lib.lzma_amd_parse_header(data_lzma, 12, lc, lp, pb, ds, us) == N.ERR_TRUNCATED


def test_writer_argument_errors_need_no_gpu():
    import io

    import lzma_amd
    w = lzma_amd.NewWriterSizeLevel(io.BytesIO(), -2, 3)
    with pytest.raises(lzma_amd.ArgumentValueError):
        w.write(b"x")
    w = lzma_amd.NewWriterLevel(io.BytesIO(), 0)
    with pytest.raises(lzma_amd.ArgumentValueError):
        w.close()


@pytest.fixture(scope="module")
def hostsim():
    import subprocess
    d = ROOT / "tests" / "hostsim"
    subprocess.check_call(["make", "-s"], cwd=d)
    L = C.CDLL(str(d / "libhostsim.so"))
    L.hs_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.hs_encode.argtypes = [C.c_char_p, C.c_size_t] + [C.c_uint32] * 5 + [C.c_int, C.c_void_p, C.c_size_t,
                                                                         C.POINTER(C.c_size_t)]
    L.hs_encode_coop.argtypes = L.hs_encode_coop3.argtypes = L.hs_encode.argtypes
    return L


def _hs_decode(L, stream, cap=1 << 20):
    out = np.empty(cap, np.uint8)
    n = C.c_size_t(0)
    rc = L.hs_decode(stream, len(stream), out.ctypes.data, cap, C.byref(n))
    return rc, out[: n.value].tobytes()


def _hs_encode(L, raw, dict_log, fb, lc=3, lp=0, pb=2, eos=True, waves=1):
    cap = len(raw) * 2 + 1024
    out = np.empty(cap, np.uint8)
    n = C.c_size_t(0)
    fn = {1: L.hs_encode, 2: L.hs_encode_coop, 3: L.hs_encode_coop3}[waves]
    rc = fn(raw, len(raw), dict_log, fb, lc, lp, pb, int(eos), out.ctypes.data, cap, C.byref(n))
    assert rc == 0
    return out[: n.value].tobytes()


def test_kernel_decode_core_on_host(hostsim, vectors, data_txt, data_lzma):
    """The decoder kernel's serial core, compiled for the host, on the reference vectors."""
    for v in vectors:
        rc, out = _hs_decode(hostsim, bytes.fromhex(v["lzma_hex"]))
        if v["err"]:
            assert rc == 1
        else:
            assert rc == 0 and out == bytes.fromhex(v["raw_hex"])
    rc, out = _hs_decode(hostsim, data_lzma)
    assert rc == 0 and out == data_txt


@pytest.mark.parametrize("cfg", [(20, 64, 3, 0, 2, True), (22, 128, 3, 0, 2, False), (16, 273, 0, 2, 0, True),
                                 (12, 5, 8, 4, 4, True), (14, 32, 4, 1, 3, False)])
def test_kernel_encode_core_on_host(hostsim, data_txt, cfg):
    """The encoder kernel's core, compiled for the host with a brute-force exhaustive
    finder, equals the oracle with the cut value raised, byte for byte: as one wave, and as
    two and three cooperating waves (host threads over the same memory)."""
    from oracle import oracle as O
    dl, fb, lc, lp, pb, eos = cfg
    for raw in (b"", b"a", b"hello world\n", data_txt[:80000]):
        want = O.encode(raw, O.params(dict_log=dl, fb=fb, lc=lc, lp=lp, pb=pb, cut=1 << 30, eos=eos))
        for waves in (1, 2, 3):
            assert _hs_encode(hostsim, raw, dl, fb, lc, lp, pb, eos, waves=waves) == want, waves


def test_exhaustive_cut_stays_within_one_percent(data_txt):
    from oracle import oracle as O
    a = O.encode(data_txt, O.params(3, eos=True, cut=1 << 30))
    b = O.encode(data_txt, O.params(3, eos=True))
    assert abs(len(a) - len(b)) <= 0.01 * len(b)


def test_synth_is_deterministic():
    from tests import synth
    a = synth.markov(3 << 19, threads=1)
    b = synth.markov(3 << 19, threads=8)
    assert np.array_equal(a, b)
    c = synth.markov(1 << 20, first_block=1)
    assert np.array_equal(a[1 << 20:], c[: 1 << 19])
    import hashlib
    assert hashlib.sha256(a.tobytes()).hexdigest() == hashlib.sha256(b.tobytes()).hexdigest()
    s = synth.stream_sizes(10000)
    assert s.min() >= 4096 and s.max() <= 65536 and 15000 < s.mean() < 30000
    lg = synth.logtext(1 << 16, block=1 << 16)
    assert lg[:5].tobytes() == b"2010-"


def test_encoder_core_under_sanitizers():
    """The host build of the encoder core with AddressSanitizer and UndefinedBehaviorSanitizer: 24
    drawn inputs and parameter sets, one / two / three waves per chunk (the cooperative ones on
    real threads), identical streams and no report."""
    import subprocess
    d = ROOT / "tests" / "hostsim"
    subprocess.run(["make", "-s", "sanitize_check"], cwd=d, check=True, capture_output=True, timeout=600)
    import os
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=0")  # (a preloaded library may come first)
    r = subprocess.run([str(d / "sanitize_check"), str(ROOT / "tests" / "golden" / "data.txt")], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count(" same") == 24 and "DIFFERENT" not in r.stdout and "runtime error" not in r.stderr
