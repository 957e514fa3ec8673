"""ctypes loader for the deterministic synthetic-input generators (tools/synth.c)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_SO = ROOT / "tools" / "libsynth.so"
SEED = 0x9E3779B97F4A7C15
_lib = None
_model = None


def lib():
    global _lib
    if _lib is None:
        src = ROOT / "tools" / "synth.c"
        if not _SO.exists() or (src.exists() and _SO.stat().st_mtime < src.stat().st_mtime):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-pthread", "-o", str(_SO), str(src)])
        L = C.CDLL(str(_SO))
        L.synth_markov_build.restype = C.c_void_p
        L.synth_markov_build.argtypes = [C.c_void_p, C.c_size_t]
        L.synth_markov_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int]
        L.synth_log_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.synth_stream_sizes.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _markov_model():
    global _model
    if _model is None:
        sample = np.fromfile(ROOT / "tests" / "golden" / "data.txt", dtype=np.uint8)
        _model = lib().synth_markov_build(sample.ctypes.data, sample.size)
    return _model


def markov(n, cfg=2, first_block=0, threads=8, out=None):
    """n bytes of order-2 Markov text (model: data.txt), 1 MiB blocks first_block.."""
    buf = np.empty(n, dtype=np.uint8) if out is None else out
    lib().synth_markov_fill(_markov_model(), SEED + cfg, first_block, buf.ctypes.data, n, threads)
    return buf


def logtext(n, cfg=4, block=1 << 24, first_block=0, threads=8, out=None):
    """n bytes of redundant dpkg-style log in independent blocks of `block` bytes."""
    buf = np.empty(n, dtype=np.uint8) if out is None else out
    lib().synth_log_fill(SEED + cfg, first_block, block, buf.ctypes.data, n, threads)
    return buf


def stream_sizes(n, lo=4096, hi=65536, cfg=3):
    s = np.empty(n, dtype=np.uint32)
    lib().synth_stream_sizes(SEED + cfg, lo, hi, s.ctypes.data, n)
    return s
