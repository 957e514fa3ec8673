"""GPU match finder parity: lzma_amd_bt4_matches against the oracle's BT4 run with the
cut value raised so the tree walk is exhaustive (lz_bin_tree.go:115-253).  The
reference has no direct getMatches test; its lists are pinned transitively by the
byte-exact encoder vectors the oracle passes.  Bit-exact is the bar.
"""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
CUT_INF = 1 << 30


def _check(data, dict_size, fb):
    import lzma_amd
    from oracle import oracle as O
    data = bytes(data)
    c_ref, p_ref = O.bt4_matches(data, dict_size, fb, CUT_INF)
    c, p = lzma_amd.bt4_matches(data, dict_size, fb)
    assert np.array_equal(c, c_ref), np.flatnonzero(c != c_ref)[:10]
    assert np.array_equal(p, p_ref)
    return int(c.sum())


def test_data_txt_level3(data_txt):
    assert _check(data_txt, 1 << 20, 64) > 400000


@pytest.mark.parametrize("dict_size,fb", [(4096, 64), (1 << 16, 273), (1 << 12, 5), (1 << 22, 128),
                                          (1 << 26, 7), (1 << 27, 8)])
def test_data_txt_params(data_txt, dict_size, fb):
    _check(data_txt[:120000], dict_size, fb)


def test_markov_sliding_window():
    _check(synth.markov(3 << 19), 1 << 18, 64)


def test_log_long_matches():
    _check(synth.logtext(1 << 20, block=1 << 20), 1 << 20, 273)


def test_random_bytes_and_runs():
    rng = np.random.default_rng(11)
    parts = [rng.integers(0, 256, 50000, dtype=np.uint8), np.zeros(70000, np.uint8),
             rng.integers(0, 4, 60000, dtype=np.uint8), np.tile(np.arange(7, dtype=np.uint8), 9000)]
    _check(np.concatenate(parts), 1 << 16, 64)


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 8, 9, 63, 64, 65, 300])
def test_tiny(n, data_txt):
    _check(data_txt[:n], 1 << 20, 64)
    _check(b"a" * n, 1 << 12, 32)


def _mutated_repeats(n, block, seed):
    """Blocks of `block` random bytes repeated with a few bytes changed per copy: common
    prefixes of every length up to the block size, groups of thousands of positions."""
    rng = np.random.default_rng(seed)
    base = rng.integers(97, 123, block, dtype=np.uint8)
    out = np.tile(base, n // block + 1)[:n].copy()
    hits = rng.integers(0, n, n // 97)
    out[hits] = rng.integers(97, 123, hits.size, dtype=np.uint8)
    return out


@pytest.mark.parametrize("block,fb", [(400, 273), (150, 273), (70, 256), (33, 273), (400, 255)])
def test_deep_groups(block, fb):
    """Searches of 256 bytes and more use the 32-, 64- and 128-byte groups (fb 255: they are
    not built); the data has matches on both sides of each of those lengths."""
    _check(_mutated_repeats(300000, block, block), 1 << 20, fb)


@pytest.mark.parametrize("n", [31, 32, 63, 64, 127, 128, 255, 256, 511, 512, 513, 1100])
def test_deep_levels_short_texts(n):
    """Texts around the lengths at which a deep level starts to be built."""
    _check(_mutated_repeats(n, 37, n), 1 << 16, 273)
    _check(b"ab" * (n // 2), 1 << 16, 273)


def test_runs_at_full_length():
    _check(np.concatenate([np.zeros(5000, np.uint8), _mutated_repeats(20000, 300, 5), np.zeros(3000, np.uint8)]),
           1 << 12, 273)
