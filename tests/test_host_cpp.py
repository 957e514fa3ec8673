"""The C++ mirror of the reference API (include/lzma_amd.hpp), driven through a small
test program that follows the reference's own API tests (lzma_encoder_test.go:30-91,
lzma_decoder_test.go:14-32)."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "tests" / "host_cpp" / "host_test"


def _run(*args):
    assert EXE.exists(), f"{EXE} is not built (python -c 'import __graft_entry__ as g; g.build()')"
    return subprocess.run([str(EXE), *map(str, args)], capture_output=True, text=True, timeout=300)


def test_argument_errors_without_gpu():
    r = _run("args")
    assert r.returncode == 0, r.stderr
    assert "args ok" in r.stdout


@pytest.mark.gpu
def test_round_trip_through_cpp_api():
    g = ROOT / "tests" / "golden"
    r = _run("roundtrip", g / "data.txt", g / "data.eos.l3.lzma")
    assert r.returncode == 0, r.stderr
    assert "roundtrip ok" in r.stdout
