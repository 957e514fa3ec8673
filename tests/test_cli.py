"""The command-line front end (lzma_amd/cli.py) against the checks and the file handling of the
reference's lzma_go (lzma_go/main.go:55-259).  Argument checks run without a GPU; the round trips
go through the GPU library and are checked with liblzma."""
import lzma as pylzma
import os
import subprocess
import sys

import pytest

from lzma_amd import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, capsys):
    code = cli.main(args)
    return code, capsys.readouterr().err


@pytest.mark.parametrize("args,msg", [
    (["-c", "-s", "xz", "f"], "stdout set, suffix not used"),
    (["-c", "-f", "f"], "stdout set, force not used"),
    (["-c", "-k", "f"], "stdout set, keep is redundant"),
    (["a", "b"], "too many file"),
    (["-d", "-l", "3", "f"], "compression level is used for compression only"),
    (["-l", "0", "f"], "compression level out of range"),
    (["-l", "10", "f"], "compression level out of range"),
    (["-cores", "0", "f"], "invalid number of cores"),
    (["-cores", "33", "f"], "invalid number of cores"),
    ([], "reading from stdin, can write only to stdout"),
    (["-"], "reading from stdin, can write only to stdout"),
])
def test_argument_checks(args, msg, capsys):
    code, err = _run(args, capsys)
    assert code == 1 and f"check args: {msg}" in err and "Usage" in err.replace("usage", "Usage")


def test_help_and_unknown_flag(capsys):
    code, err = _run(["-h"], capsys)
    assert code == 1 and "-cores" in err and "read standard input" in err
    code, _ = _run(["-z"], capsys)
    assert code == 2


def test_file_checks(tmp_path, capsys):
    code, err = _run([str(tmp_path / "missing")], capsys)
    assert code == 1 and "No such file" in err
    code, err = _run([str(tmp_path)], capsys)
    assert code == 1 and "is not a regular file" in err
    f = tmp_path / "a.txt"
    f.write_bytes(b"x")
    code, err = _run(["-s", "", str(f)], capsys)
    assert code == 1 and "suffix can't be an empty string" in err
    code, err = _run(["-d", str(f)], capsys)
    assert code == 1 and "doesn't have suffix .lzma" in err
    g = tmp_path / ".lzma"
    g.write_bytes(b"x")
    code, err = _run(["-d", str(g)], capsys)
    assert code == 1 and "can't strip suffix" in err
    (tmp_path / "a.txt.lzma").write_bytes(b"old")
    code, err = _run([str(f)], capsys)
    assert code == 1 and "exists. use force to overwrite" in err
    assert (tmp_path / "a.txt.lzma").read_bytes() == b"old" and f.exists()
    (tmp_path / "a.txt.xz").mkdir()
    code, err = _run(["-s", "xz", str(f)], capsys)
    assert code == 1 and "exists and is not a regular file" in err


def _cli(args, **kw):
    return subprocess.run([sys.executable, "-m", "lzma_amd.cli"] + args, cwd=ROOT, capture_output=True, timeout=240, **kw)


@pytest.mark.gpu
def test_round_trips_in_place_and_through_pipes(tmp_path, data_txt):
    f = tmp_path / "data.txt"
    f.write_bytes(data_txt)
    # in place, level 3: FILE -> FILE.lzma with the size in the header, the original removed
    r = _cli(["-l", "3", str(f)])
    assert r.returncode == 0, r.stderr
    z = tmp_path / "data.txt.lzma"
    assert not f.exists() and z.exists()
    enc = z.read_bytes()
    assert int.from_bytes(enc[5:13], "little") == len(data_txt)
    assert pylzma.decompress(enc, format=pylzma.FORMAT_ALONE) == data_txt
    # -k keeps the input, -f replaces an output that exists
    r = _cli(["-d", "-k", str(z)])
    assert r.returncode == 0 and z.exists() and f.read_bytes() == data_txt
    r = _cli(["-d", "-k", str(z)])
    assert r.returncode == 1 and b"use force to overwrite" in r.stderr
    r = _cli(["-d", "-f", str(z)])
    assert r.returncode == 0 and not z.exists() and f.read_bytes() == data_txt
    # standard input to standard output: end marker instead of a size
    r = _cli(["-c"], input=data_txt[:50000])
    assert r.returncode == 0 and r.stdout[5:13] == b"\xff" * 8
    assert pylzma.decompress(r.stdout, format=pylzma.FORMAT_ALONE) == data_txt[:50000]
    r2 = _cli(["-d", "-c", "-"], input=r.stdout)
    assert r2.returncode == 0 and r2.stdout == data_txt[:50000]
    # -c with a file leaves it alone; another suffix
    r = _cli(["-c", str(f)])
    assert r.returncode == 0 and f.exists() and pylzma.decompress(r.stdout, format=pylzma.FORMAT_ALONE) == data_txt
    r = _cli(["-s", "lz", "-k", "-l", "1", str(f)])
    assert r.returncode == 0 and (tmp_path / "data.txt.lz").exists() and f.exists()
    # a broken stream: message, status 1, no output file left
    bad = tmp_path / "bad.lzma"
    bad.write_bytes(b"\xff" + enc[1:])
    r = _cli(["-d", str(bad)])
    assert r.returncode == 1 and b"error in lzma header" in r.stderr and not (tmp_path / "bad").exists() and bad.exists()
