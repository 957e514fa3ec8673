"""Command-line front end with the flags and checks of the reference's `lzma_go`
(lzma_go/main.go:20-29 the flags, :55-152 the argument checks, :154-259 the file plumbing):

    python -m lzma_amd.cli [OPTION]... [FILE]

Compresses or uncompresses FILE, by default in place (FILE -> FILE.lzma, the original removed);
with no FILE, or when FILE is -, reads standard input and needs -c.  A file is compressed with its
size in the header (NewWriterSizeLevel, main.go:224), standard input with the end marker
(NewWriterLevel, main.go:212).  All codec work goes through lzma_amd.api, i.e. the GPU library.

`-cores` is accepted and range-checked as in the reference, where it only sets GOMAXPROCS
(main.go:85) and parallelises nothing: one file is one LZMA stream, and one stream is one chain on
one GPU here as well.  One difference from the reference: an output file that a failed run began
is removed again, not left half written.
"""
import argparse
import os
import sys

from . import api

_PROG = "lzma_amd.cli"


def _parser():
    # the reference uses Go's flag package: single-dash names, booleans without a value
    p = argparse.ArgumentParser(prog=_PROG, add_help=False, allow_abbrev=False,
                                usage="%(prog)s [OPTION]... [FILE]",
                                description="Compress or uncompress FILE (by default, compress FILE in-place).",
                                epilog="With no FILE, or when FILE is -, read standard input.")
    p.add_argument("-c", dest="stdout", action="store_true",
                   help="write on standard output, keep original files unchanged")
    p.add_argument("-d", dest="decompress", action="store_true", help="decompress; see also -c and -k")
    p.add_argument("-f", dest="force", action="store_true", help="force overwrite of output file")
    p.add_argument("-h", dest="help", action="store_true", help="print this help message")
    p.add_argument("-k", dest="keep", action="store_true", help="keep original files unchanged")
    p.add_argument("-s", dest="suffix", default=None, help="use provided suffix on compressed files (default lzma)")
    p.add_argument("-l", dest="level", type=int, default=None, help="compression level [1 ... 9] (default 5)")
    p.add_argument("-cores", dest="cores", type=int, default=1,
                   help="number of cores to use for parallelization (checked, 1 ... 32; one stream is one chain)")
    p.add_argument("files", nargs="*")
    return p


class _Exit(Exception):
    def __init__(self, code, msg=""):
        super().__init__(msg)
        self.code, self.msg = code, msg


def _bad_args(p, msg):
    """The reference prints the usage, then `check args: <msg>`, and exits with status 1."""
    raise _Exit(1, p.format_help() + f"\n{_PROG}: check args: {msg}\n")


def _plan(argv):
    """Parses and checks the arguments in the reference's order (main.go:56-152).  Returns
    (args, in_path or None for standard input, out_path or None for standard output)."""
    p = _parser()
    try:
        a = p.parse_args(argv)
    except SystemExit as e:  # argparse's own complaint (unknown flag, bad integer): status 2, as Go's flag
        raise _Exit(2 if e.code else 0)
    if a.help:
        raise _Exit(1, p.format_help())  # (the reference ends with log.Fatal here: status 1)
    suffix_set, level_set = a.suffix is not None, a.level is not None
    suffix = a.suffix if suffix_set else "lzma"
    level = a.level if level_set else 5
    if a.stdout and suffix_set:
        _bad_args(p, "stdout set, suffix not used")
    if a.stdout and a.force:
        _bad_args(p, "stdout set, force not used")
    if a.stdout and a.keep:
        _bad_args(p, "stdout set, keep is redundant")
    if len(a.files) > 1:
        _bad_args(p, "too many file, provide at most one file at a time or check order of flags")
    if a.decompress and level_set:
        _bad_args(p, "compression level is used for compression only")
    if level < 1 or level > 9:
        _bad_args(p, "compression level out of range")
    if a.cores < 1 or a.cores > 32:
        _bad_args(p, "invalid number of cores")
    a.level, a.suffix = level, suffix

    if not a.files or a.files[0] == "-":
        if not a.stdout:
            _bad_args(p, "reading from stdin, can write only to stdout")
        if suffix_set:
            _bad_args(p, "reading from stdin, suffix not needed")
        return a, None, None
    in_path = a.files[0]
    try:
        st = os.lstat(in_path)
    except OSError as e:
        raise _Exit(1, f"{e}\n")
    import stat as _stat
    if _stat.S_ISDIR(st.st_mode):
        _bad_args(p, f"{in_path} is not a regular file")
    if a.stdout:
        return a, in_path, None
    if suffix == "":
        _bad_args(p, "suffix can't be an empty string")
    if a.decompress:
        d, name = os.path.split(in_path)
        ext = "." + suffix
        if not name.endswith(ext):
            _bad_args(p, f"file {in_path} doesn't have suffix .{suffix}")
        if len(name) <= len(ext):
            raise _Exit(1, f"error: can't strip suffix .{suffix} from file {in_path}\n")
        out_path = os.path.join(d, name[: -len(ext)])
    else:
        out_path = in_path + "." + suffix
    if os.path.lexists(out_path):
        if os.path.isdir(out_path) and not os.path.islink(out_path):
            _bad_args(p, f"outFile {out_path} exists and is not a regular file")
        if not a.force:
            _bad_args(p, f"outFile {out_path} exists. use force to overwrite")
        os.remove(out_path)
    return a, in_path, out_path


def _copy(dst, src, block=1 << 20):
    while True:
        b = src.read(block)
        if not b:
            return
        dst.write(b)


def main(argv=None):
    try:
        a, in_path, out_path = _plan(sys.argv[1:] if argv is None else argv)
    except _Exit as e:
        if e.msg:
            sys.stderr.write(e.msg)
        return e.code
    fin = sys.stdin.buffer if in_path is None else open(in_path, "rb")
    fout = sys.stdout.buffer if out_path is None else open(out_path, "wb")
    try:
        if a.decompress:
            z = api.NewReader(fin)
            _copy(fout, z)
            z.close()
        else:
            if in_path is None:
                z = api.NewWriterLevel(fout, a.level)
            else:
                z = api.NewWriterSizeLevel(fout, os.lstat(in_path).st_size, a.level)
            _copy(z, fin)
            z.close()
        fout.flush()
    except Exception as e:  # the reference ends with log.Fatal(err): message, status 1
        sys.stderr.write(f"{_PROG}: {e}\n")
        if out_path is not None:
            fout.close()
            try:
                os.remove(out_path)
            except OSError:
                pass
        return 1
    finally:
        if in_path is not None:
            fin.close()
        if out_path is not None and not fout.closed:
            fout.close()
    if out_path is not None and not a.keep:
        os.remove(in_path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
