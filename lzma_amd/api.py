"""Host-side mirror of the reference package's public API for the hot path.

Same names, argument meaning and error behaviour as the Go package:
NewReader (lzma_decoder.go:351-363), NewWriterSizeLevel / NewWriterLevel /
NewWriterSize / NewWriter (lzma_encoder.go:1109-1150), BestSpeed /
BestCompression / DefaultCompression (lzma_encoder.go:14-18), headerError /
streamError (lzma_decoder.go:91-95), argumentValueError (lzma_encoder.go:26-34).

The reference streams through a goroutine and a pipe; the bytes it produces only
depend on the whole input, so the writer here buffers what is written and runs
the GPU encoder on Close(), and the reader decodes on first Read().  All codec
work goes through the C ABI (liblzma_amd.so); nothing here computes LZMA on the
CPU.
"""
import io

from . import _native as N
from . import codec

BestSpeed = 1
BestCompression = 9
DefaultCompression = 5


class HeaderError(ValueError):
    """error in lzma header (headerError, lzma_decoder.go:95)"""


class StreamError(ValueError):
    """error in lzma encoded data stream (streamError, lzma_decoder.go:92)"""


class ArgumentValueError(ValueError):
    """illegal argument value error (argumentValueError, lzma_encoder.go:27-34)"""

    def __init__(self, msg, val):
        super().__init__(f"illegal argument value error: {msg} with value {val}")
        self.msg, self.val = msg, val


class _Reader(io.RawIOBase):
    def __init__(self, r):
        self._r = r
        self._buf = None
        self._pos = 0

    def readable(self):
        return True

    def _fill(self):
        raw = self._r.read()
        status, outs = codec.decode_batch([raw])
        st = int(status[0])
        if st == N.ERR_HEADER:
            raise HeaderError("error in lzma header")
        if st == N.ERR_STREAM:
            raise StreamError("error in lzma encoded data stream")
        if st == N.ERR_TRUNCATED:
            # the reference surfaces a plain io.EOF here, which io.Copy treats as a clean end
            if len(raw) < 13:
                raise EOFError("unexpected EOF")
        elif st != N.OK:
            raise N.NativeError(st, "decode failed")
        self._buf = outs[0]

    def readinto(self, b):
        if self._buf is None:
            self._fill()
        n = min(len(b), len(self._buf) - self._pos)
        b[:n] = self._buf[self._pos:self._pos + n]
        self._pos += n
        return n


def NewReader(r):
    """Returns a reader of the uncompressed version of r (a binary file object)."""
    return _Reader(r)


class _Writer(io.RawIOBase):
    def __init__(self, w, size, level):
        self._w, self._size, self._level = w, size, level
        self._chunks = []
        self._err = None
        if level < 1 or level > 9:
            self._err = ArgumentValueError("level out of range", level)
        elif size < -1:
            self._err = ArgumentValueError("illegal size", size)

    def writable(self):
        return True

    def write(self, b):
        # the reference reports encoder errors on a later Write or on Close
        if self._err is not None:
            raise self._err
        self._chunks.append(bytes(b))
        return len(b)

    def close(self):
        if self.closed:
            return
        super().close()
        if self._err is not None:
            raise self._err
        data = b"".join(self._chunks)
        p = codec.params_from_level(self._level, eos=(self._size == -1))
        status, outs = codec.encode_chunks([data], p)
        if int(status[0]) != N.OK:
            raise N.NativeError(int(status[0]), "encode failed")
        out = bytearray(outs[0])
        if self._size != -1:
            # the reference writes the caller's size verbatim and never checks it
            out[5:13] = int(self._size).to_bytes(8, "little", signed=True)
        self._w.write(bytes(out))


def NewWriterSizeLevel(w, size, level):
    """Writes to w the compressed version of data written to the returned writer;
    size is the uncompressed size or -1 if unknown (end marker is then written)."""
    return _Writer(w, size, level)


def NewWriterLevel(w, level):
    return NewWriterSizeLevel(w, -1, level)


def NewWriterSize(w, size):
    return NewWriterSizeLevel(w, size, DefaultCompression)


def NewWriter(w):
    return NewWriterSizeLevel(w, -1, DefaultCompression)
