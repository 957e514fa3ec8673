"""Batch decoding of the LZMA-compressed entries of a zip archive.

The reference's README names this as the package's purpose: decoding zip entries
compressed with method 14 (README.md:7-8; its reader therefore never reads ahead,
range_coder.go:33-38).  An entry's data is, per the ZIP application note (section 5.8,
not anything in the reference), a 4-byte prefix (LZMA SDK version, size of the
properties = 5), the 5 property bytes of an .lzma header, and the raw LZMA stream;
the uncompressed size lives in the zip directory, and general-purpose bit 1 says
whether the stream ends with an end marker.  Prepending the missing 8-byte size field
turns every entry into a complete .lzma stream, and all entries of an archive go to
the GPU in one lzma_amd_decode_batch call.
"""
import struct
import zipfile
import zlib

from . import _native as N
from . import codec

METHOD_LZMA = 14


class ZipEntryError(ValueError):
    pass


def entry_stream(raw: bytes, uncompressed_size: int, has_end_marker: bool) -> bytes:
    """The .lzma stream (13-byte header + payload) equivalent to one method-14 entry."""
    if len(raw) < 9:
        raise ZipEntryError("entry shorter than its LZMA prefix")
    _major, _minor, props_size = struct.unpack_from("<BBH", raw, 0)
    if props_size != 5:
        raise ZipEntryError(f"unexpected LZMA properties size {props_size}")
    size = -1 if has_end_marker else uncompressed_size
    return raw[4:9] + struct.pack("<q", size) + raw[9:]


def read_lzma_entries(zf: zipfile.ZipFile, names=None, check_crc: bool = True):
    """Decodes the method-14 entries of an open archive on the GPU.

    Returns {name: bytes}.  Entries stored with other methods are skipped unless named
    explicitly (then it is an error).
    """
    infos = [zf.getinfo(n) for n in names] if names is not None else \
        [i for i in zf.infolist() if i.compress_type == METHOD_LZMA]
    streams, caps = [], []
    fp = zf.fp
    for info in infos:
        if info.compress_type != METHOD_LZMA:
            raise ZipEntryError(f"{info.filename}: method {info.compress_type}, not LZMA")
        fp.seek(info.header_offset)
        hdr = fp.read(30)
        if hdr[:4] != b"PK\x03\x04":
            raise ZipEntryError(f"{info.filename}: bad local header")
        name_len, extra_len = struct.unpack_from("<HH", hdr, 26)
        fp.seek(info.header_offset + 30 + name_len + extra_len)
        raw = fp.read(info.compress_size)
        streams.append(entry_stream(raw, info.file_size, bool(info.flag_bits & 2)))
        # the directory's size is untrusted too: never more room than the entry can expand to
        caps.append(min(info.file_size, codec.max_output(len(raw) + 4)))
    status, outs = codec.decode_batch(streams, out_caps=caps)
    res = {}
    for info, st, data in zip(infos, status, outs):
        if int(st) != N.OK or len(data) != info.file_size:
            raise ZipEntryError(f"{info.filename}: decode status {N.STATUS_NAMES.get(int(st), st)}, "
                                f"{len(data)} of {info.file_size} bytes")
        if check_crc and (zlib.crc32(data) & 0xFFFFFFFF) != info.CRC:
            raise ZipEntryError(f"{info.filename}: CRC mismatch")
        res[info.filename] = data
    return res
