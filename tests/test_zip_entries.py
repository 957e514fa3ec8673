"""Zip archives with LZMA (method 14) entries, the reference's stated use (README.md:7-8):
all entries of an archive decoded in one GPU batch must equal what zipfile extracts."""
import io
import zipfile

import numpy as np
import pytest

from tests import synth

@pytest.mark.gpu
def test_zip_lzma_entries(data_txt):
    from lzma_amd import zipentries
    rng = np.random.default_rng(3)
    text = synth.markov(1 << 20).tobytes()
    files = {"empty.txt": b"", "data.txt": data_txt, "one.bin": b"\x00"}
    for i in range(150):
        a = int(rng.integers(0, len(text) - 70000))
        files[f"dir/f{i:03d}.txt"] = text[a:a + int(rng.integers(1, 65536))]
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w", compression=zipfile.ZIP_LZMA) as zf:
        for name, data in files.items():
            zf.writestr(name, data)
        zf.writestr("stored.txt", b"not lzma", compress_type=zipfile.ZIP_STORED)
    buf.seek(0)
    with zipfile.ZipFile(buf) as zf:
        got = zipentries.read_lzma_entries(zf)
        assert set(got) == set(files)
        for name, data in files.items():
            assert got[name] == data == zf.read(name), name
        with pytest.raises(zipentries.ZipEntryError):
            zipentries.read_lzma_entries(zf, names=["stored.txt"])


def test_entry_stream_header_cpu_side():
    from lzma_amd import zipentries
    raw = bytes([9, 20, 5, 0, 0x5D, 0, 0, 16, 0]) + b"payload"
    s = zipentries.entry_stream(raw, 1234, has_end_marker=False)
    assert s[:5] == bytes([0x5D, 0, 0, 16, 0]) and int.from_bytes(s[5:13], "little") == 1234 and s[13:] == b"payload"
    assert zipentries.entry_stream(raw, 1234, True)[5:13] == b"\xff" * 8
    with pytest.raises(zipentries.ZipEntryError):
        zipentries.entry_stream(raw[:2] + b"\x04\x00" + raw[4:], 1, True)
