#!/usr/bin/env python3
"""Extract the reference's known-answer vectors into a data-only JSON fixture.

Reads the test table `lzmaTests` of the reference (lzma_data_test.go:21-289) as
TEXT and writes tests/golden/vectors.json holding, for each vector, its
description, level, whether the size is known, the raw bytes (hex), the
encoded bytes (hex) and the expected error name.  Only data is kept.

Run once in the build container (the reference is not present on the GPU box):
    python3 tests/golden/make_vectors.py /root/reference/lzma_data_test.go
"""
import json
import re
import sys
from pathlib import Path


def go_string(lit: str) -> bytes:
    """Decode a Go interpreted string literal body (the part between quotes)."""
    out = bytearray()
    i = 0
    raw = lit.encode("utf-8")
    while i < len(raw):
        c = raw[i]
        if c == 0x5C:  # backslash
            n = chr(raw[i + 1])
            out += {"n": b"\n", '"': b'"', "\\": b"\\", "t": b"\t"}[n]
            i += 2
        else:
            out.append(c)
            i += 1
    return bytes(out)


def main(src: str) -> None:
    text = Path(src).read_text(encoding="utf-8")
    body = text[text.index("var lzmaTests") : text.index("type lzmaBenchmark")]
    entries = body.split("lzmaTest{")[2:]  # [0] is the slice header, [1] empty prefix
    vectors = []
    for e in entries:
        strings = re.findall(r'"((?:[^"\\]|\\.)*)"', e)
        descr = strings[0]
        head = e[: e.index("[]byte{")]
        level = int(re.search(r"\n\s*(\d+),\n", head).group(1))
        size_known = re.search(r"\n\s*(true|false),\n", head).group(1) == "true"
        raw = b"".join(go_string(s) for s in strings[1:])
        blob = e[e.index("[]byte{") + 7 :]
        blob = blob[: blob.index("}")]
        enc = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", blob))
        tail = e[e.index("[]byte{") :]
        err = "headerError" if "headerError" in tail else None
        vectors.append(
            dict(descr=descr, level=level, size_known=size_known, raw_hex=raw.hex(),
                 lzma_hex=enc.hex(), err=err)
        )
    out = Path(__file__).with_name("vectors.json")
    out.write_text(json.dumps(vectors, indent=1) + "\n")
    for v in vectors:
        print(v["descr"], len(v["raw_hex"]) // 2, len(v["lzma_hex"]) // 2, v["err"])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/lzma_data_test.go")
