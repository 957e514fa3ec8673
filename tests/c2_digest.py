"""64-bit digests of match-finder output, per block of positions, in numpy (for the oracle's
lists) and in torch (for lists that sit in GPU memory).

digest(block) = H(counts of the block's positions, indexed by position)
              + 3 * H(pairs of those positions as len << 32 | dist-1, indexed by the pair's
                      index in the stream-wide pair array)            (mod 2^64)
H(x, i)       = sum over elements of t * t + t,  t = (x + K0) * ((i + 1) * K1)   (mod 2^64)

Every term depends on the value and on where it stands, so an omitted, added, moved or altered
pair changes the block's digest (and, through the pair index, those of the blocks after it).
Only additions and multiplications modulo 2^64 are used: numpy's uint64 and torch's int64
wrap the same way.
"""
import numpy as np

K0 = 0x2545F4914F6CDD1D
K1 = 0x9E3779B97F4A7C15
MASK = (1 << 64) - 1


def _terms_np(x, first_index):
    x = x.astype(np.uint64)
    i = np.arange(first_index + 1, first_index + 1 + x.size, dtype=np.uint64)
    with np.errstate(over="ignore"):
        t = (x + np.uint64(K0)) * (i * np.uint64(K1))
        return t * t + t


def digests_np(counts, pairs, block):
    """counts: uint32[n]; pairs: uint32[total, 2] in position order -> list of ints, one per block."""
    n = counts.size
    tc = _terms_np(counts, 0)
    packed = (pairs[:, 0].astype(np.uint64) << np.uint64(32)) | pairs[:, 1].astype(np.uint64)
    tp = _terms_np(packed, 0)
    ends = np.cumsum(counts.astype(np.int64))
    out = []
    with np.errstate(over="ignore"):
        for b in range(0, n, block):
            e = min(n, b + block)
            p0 = int(ends[b - 1]) if b else 0
            p1 = int(ends[e - 1])
            hc = int(tc[b:e].sum(dtype=np.uint64))
            hp = int(tp[p0:p1].sum(dtype=np.uint64))
            out.append((hc + 3 * hp) & MASK)
    return out


def digests_torch(counts, offs, pairs, block):
    """The same on device tensors: counts int32[n], offs int64[n] (exclusive prefix sum of the
    counts), pairs int32[>= 2 * total] as (len, dist-1) words.  Works block by block, so the
    temporaries stay small."""
    import torch

    def s64(v):  # constant as a signed 64-bit value
        return v - (1 << 64) if v >= 1 << 63 else v

    n = counts.numel()
    dev = counts.device
    k0, k1 = s64(K0), s64(K1)
    out = []
    for b in range(0, n, block):
        e = min(n, b + block)
        c = counts[b:e].to(torch.int64)
        i = torch.arange(b + 1, e + 1, dtype=torch.int64, device=dev)
        t = (c + k0) * (i * k1)
        hc = int((t * t + t).sum().item())
        p0 = int(offs[b].item())
        p1 = int(offs[e - 1].item()) + int(counts[e - 1].item())
        w = pairs[2 * p0: 2 * p1].view(-1, 2).to(torch.int64)
        packed = ((w[:, 0] & 0xFFFFFFFF) << 32) | (w[:, 1] & 0xFFFFFFFF)
        j = torch.arange(p0 + 1, p1 + 1, dtype=torch.int64, device=dev)
        t = (packed + k0) * (j * k1)
        hp = int((t * t + t).sum().item())
        out.append((hc + 3 * hp) & MASK)
    return out
