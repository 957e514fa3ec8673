"""Two-rank CPU rehearsal (gloo) of the multi-GPU path: shard chunks by rank, encode the
shard, all-gather the compressed sizes, derive container offsets, write the shard's
streams at those offsets.  The container must equal the serial concatenation.

The GPU encoder is not available on CPU, so each rank's shard is encoded by the CPU
oracle here; what is under test is the sharding / size exchange / offset logic of
lzma_amd.parallel, which is exactly what bench.py runs over RCCL.
"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lzma_amd import parallel
from oracle import oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _chunks(n_chunks):
    rng = np.random.default_rng(7)
    words = [b"alpha ", b"beta ", b"gamma\n", b"delta ", b"epsilon "]
    out = []
    for i in range(n_chunks):
        k = int(rng.integers(0, 400)) if i != 2 else 0  # chunk 2 is empty
        out.append(b"".join(words[int(j)] for j in rng.integers(0, len(words), k)))
    return out


def _worker(rank, world, port, n_chunks, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chunks = _chunks(n_chunks)
        b, e = parallel.shard_range(n_chunks, rank, world)
        p = O.params(3, eos=True)
        mine = [O.encode(c, p) for c in chunks[b:e]]
        local = torch.tensor([len(s) for s in mine], dtype=torch.int64)
        allsz = parallel.gather_sizes(local, n_chunks, rank, world)
        off = parallel.container_offsets(allsz)
        assert allsz.numel() == n_chunks
        if rank == 0:
            with open(path, "wb") as f:
                f.truncate(int(off[-1]))
        dist.barrier()
        with open(path, "r+b") as f:
            for i, s in enumerate(mine):
                f.seek(int(off[b + i]))
                f.write(s)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_chunks", [8, 7])  # even and ragged shards
def test_two_rank_container(n_chunks):
    O.lib()
    world = 2
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "container.bin")
        mp.spawn(_worker, args=(world, _free_port(), n_chunks, path), nprocs=world, join=True)
        got = open(path, "rb").read()
    p = O.params(3, eos=True)
    want = b"".join(O.encode(c, p) for c in _chunks(n_chunks))
    assert got == want


def test_shard_range_covers_everything():
    for n in (0, 1, 5, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(4, 2, 2)


def test_offsets_single_rank():
    s = torch.tensor([3, 0, 5], dtype=torch.int64)
    assert parallel.gather_sizes(s, 3, 0, 1).tolist() == [3, 0, 5]
    assert parallel.container_offsets(s).tolist() == [0, 3, 3, 8]
