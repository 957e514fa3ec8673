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


def _decode_worker(rank, world, port, q):
    """The decode workload's exchange: streams sharded by raw bytes (shares of different length),
    every rank decodes its own (oracle here), the sizes of all streams go round in one all-gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        raws = _chunks(13)
        p = O.params(3, eos=True)
        sizes = np.array([len(r) for r in raws])
        spans = [parallel.shard_by_bytes(sizes, r, world) for r in range(world)]
        b, e = spans[rank]
        mine = [O.decode(O.encode(r, p))[1] for r in raws[b:e]]
        local = torch.tensor([len(m) for m in mine], dtype=torch.int64)
        allsz = parallel.gather_ragged(local, [y - x for x, y in spans], rank, world)
        q.put((rank, allsz.tolist(), parallel.container_offsets(allsz).tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_decode_sizes():
    O.lib()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_decode_worker, args=(2, _free_port(), q), nprocs=2, join=True)
    got = sorted(q.get() for _ in range(2))
    want = [len(r) for r in _chunks(13)]
    for rank, sizes, off in got:
        assert sizes == want and off[-1] == sum(want) and len(off) == 14
    with pytest.raises(ValueError):
        parallel.gather_ragged(torch.zeros(3, dtype=torch.int64), [3], 0, 2)


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


def test_shard_by_bytes_balances_and_covers():
    rng = np.random.default_rng(1)
    for n in (0, 1, 7, 5000):
        sizes = rng.integers(4096, 65537, n)
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_by_bytes(sizes, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            if n == 5000:  # no shard is further than one stream from its fair share of the bytes
                share = sizes.sum() / world
                for b, e in spans:
                    assert abs(int(sizes[b:e].sum()) - share) <= 65536
    # count-balanced sharding of the same sizes would be off by far more than one stream
    sizes = np.concatenate([np.full(100, 65536), np.full(100, 4096)])
    b, e = parallel.shard_by_bytes(sizes, 0, 2)
    assert e < 100 and abs(int(sizes[b:e].sum()) - int(sizes.sum()) // 2) <= 65536
    assert parallel.shard_by_bytes([0, 0, 0], 1, 2) == parallel.shard_range(3, 1, 2)
    with pytest.raises(ValueError):
        parallel.shard_by_bytes([1, 2], 2, 2)


def _bytes_worker(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chunks = _chunks(9)
        sizes = [len(c) for c in chunks]
        b, e = parallel.shard_by_bytes(sizes, rank, world)
        p = O.params(3, eos=True)
        mine = [O.encode(c, p) for c in chunks[b:e]]
        # ragged shards of unknown width on the other ranks: gather counts first
        cnt = torch.tensor([e - b], dtype=torch.int64)
        allcnt = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allcnt, cnt)
        counts = [int(c) for c in allcnt]
        width = max(counts)
        padded = torch.zeros(width, dtype=torch.int64)
        padded[: e - b] = torch.tensor([len(s) for s in mine], dtype=torch.int64)
        out = torch.empty(world * width, dtype=torch.int64)
        dist.all_gather_into_tensor(out, padded)
        allsz = torch.cat([out[r * width: r * width + counts[r]] for r in range(world)])
        off = parallel.container_offsets(allsz)
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


def test_two_rank_container_sharded_by_bytes():
    O.lib()
    world = 2
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "container.bin")
        mp.spawn(_bytes_worker, args=(world, _free_port(), path), nprocs=world, join=True)
        got = open(path, "rb").read()
    p = O.params(3, eos=True)
    assert got == b"".join(O.encode(c, p) for c in _chunks(9))


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` with no launcher environment starts two ranks itself (here over
    gloo with the stub workload: no GPU, no codec) and rank 0 reports n_gpus 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "stub",
                        "--backend", "gloo", "--steps", "3", "--warmup", "1", "--chunks", "11"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "strong"
    assert line["config"]["chunks_total"] == 11 and line["cpu_baseline"] is None
    # the same through the launcher the driver uses
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "stub", "--backend", "gloo",
                        "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_bench_stops_when_a_rank_dies():
    """A rank that raises leaves the other one waiting in the size exchange: bench.py must end
    within seconds with a non-zero code and the dead rank's last lines, not hang."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_STUB_FAIL_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "stub",
                        "--backend", "gloo", "--steps", "3", "--warmup", "1", "--chunks", "11"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert time.time() - t0 < 60
    assert "rank 1 exited with code" in r.stderr and "fails on purpose" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
