#!/usr/bin/env python3
"""Headline benchmark: chunked LZMA encode on MI355X, input MB/s.

Workload (default, BASELINE.json configs[2]): 4096 independent 1 MiB chunks (4 GiB in
all) of synthetic order-2 Markov text, encoder parameters {dict 4 MiB, fb 128, lc3 lp0
pb2, size-known header}.  One process per GPU; the 4096 chunks are sharded over the
ranks (strong scaling: the total stays 4 GiB whatever N is) with no data-path
collective; the only exchange is an all-gather of the per-chunk compressed sizes
(RCCL) from which every rank derives the container offsets.  A "step" is one pass of
lzma_amd_encode_chunks_dev over the rank's chunks with the input already resident in
HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--workload encode|decode|mf|c5] [--chunks C] [--no-cpu-baseline]

`--gpus N` with N > 1 and no launcher environment starts the N ranks itself (one child
process per GPU, before anything touches a GPU); under `python -m torch.distributed.run`
the ranks are already there and RANK / LOCAL_RANK / WORLD_SIZE are taken from the
environment.

Prints ONE JSON line (rank 0).  Also reports:
  roofline     -- algorithmic bytes (N + C) of the dominant kernel / its HIP-event time
  cpu_baseline -- the CPU oracle (a C restatement of the reference, default cut value)
                  on a bounded sample of the same chunks, on this host's cores
  extra        -- (single-GPU encode run only) two steps each of the batched-decode and
                  the match-finder workloads, so that they are measured in the same run
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from lzma_amd import parallel  # noqa: E402

HBM_PEAK_GBS = 8000.0
LDS_PEAK_GBS = 256 * 256 * 2.4  # 256 CUs x 256 B/clk x 2.4 GHz
MIB = 1 << 20
T_START = time.perf_counter()


def native():
    from lzma_amd import _native as N
    return N


def pmc_summary(name):
    """A committed rocprofv3 PMC summary (profiles/), newest round first; None if absent."""
    for rnd in ("r03", "r02", "r01"):
        path = ROOT / "profiles" / f"{rnd}_{name}.json"
        try:
            return json.loads(path.read_text())
        except (OSError, ValueError):
            continue
    return None


def pmc_traffic(kernel, input_bytes_per_launch):
    """HBM bytes per launch of `kernel`, from the committed PMC summary of this workload
    (FETCH_SIZE + WRITE_SIZE per input byte, measured in separate counter runs), scaled to
    this run's launch size."""
    s = pmc_summary("pmc_encode_832x1MiB") or {}
    try:
        return int(s["kernels"][kernel]["hbm_bytes_per_input_byte_reported"] * input_bytes_per_launch)
    except (KeyError, TypeError):
        return None


def pmc_lds(kernel, input_bytes_per_launch, kernel_ms):
    """LDS bytes per second of `kernel` from the same summaries: SQ_LDS_IDX_ACTIVE counts
    LDS-array cycles, each of which moves at most 128 bytes for the 4-byte accesses these
    kernels make; reported as an upper bound next to the LDS peak."""
    s = pmc_summary("pmc_encode_832x1MiB") or {}
    try:
        k = s["kernels"][kernel]
        cyc_per_byte = k["SQ_LDS_IDX_ACTIVE"] / s["input_bytes"]
        gbs = cyc_per_byte * input_bytes_per_launch * 128 / (kernel_ms * 1e-3) / 1e9
        return {"achieved_upper_bound": round(gbs, 1), "peak": LDS_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / LDS_PEAK_GBS, 6)}
    except (KeyError, TypeError, ZeroDivisionError):
        return None


def pmc_traffic_mf(input_bytes_per_launch):
    s = pmc_summary("pmc_mf_256MiB") or {}
    try:
        return int(s["kernels"]["mf_scan_kernel"]["hbm_bytes_per_input_byte_reported"] * input_bytes_per_launch)
    except (KeyError, TypeError):
        return None


def pmc_traffic_decode(output_bytes_per_launch):
    s = pmc_summary("pmc_decode_125k_streams") or {}
    try:
        k = s["decode_kernel_per_output_byte"]
        return int(k["hbm_bytes_per_output_byte_reported"] * output_bytes_per_launch)
    except (KeyError, TypeError):
        return None


def usable_cores():
    """Cores this process may really use: the scheduler affinity, cut by a cgroup quota if
    there is one and by the pool-size hint the GPU boxes export (OMP_NUM_THREADS)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    hint = os.environ.get("OMP_NUM_THREADS", "")
    if hint.isdigit() and int(hint) > 0:
        n = min(n, int(hint))
    return max(1, n)


# ------------------------------------------------------------------ ranks
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """Start n copies of this script, one per GPU, and relay rank 0's output.  Runs before
    anything in this process touches a GPU (no process that has initialised the GPU is ever
    replaced or forked here)."""
    import collections
    import threading
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs, tails, pumps = [], [], []

    def pump(pipe, tail):  # a rank's stderr goes through as it comes; its last lines are kept
        for ln in pipe:
            tail.append(ln)
            sys.stderr.write(ln)
            sys.stderr.flush()

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        p = subprocess.Popen([sys.executable, "-u", str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                             stderr=subprocess.PIPE, text=True, errors="replace")
        tail = collections.deque(maxlen=25)
        t = threading.Thread(target=pump, args=(p.stderr, tail), daemon=True)
        t.start()
        procs.append(p), tails.append(tail), pumps.append(t)
    # A rank that dies leaves the others waiting in their next collective: the first non-zero
    # exit ends the run with that rank's code instead of a time-out.  (Children are only
    # started and stopped here; no process that has touched a GPU is ever replaced.)
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
                break
        else:
            time.sleep(0.2)
    if failed is None:
        failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is None:
        for t in pumps:
            t.join(timeout=5)
        return 0
    rc = procs[failed].returncode
    for p in procs:
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    for t in pumps:
        t.join(timeout=5)
    sys.stderr.write(f"[bench] rank {failed} exited with code {rc}; the other ranks were stopped.  "
                     f"Its last lines:\n" + "".join("    " + ln for ln in tails[failed]))
    sys.stderr.flush()
    return rc if rc > 0 else 1


COLL_CPU = False  # collectives on CPU tensors (gloo) although the work runs on a GPU: --rehearse


def dist_init(backend, rehearse=False):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if rehearse:
        # every rank on GPU 0, the ranks' exchange over gloo: the multi-rank path of the real
        # workloads on a box with one GPU (RCCL refuses two ranks on one device)
        global COLL_CPU
        COLL_CPU = True
        torch.cuda.set_device(0)
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        return rank, world, torch.device("cuda", 0)
    cuda = backend == "nccl"
    if cuda:
        torch.cuda.set_device(local if world > 1 else 0)
    # a launcher's single rank (torch.distributed.run --nproc-per-node 1) goes through the process
    # group as well: every collective of a step then runs over RCCL with one rank, which is how a
    # box with one GPU exercises that path
    if world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cuda:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        if rank == 0:
            note(f"process group: {dist.get_backend()}, {world} rank(s)")
    dev = torch.device("cuda", local if world > 1 else 0) if cuda else torch.device("cpu")
    return rank, world, dev


def group_up():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def barrier_sync(world, dev):
    if world > 1 or group_up():
        import torch.distributed as dist
        dist.barrier()
    if dev.type == "cuda":
        torch.cuda.synchronize()


def reduce_ranks(x, world, dev, op):
    if world == 1 and not group_up():
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if COLL_CPU else dev)
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
    return float(t.item())


_DEVICE_COPY = {}


def device_copy_gbs(dev):
    """Bytes read plus bytes written per second by a plain device-to-device copy of 1 GiB,
    measured here with events on the current stream: what "all of the HBM bandwidth" means on
    this very device, beside the vendor's 8 TB/s."""
    if dev.type != "cuda":
        return None
    if dev.index not in _DEVICE_COPY:
        n = 1 << 30
        a = torch.empty(n, dtype=torch.uint8, device=dev)
        b = torch.empty(n, dtype=torch.uint8, device=dev)
        for _ in range(3):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        _DEVICE_COPY[dev.index] = round(2 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del a, b
    return _DEVICE_COPY[dev.index]


def with_copy_rate(r, dev):
    """Adds the measured device-copy rate and the fraction of it to a roofline object."""
    if r is None:
        return None
    c = device_copy_gbs(dev)
    if c:
        r["device_copy"] = {"achieved": c, "unit": "GB/s", "what": "1 GiB device-to-device copy, read + written bytes"}
        r["frac_of_device_copy"] = round(r["achieved"] / c, 8)
    return r


# ------------------------------------------------------------------ workloads
class EncodeWorkload:
    """configs[2] (default) and configs[4] (`c5`): every chunk an independent .lzma stream."""

    def __init__(self, rank, world, dev, total_chunks, chunk_bytes=MIB, dict_log=22, fb=128, kind="markov",
                 label="configs[2]: encode, 1 MiB independent chunks of order-2 Markov text"):
        from lzma_amd import codec
        from tests import synth
        N = native()
        self.dev, self.world, self.rank = dev, world, rank
        self.total_chunks, self.chunk = total_chunks, chunk_bytes
        self.c0, c1 = parallel.shard_range(total_chunks, rank, world)
        self.nchunks = c1 - self.c0
        self.n = self.nchunks * chunk_bytes
        self.params = codec.make_params(dict_log, fb, 3, 0, 2, False)
        self.dict_log, self.fb, self.kind, self.label = dict_log, fb, kind, label
        L = N.lib()
        threads = usable_cores()
        # the rank's chunks are global chunks [c0, c1): the same bytes whatever the world size
        if kind == "markov":
            self.host = synth.markov(self.n, cfg=2, first_block=self.c0 * (chunk_bytes // MIB), threads=threads)
        else:
            self.host = synth.logtext(self.n, cfg=4, block=chunk_bytes, first_block=self.c0, threads=threads)
        self.d_in = torch.from_numpy(self.host).to(dev)
        self.in_off = torch.arange(0, self.n + 1, chunk_bytes, dtype=torch.int64, device=dev)
        cap = L.lzma_amd_encode_bound(chunk_bytes)
        self.out_off = torch.arange(0, (self.nchunks + 1) * cap, cap, dtype=torch.int64, device=dev)
        self.d_out = torch.empty(max(self.nchunks * cap, 1), dtype=torch.uint8, device=dev)
        self.sizes = torch.zeros(self.nchunks, dtype=torch.int64, device=dev)
        self.status = torch.zeros(self.nchunks, dtype=torch.int32, device=dev)
        self.kernel_ms = []
        self.offsets = None

    def step(self):
        N = native()
        L = N.lib()
        if self.nchunks:
            N.check(L.lzma_amd_encode_chunks_dev(self.d_in.data_ptr(), self.in_off.data_ptr(),
                                                 self.nchunks, self.n, C.byref(self.params),
                                                 self.d_out.data_ptr(), self.out_off.data_ptr(),
                                                 self.sizes.data_ptr(), self.status.data_ptr(), None))
            self.kernel_ms.append(N.last_timing())
        # container assembly: every rank learns every chunk's size -> global offsets
        allsz = parallel.gather_sizes(self.sizes.cpu() if COLL_CPU else self.sizes, self.total_chunks, self.rank,
                                      self.world)
        self.offsets = parallel.container_offsets(allsz)

    def units(self):  # uncompressed bytes per step on this rank
        return self.n

    def verify(self):
        """Every chunk of this rank goes back through the GPU decoder and must equal its input."""
        assert bool((self.status == 0).all().item()), "encoder reported a failing chunk"
        if not self.nchunks:
            return
        N = native()
        in_off = self.out_off  # streams sit at their capacity offsets inside d_out
        # decode_batch takes stream s as in[in_off[s]..in_off[s+1]): trailing slack after a
        # size-known stream is never read, so the capacity layout can be passed as it is
        out = torch.empty(self.n, dtype=torch.uint8, device=self.dev)
        osz = torch.zeros(self.nchunks, dtype=torch.int64, device=self.dev)
        st = torch.zeros(self.nchunks, dtype=torch.int32, device=self.dev)
        N.check(N.lib().lzma_amd_decode_batch_dev(self.d_out.data_ptr(), in_off.data_ptr(), self.nchunks,
                                                  out.data_ptr(), self.in_off.data_ptr(), osz.data_ptr(),
                                                  st.data_ptr(), None))
        assert bool((st == 0).all().item()), "a GPU-encoded chunk did not decode"
        assert bool((osz == self.chunk).all().item())
        assert torch.equal(out, self.d_in), "round trip through the GPU decoder differs from the input"

    def end_to_end(self):
        """One step through the host entry point (lzma_amd_encode_chunks): input and output in
        pinned host memory, the bytes crossing PCIe inside the timed region, copies on a stream
        of their own beside the kernels.  One warm-up call (it sizes the kept staging buffers),
        one timed; the sizes must equal those of the device-resident steps."""
        N = native()
        L = N.lib()
        if not self.nchunks:
            return None
        cap = L.lzma_amd_encode_bound(self.chunk)
        h_in = torch.from_numpy(self.host).pin_memory()
        h_out = torch.empty(self.nchunks * cap, dtype=torch.uint8).pin_memory()
        in_off = np.arange(0, self.n + 1, self.chunk, dtype=np.uint64)
        out_off = np.arange(0, (self.nchunks + 1) * cap, cap, dtype=np.uint64)
        sizes = np.zeros(self.nchunks, dtype=np.uint64)
        status = np.zeros(self.nchunks, dtype=np.int32)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            N.check(L.lzma_amd_encode_chunks(h_in.data_ptr(), in_off.ctypes.data, self.nchunks, C.byref(self.params),
                                             h_out.data_ptr(), out_off.ctypes.data, sizes.ctypes.data,
                                             status.ctypes.data))
            best = time.perf_counter() - t0
        assert (status == 0).all()
        assert np.array_equal(sizes.astype(np.int64), self.sizes.cpu().numpy()), "host path sizes differ"
        k = self.nchunks // 2  # a chunk's bytes came back whole
        got = h_out[k * cap: k * cap + int(sizes[k])]
        want = self.d_out[k * cap: k * cap + int(sizes[k])].cpu()
        assert torch.equal(got, want)
        return {"value": round(self.n / best / 1e6, 3), "unit": "MB/s", "ms": round(best * 1e3, 1),
                "what": "lzma_amd_encode_chunks on pinned host buffers: H2D of the input batch by batch ahead of "
                        "the finder, D2H of every batch's output region when its launch is done, all inside the "
                        "timed call", "host_bytes_in": self.n, "host_bytes_out": int(self.nchunks * cap)}

    def roofline(self):
        # encode_kernel is launched once per batch of chunks and several launches run at the
        # same time: `achieved` is per launch (bytes of one launch / its duration, as a kernel
        # trace shows it); `aggregate_achieved` is all launches' bytes / the span they cover.
        if not self.kernel_ms:
            return None
        last = self.kernel_ms[-max(1, len(self.kernel_ms) // 2):]
        launches = max(1, last[-1]["coder_launches"])
        avg_ms = sum(t["coder_ms"] for t in last) / len(last) / launches
        span_ms = sum(t["total_ms"] for t in last) / len(last)
        comp = int(self.sizes.sum().item())
        alg = self.n + comp  # N input bytes read once + C compressed bytes written
        gbs = alg / launches / (avg_ms * 1e-3) / 1e9
        # the committed counter summaries were taken on the 1 MiB Markov workload only
        counted = self.kind == "markov" and self.chunk == MIB
        return {"bound": "hbm", "kernel": "encode_kernel", "achieved": round(gbs, 4),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 8),
                "traffic": pmc_traffic("encode_kernel", self.n // launches) if counted else None,
                "kernel_ms": round(avg_ms, 3), "launches_per_step": launches,
                "algorithmic_bytes_per_launch": alg // launches,
                "aggregate_achieved": round(alg / (span_ms * 1e-3) / 1e9, 4),
                "span_ms": round(span_ms, 3),
                "finder_ms_sum": round(last[-1]["finder_ms"], 3),
                "lds": pmc_lds("encode_kernel", self.n // launches, avg_ms) if counted else None}

    def cpu_baseline(self, budget_s=25.0):
        """The oracle at the reference's default cut value, one thread per chunk on every
        usable core, on a sample of this run's chunks sized to about budget_s of CPU work."""
        from oracle import oracle as O
        O.lib()
        cores = usable_cores()
        p = O.params(dict_log=self.dict_log, fb=self.fb, lc=3, lp=0, pb=2, cut=0, eos=False)
        # each oracle encoder holds 8 * (dict + 1) bytes of tree links plus its hash table
        per_encoder = 10 * (1 << self.dict_log) + (self.chunk * 3)
        workers = max(1, min(cores, int(48e9 // per_encoder)))
        # one chunk first, to size the sample
        t0 = time.perf_counter()
        first = O.encode(self.host[: self.chunk].tobytes(), p)
        t1 = time.perf_counter() - t0
        k = int(max(workers, min(self.nchunks, 4 * workers, workers * budget_s / max(t1, 1e-3))))
        k = max(1, min(k, self.nchunks))
        chunks = [self.host[i * self.chunk:(i + 1) * self.chunk].tobytes() for i in range(k)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(workers) as ex:
            outs = list(ex.map(lambda c: O.encode(c, p), chunks))
        dt = time.perf_counter() - t0
        assert outs[0] == first
        gpu_sizes = self.sizes[:k].cpu().numpy()
        ref_total = sum(len(o) for o in outs)
        return {"value": round(k * self.chunk / dt / 1e6, 3), "unit": "MB/s", "cores": workers,
                "host_cpus": os.cpu_count(), "kind": "port",
                "sample": f"first {k} chunks of {self.chunk >> 20} MiB, one thread per chunk on {workers} threads, "
                          "reference default cut value",
                "size_ratio_gpu_over_cpu": round(float(gpu_sizes.sum()) / ref_total, 6)}

    def config(self):
        return {"workload": self.label, "chunks_total": self.total_chunks,
                "chunks_this_rank": self.nchunks, "chunk_bytes": self.chunk,
                "bytes_total": self.total_chunks * self.chunk, "dict_log": self.dict_log, "fb": self.fb,
                "lc": 3, "lp": 0, "pb": 2, "parallelism": "chunks sharded over ranks, no data-path collective"}


def mixed_streams(total_streams, lo, hi):
    """Raw sizes of the decode batch and which streams hold log lines instead of Markov text
    (one in four), identical on every rank."""
    from tests import synth
    sizes = synth.stream_sizes(total_streams, lo, hi).astype(np.int64)
    is_log = (np.arange(total_streams) % 4) == 3
    return sizes, is_log


class DecodeWorkload:
    """configs[3] shape: batch of small independent streams (4-64 KiB, EOS, level-3
    parameters, dictionary field 1 MiB), Markov text and dpkg-style log lines mixed 3:1.
    Streams are sharded over ranks by raw bytes, not by count (sizes vary 16x)."""

    def __init__(self, rank, world, dev, total_streams):
        from lzma_amd import codec
        from tests import synth
        N = native()
        self.dev, self.world, self.total = dev, world, total_streams
        sizes_all, is_log_all = mixed_streams(total_streams, 4096, 65536)
        s0, s1 = parallel.shard_by_bytes(sizes_all, rank, world)
        self.s0, self.ns = s0, s1 - s0
        self.rank = rank
        self.counts = [b - a for a, b in (parallel.shard_by_bytes(sizes_all, r, world) for r in range(world))]
        sizes = sizes_all[s0:s1]
        self.raw_sizes = sizes
        self.n = int(sizes.sum())
        raw_off = np.zeros(self.ns + 1, dtype=np.int64)
        np.cumsum(sizes, out=raw_off[1:])
        gstart = int(sizes_all[:s0].sum())  # byte offset of this shard in the whole batch
        threads = usable_cores()
        b0 = gstart // MIB
        nblocks = (gstart + self.n + MIB - 1) // MIB - b0
        text = synth.markov(max(nblocks, 1) * MIB, cfg=3, first_block=b0, threads=threads)
        self.host = np.ascontiguousarray(text[gstart - b0 * MIB: gstart - b0 * MIB + self.n])
        logs = np.flatnonzero(is_log_all[s0:s1])
        if logs.size:  # log streams: independent 64 KiB log blocks, one per global stream index
            for i in logs:
                blk = synth.logtext(65536, cfg=3, block=65536, first_block=int(s0 + i), threads=1)
                self.host[raw_off[i]: raw_off[i + 1]] = blk[: sizes[i]]
        L = N.lib()
        d_raw = torch.from_numpy(self.host).to(dev)
        d_raw_off = torch.from_numpy(raw_off).to(dev)
        caps = sizes + sizes // 8 + 128  # lzma_amd_encode_bound
        enc_off = np.zeros(self.ns + 1, dtype=np.int64)
        np.cumsum(caps, out=enc_off[1:])
        d_enc = torch.empty(max(int(enc_off[-1]), 1), dtype=torch.uint8, device=dev)
        d_enc_off = torch.from_numpy(enc_off).to(dev)
        enc_sizes = torch.zeros(self.ns, dtype=torch.int64, device=dev)
        status = torch.zeros(self.ns, dtype=torch.int32, device=dev)
        self.p = codec.params_from_level(3, eos=True)
        # the streams are produced by the GPU encoder; a sample is cross-checked against the
        # oracle's encoder in cpu_baseline() (and the encoder is parity-tested in tests/)
        if self.ns:
            N.check(L.lzma_amd_encode_chunks_dev(d_raw.data_ptr(), d_raw_off.data_ptr(), self.ns, self.n,
                                                 C.byref(self.p), d_enc.data_ptr(), d_enc_off.data_ptr(),
                                                 enc_sizes.data_ptr(), status.data_ptr(), None))
            N.release_workspace()
        assert bool((status == 0).all().item())
        # compact the streams
        es = enc_sizes.cpu().numpy()
        coff = np.zeros(self.ns + 1, dtype=np.int64)
        np.cumsum(es, out=coff[1:])
        self.comp = int(coff[-1])
        d_c = torch.empty(max(self.comp, 1), dtype=torch.uint8, device=dev)
        d_coff = torch.from_numpy(coff).to(dev)
        # in blocks of streams: index tensors of a whole 1M-stream batch would have billions of elements
        blk = 50000
        for a in range(0, self.ns, blk):
            b = min(self.ns, a + blk)
            n_b = int(coff[b] - coff[a])
            if not n_b:
                continue
            sz = enc_sizes[a:b]
            idx_src = torch.repeat_interleave(d_enc_off[a:b], sz) + (
                torch.arange(n_b, device=dev) - torch.repeat_interleave(d_coff[a:b] - d_coff[a], sz))
            d_c[int(coff[a]): int(coff[b])] = d_enc[idx_src]
            del idx_src
        del d_enc
        self.d_in, self.in_off = d_c, d_coff
        self.out_off = d_raw_off
        self.d_out = torch.empty(max(self.n, 1), dtype=torch.uint8, device=dev)
        self.sizes = torch.zeros(self.ns, dtype=torch.int64, device=dev)
        self.status = status
        self.d_raw = d_raw
        self.kernel_ms = []

    def step(self):
        N = native()
        if self.ns:
            N.check(N.lib().lzma_amd_decode_batch_dev(self.d_in.data_ptr(), self.in_off.data_ptr(), self.ns,
                                                      self.d_out.data_ptr(), self.out_off.data_ptr(),
                                                      self.sizes.data_ptr(), self.status.data_ptr(), None))
            self.kernel_ms.append(N.last_timing())
        # the streams end with a marker, their headers carry no size: where a stream's bytes go
        # in the output of the whole batch is known only when every rank's sizes are
        allsz = parallel.gather_ragged(self.sizes.cpu() if COLL_CPU else self.sizes, self.counts, self.rank,
                                       self.world)
        self.offsets = parallel.container_offsets(allsz)

    def units(self):
        return self.n

    def verify(self):
        assert bool((self.status == 0).all().item())
        assert torch.equal(self.d_out[: self.n], self.d_raw), "decoded bytes differ from the raw input"
        assert torch.equal(self.sizes.cpu(), torch.from_numpy(np.asarray(self.raw_sizes, dtype=np.int64)))
        assert self.offsets.numel() == self.total + 1
        lo = int(self.offsets[self.s0].item())
        assert int(self.offsets[self.s0 + self.ns].item()) - lo == self.n

    def end_to_end(self):
        """One step through the host entry point (lzma_amd_decode_batch) on pinned host buffers:
        compressed bytes up, raw bytes down, inside the timed call.  Warm-up call, then a timed one."""
        N = native()
        L = N.lib()
        if not self.ns:
            return None
        h_in = self.d_in[: self.comp].cpu().pin_memory()
        h_out = torch.empty(max(self.n, 1), dtype=torch.uint8).pin_memory()
        in_off = self.in_off.cpu().numpy().astype(np.uint64)
        out_off = self.out_off.cpu().numpy().astype(np.uint64)
        sizes = np.zeros(self.ns, dtype=np.uint64)
        status = np.zeros(self.ns, dtype=np.int32)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            N.check(L.lzma_amd_decode_batch(h_in.data_ptr(), in_off.ctypes.data, self.ns, h_out.data_ptr(),
                                            out_off.ctypes.data, sizes.ctypes.data, status.ctypes.data))
            best = time.perf_counter() - t0
        assert (status == 0).all()
        assert torch.equal(h_out[: self.n], torch.from_numpy(self.host)), "host path output differs from the raw input"
        return {"value": round(self.n / best / 1e6, 3), "unit": "MB/s", "ms": round(best * 1e3, 1),
                "what": "lzma_amd_decode_batch on pinned host buffers, copies inside the timed call",
                "host_bytes_in": self.comp, "host_bytes_out": self.n}

    def roofline(self):
        if not self.kernel_ms:
            return None
        ms = [t["coder_ms"] for t in self.kernel_ms[-max(1, len(self.kernel_ms) // 2):]]
        avg_ms = sum(ms) / len(ms)
        alg = self.n + self.comp
        gbs = alg / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "decode_kernel", "achieved": round(gbs, 4),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 8),
                "traffic": pmc_traffic_decode(self.n), "kernel_ms": round(avg_ms, 3),
                "algorithmic_bytes_per_launch": alg}

    def cpu_baseline(self, budget_streams=3000):
        """Oracle decoder on the first streams, one thread per stream; the same sample is
        encoded by the oracle (exhaustive cut value) and must equal the GPU-made streams."""
        from oracle import oracle as O
        O.lib()
        cores = usable_cores()
        k = min(budget_streams, self.ns)
        off = self.in_off.cpu().numpy()
        blob = self.d_in[: int(off[k])].cpu().numpy()
        streams = [blob[int(off[i]):int(off[i + 1])].tobytes() for i in range(k)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            outs = list(ex.map(lambda s: O.decode(s), streams))
        dt = time.perf_counter() - t0
        nbytes = sum(len(o[1]) for o in outs)
        ro = self.out_off.cpu().numpy()
        po = O.params(3, eos=True, cut=1 << 30)
        check = list(range(0, k, max(1, k // 64)))
        with ThreadPoolExecutor(cores) as ex:
            want = list(ex.map(lambda i: O.encode(self.host[int(ro[i]):int(ro[i + 1])].tobytes(), po), check))
        same = all(w == streams[i] for w, i in zip(want, check))
        assert same, "a GPU-encoded input stream differs from the oracle's encoding"
        return {"value": round(nbytes / dt / 1e6, 3), "unit": "MB/s", "cores": cores,
                "host_cpus": os.cpu_count(), "kind": "port",
                "sample": f"first {k} streams, one thread per stream on {cores} threads",
                "inputs_equal_oracle_encoding": f"{len(check)} sampled streams byte-identical"}

    def config(self):
        return {"workload": "configs[3] shape: batched decode of independent 4-64 KiB streams (EOS, level-3 "
                            "parameters; Markov text and log lines 3:1)",
                "streams_total": self.total, "streams_this_rank": self.ns, "raw_bytes_this_rank": self.n,
                "compressed_bytes_this_rank": self.comp, "parallelism": "streams sharded over ranks by raw bytes"}


class MfWorkload:
    """configs[1]: BT4 match finder only, one stream, 1 MiB dictionary, exhaustive."""

    def __init__(self, rank, world, dev, mib):
        from tests import synth
        self.dev = dev
        self.n = mib * MIB
        self.first_block = rank * mib
        self.host = synth.markov(self.n, cfg=1, first_block=self.first_block, threads=usable_cores())
        self.d_in = torch.from_numpy(self.host).to(dev)
        self.counts = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.offs = torch.zeros(self.n, dtype=torch.int64, device=dev)
        self.cap = 6 * self.n
        self.pairs = torch.empty(2 * self.cap, dtype=torch.int32, device=dev)
        self.total = torch.zeros(1, dtype=torch.int64, device=dev)
        self.kernel_ms = []
        self.dict, self.fb = 1 << 20, 64

    def step(self):
        N = native()
        N.check(N.lib().lzma_amd_bt4_matches_dev(self.d_in.data_ptr(), self.n, self.dict, self.fb,
                                                 self.counts.data_ptr(), self.offs.data_ptr(),
                                                 self.pairs.data_ptr(), self.cap, self.total.data_ptr(), None))
        self.kernel_ms.append(N.last_timing())

    def units(self):
        return self.n

    def verify(self):
        """Size-independent properties on a 1 % sample of the positions (every pair of a
        sampled position): the bytes `dist+1` back agree for `len` bytes and, for lengths
        from the tree search, differ right after unless the length limit was reached;
        lengths increase strictly inside a list; distances stay inside the dictionary and
        the stream; offsets are the exclusive prefix sum of the counts."""
        n, fb, dev = self.n, self.fb, self.dev
        tot = int(self.total.item())
        counts = self.counts.to(torch.int64)
        assert int(counts.sum().item()) == tot
        g = torch.Generator(device="cpu").manual_seed(17)
        P = torch.randint(0, n, (n // 100,), generator=g).to(dev)
        off = self.offs[P]
        assert torch.equal(off, (torch.cumsum(counts, 0) - counts)[P])
        cnt = counts[P]
        rep = torch.repeat_interleave(torch.arange(P.numel(), device=dev), cnt)
        within = torch.arange(rep.numel(), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
        idx = off[rep] + within
        pairs = self.pairs.view(-1, 2)
        L = pairs[idx, 0].to(torch.int64)
        D = pairs[idx, 1].to(torch.int64) + 1
        pos = P[rep]
        assert int(L.min().item()) >= 2 and int(L.max().item()) <= fb
        assert bool((D <= self.dict).all().item()) and bool((D <= pos).all().item())
        limit = torch.clamp(n - pos, max=fb)
        assert bool((L <= limit).all().item())
        same = rep[1:] == rep[:-1]
        assert bool((L[1:][same] > L[:-1][same]).all().item())
        data = self.d_in
        for k in (0, 1):  # first and last byte of every sampled match, then the middle
            a = pos + (L - 1) * k
            assert bool((data[a] == data[a - D]).all().item())
        mid = pos + L // 2
        assert bool((data[mid] == data[mid - D]).all().item())
        open_end = (L < limit) & (L >= 4)
        a = (pos + L)[open_end]
        assert bool((data[a] != data[a - D[open_end]]).all().item())
        self.digest_check()

    def digest_check(self):
        """Every 1 MiB block's match lists against the digests of the CPU restatement's lists
        (tests/golden/c2_digests.json, made by tests/golden/make_c2_digests.py from the same
        generator at the exhaustive cut value).  A list depends on its window alone, so the
        blocks of a longer or shorter stream of the same text compare too, all but the last one
        of the shorter of the two (where the stream's end limits the lengths)."""
        path = Path(__file__).resolve().parent / "tests" / "golden" / "c2_digests.json"
        self.digests_equal = None
        if self.first_block != 0 or (self.dict, self.fb) != (1 << 20, 64) or not path.exists():
            return
        from tests import c2_digest
        gold = json.loads(path.read_text())
        mib = self.n // MIB
        k = min(mib, gold["mib"]) - (0 if mib == gold["mib"] else 1)
        if k <= 0:
            return
        got = c2_digest.digests_torch(self.counts[: k * MIB], self.offs[: k * MIB], self.pairs, MIB)
        bad = [b for b in range(k) if f"{got[b]:016x}" != gold["digests"][b]]
        assert not bad, f"match lists of MiB blocks {bad[:8]} (of {len(bad)}) differ from the oracle's"
        self.digests_equal = k

    def roofline(self):
        ms = [t["finder_ms"] for t in self.kernel_ms[-max(1, len(self.kernel_ms) // 2):]]
        avg_ms = sum(ms) / len(ms)
        tot_ms = [t["total_ms"] for t in self.kernel_ms[-max(1, len(self.kernel_ms) // 2):]]
        alg = 5 * self.n + 8 * int(self.total.item())
        gbs = alg / (avg_ms * 1e-3) / 1e9
        whole = alg / (sum(tot_ms) / len(tot_ms) * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "mf_scan_kernel", "achieved": round(gbs, 4),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 8),
                "traffic": pmc_traffic_mf(self.n) if self.n == 256 * MIB else None,
                "kernel_ms": round(avg_ms, 3), "algorithmic_bytes_per_launch": alg,
                "total_ms_with_sorts": round(sum(tot_ms) / len(tot_ms), 3),
                "achieved_with_sorts": round(whole, 4), "frac_with_sorts": round(whole / HBM_PEAK_GBS, 8)}

    def cpu_baseline(self, sample_mib=8):
        from oracle import oracle as O
        O.lib()
        k = min(sample_mib * MIB, self.n)
        data = self.host[:k].tobytes()
        t0 = time.perf_counter()
        O.bt4_matches(data, self.dict, self.fb, 1 << 30)
        dt = (time.perf_counter() - t0) / 2  # bt4_matches runs the finder twice (count, fill)
        return {"value": round(k / dt / 1e6, 3), "unit": "MB/s", "cores": 1, "host_cpus": os.cpu_count(),
                "kind": "port",
                "sample": f"first {k >> 20} MiB of the stream, exhaustive cut, single thread (one stream is one chain)"}

    def config(self):
        return {"workload": "configs[1]: BT4 match finder only, one stream of order-2 Markov text, dict 1 MiB, "
                            "fb 64, exhaustive", "bytes_per_gpu": self.n,
                "blocks_equal_to_oracle_digests": getattr(self, "digests_equal", None)}

    def dump_first_mib(self, path):
        """Counts and pairs of the first MiB of positions, for a test to compare with the oracle."""
        k = min(MIB, self.n)
        c = self.counts[:k].cpu().numpy()
        tot = int(c.sum())
        np.savez(path, counts=c, pairs=self.pairs[: 2 * tot].cpu().numpy().reshape(-1, 2),
                 data=self.host[: min(self.n, k + 4096)])


class StubWorkload:
    """No GPU, no codec: lets the rank plumbing of this script be rehearsed on CPU (gloo)."""

    def __init__(self, rank, world, dev, total_chunks):
        self.rank, self.world, self.total = rank, world, total_chunks
        self.c0, c1 = parallel.shard_range(total_chunks, rank, world)
        self.sizes = torch.arange(self.c0, c1, dtype=torch.int64) + 1
        self.offsets = None

    def step(self):
        if os.environ.get("BENCH_STUB_FAIL_RANK") == str(self.rank):  # (tests: a rank that dies mid-run)
            raise RuntimeError(f"stub: rank {self.rank} fails on purpose")
        allsz = parallel.gather_sizes(self.sizes, self.total, self.rank, self.world)
        self.offsets = parallel.container_offsets(allsz)

    def units(self):
        return int(self.sizes.numel()) * MIB

    def verify(self):
        assert int(self.offsets[-1]) == self.total * (self.total + 1) // 2

    def roofline(self):
        return None

    def cpu_baseline(self):
        return None

    def config(self):
        return {"workload": "stub (rank plumbing only)", "chunks_total": self.total}


def note(msg):
    """Progress on stderr (stdout carries the one JSON line)."""
    print(f"[bench {time.perf_counter() - T_START:6.1f} s] {msg}", file=sys.stderr, flush=True)


def timed(wl, steps, warmup, world, dev, quiet=False):
    # a progress line per step on stderr (no device synchronization: steps of workloads that
    # return before the device is done are not held up by it)
    say = (lambda m: None) if (quiet or int(os.environ.get("RANK", "0")) != 0) else note
    for i in range(warmup):
        wl.step()
        say(f"warm-up step {i + 1}/{warmup}")
    barrier_sync(world, dev)
    t0 = time.perf_counter()
    for i in range(steps):
        wl.step()
        say(f"step {i + 1}/{steps}")
    barrier_sync(world, dev)
    dt = time.perf_counter() - t0
    dt = reduce_ranks(dt, world, dev, "MAX")
    units = reduce_ranks(float(wl.units()), world, dev, "SUM") * steps
    return dt, units


def extra_leg(name, make, budget_s):
    """One of the other workloads inside the encode run: 1 warm-up + 2 timed steps."""
    elapsed = time.perf_counter() - T_START
    if elapsed > budget_s:
        return {"skipped": f"{elapsed:.0f} s already elapsed (limit {budget_s:.0f} s for starting this leg)"}
    try:
        wl = make()
        dev = wl.dev
        note(f"{name}: inputs ready")
        dt, units = timed(wl, 2, 1, 1, dev, quiet=True)
        wl.verify()
        note(f"{name}: timed and verified")
        out = {"value": round(units / dt / 1e6, 3), "unit": "MB/s", "steps": 2, "warmup": 1,
               "ms_per_step": round(dt / 2 * 1e3, 3), "config": wl.config(),
               "roofline": with_copy_rate(wl.roofline(), dev), "cpu_baseline": wl.cpu_baseline()}
        if hasattr(wl, "end_to_end"):
            out["e2e"] = wl.end_to_end()
        note(f"{name}: CPU leg done")
        del wl
        torch.cuda.empty_cache()
        return out
    except Exception as e:  # an extra leg must not lose the headline line
        return {"failed": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="encode", choices=["encode", "decode", "mf", "c5", "stub"])
    ap.add_argument("--chunks", type=int, default=None,
                    help="chunks in all ranks together (encode: 4096 x 1 MiB; c5: 64 x 16 MiB)")
    ap.add_argument("--streams", type=int, default=125000, help="streams in all ranks together (decode)")
    ap.add_argument("--mib", type=int, default=256, help="stream size in MiB (mf)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: CPU rehearsal of the rank plumbing (stub workload only)")
    ap.add_argument("--rehearse", action="store_true",
                    help="all ranks on GPU 0 and their exchange over gloo: the multi-rank path on a one-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the step through the host entry point")
    ap.add_argument("--dump-first-mib", default=None,
                    help="mf: save the lists of the first MiB of positions to this .npz (tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))  # nothing here has touched a GPU yet
    if args.backend == "gloo" and args.workload != "stub":
        raise SystemExit("--backend gloo runs the stub workload only: the codec has no CPU path")

    rank, world, dev = dist_init(args.backend, args.rehearse)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if dev.type == "cuda":
        N = native()
        if N.lib().lzma_amd_device_count() < 1:
            raise SystemExit("no HIP device: the benchmark has no CPU fallback")
        N.check(N.lib().lzma_amd_set_device(dev.index))

    if args.workload == "encode":
        wl = EncodeWorkload(rank, world, dev, args.chunks or 4096)
        metric = "encode input MB/s per node (dict 4 MiB, fb 128 = the reference's level 4 row, 1 MiB chunks)"
    elif args.workload == "c5":
        wl = EncodeWorkload(rank, world, dev, args.chunks or 64, chunk_bytes=16 * MIB, dict_log=26, fb=273,
                            kind="log", label="configs[4]: encode, 16 MiB independent chunks of redundant "
                                              "dpkg-style log, dict 64 MiB, fb 273")
        metric = "encode input MB/s per node (dict 64 MiB, fb 273, 16 MiB chunks of redundant log)"
    elif args.workload == "decode":
        wl = DecodeWorkload(rank, world, dev, args.streams)
        metric = "batched decode output MB/s per node (4-64 KiB streams)"
    elif args.workload == "mf":
        wl = MfWorkload(rank, world, dev, args.mib)
        metric = "BT4 match-finder input MB/s (exhaustive, dict 1 MiB)"
    else:
        wl = StubWorkload(rank, world, dev, args.chunks or 4096)
        metric = "stub"

    if rank == 0:
        note("inputs ready")
    dt, total_units = timed(wl, args.steps, args.warmup, world, dev)
    if rank == 0:
        note(f"{args.warmup}+{args.steps} steps done")
    if not args.no_verify:
        wl.verify()
    if args.dump_first_mib and args.workload == "mf" and rank == 0:
        wl.dump_first_mib(args.dump_first_mib)
    barrier_sync(world, dev)
    if rank == 0:
        note("verified")

    if rank == 0:
        value = total_units / dt / 1e6
        line = {
            "metric": metric, "value": round(value, 3), "unit": "MB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak" if args.workload == "mf" else "strong", "vs_baseline": None,
            "dtype": "u8/u16/u32 integer", "data": "synthetic",
            "config": wl.config(), "roofline": with_copy_rate(wl.roofline(), dev),
        }
        # the CPU leg and the extra legs belong to a single-GPU run: with more ranks the
        # others would only wait for rank 0
        solo = world == 1 and dev.type == "cuda"
        line["cpu_baseline"] = wl.cpu_baseline() if (solo and not args.no_cpu_baseline) else None
        note("CPU leg done")
        if solo and hasattr(wl, "end_to_end") and not args.no_e2e and dt / max(args.steps, 1) > 40.0:
            line["e2e"] = {"skipped": "a step takes longer than 40 s: the leg's two calls would double the run"}
        elif solo and hasattr(wl, "end_to_end") and not args.no_e2e:
            try:
                line["e2e"] = wl.end_to_end()
                if line["e2e"]:
                    line["e2e_value"] = line["e2e"]["value"]
            except Exception as e:  # must not lose the headline line
                line["e2e"] = {"failed": f"{type(e).__name__}: {e}"}
            note("end-to-end leg done")
        if solo and args.workload == "encode" and not args.no_extra:
            native().release_workspace()
            del wl
            torch.cuda.empty_cache()
            line["extra"] = {
                "decode": extra_leg("decode", lambda: DecodeWorkload(0, 1, dev, args.streams), 520.0),
                "mf": extra_leg("mf", lambda: MfWorkload(0, 1, dev, args.mib), 555.0),
            }
        print(json.dumps(line), flush=True)
    if group_up():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
