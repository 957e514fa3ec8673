#!/usr/bin/env python3
"""Headline benchmark: chunked LZMA encode on MI355X, input MB/s.

Workload (default, BASELINE.json configs[2]): 4096 independent 1 MiB chunks (4 GiB)
of synthetic order-2 Markov text per GPU, encoder parameters {dict 4 MiB, fb 128,
lc3 lp0 pb2, size-known header}.  One process per GPU; chunks are sharded across
ranks with no data-path collective; the only exchange is an all-gather of the
per-chunk compressed sizes (RCCL) from which every rank derives the container
offsets.  A "step" is one pass of lzma_amd_encode_chunks_dev over the rank's
chunks with the input already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload encode|decode|mf]
                    [--chunks C] [--no-cpu-baseline]

Prints ONE JSON line (rank 0).  Also reports:
  roofline     -- algorithmic bytes (N + C) of the dominant kernel / its HIP-event time
  cpu_baseline -- the CPU oracle (a C restatement of the reference, default cut value)
                  on a bounded sample of the same chunks, on this host's cores
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from lzma_amd import _native as N  # noqa: E402
from lzma_amd import codec  # noqa: E402
from lzma_amd import parallel  # noqa: E402
from tests import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0
MIB = 1 << 20


def pmc_traffic(kernel, input_bytes_per_launch):
    """HBM bytes per launch of `kernel`, from the committed rocprofv3 PMC summary of this
    workload (FETCH_SIZE + WRITE_SIZE per input byte, measured in separate counter runs),
    scaled to this run's launch size.  None if there is no summary."""
    path = ROOT / "profiles" / "r01_pmc_encode_832x1MiB.json"
    try:
        k = json.loads(path.read_text())["kernels"][kernel]
    except (OSError, KeyError, ValueError):
        return None
    return int(k["hbm_bytes_per_input_byte_reported"] * input_bytes_per_launch)


def pmc_traffic_decode(output_bytes_per_launch):
    """The same for decode_kernel, whose summary is per output byte."""
    path = ROOT / "profiles" / "r01_pmc_decode_125k_streams.json"
    try:
        k = json.loads(path.read_text())["decode_kernel_per_output_byte"]
        return int(k["hbm_bytes_per_output_byte_reported"] * output_bytes_per_launch)
    except (OSError, KeyError, ValueError):
        return None


def dist_init(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    return rank, world, local


def barrier_sync(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class EncodeWorkload:
    """configs[2]: level-5-shaped encode of 1 MiB chunks of Markov text."""
    name = "encode_c3"

    def __init__(self, rank, world, dev, chunks, chunk_bytes=MIB, dict_log=22, fb=128):
        self.dev, self.world, self.rank = dev, world, rank
        self.nchunks, self.chunk = chunks, chunk_bytes
        self.n = chunks * chunk_bytes
        self.params = codec.make_params(dict_log, fb, 3, 0, 2, False)
        self.dict_log, self.fb = dict_log, fb
        L = N.lib()
        threads = min(16, os.cpu_count() or 1)
        # rank r owns global 1 MiB blocks [r*chunks, (r+1)*chunks): distinct data per rank
        self.host = synth.markov(self.n, cfg=2, first_block=rank * (self.n // MIB), threads=threads)
        self.d_in = torch.from_numpy(self.host).to(dev)
        self.in_off = torch.arange(0, self.n + 1, chunk_bytes, dtype=torch.int64, device=dev)
        cap = L.lzma_amd_encode_bound(chunk_bytes)
        self.out_off = torch.arange(0, (chunks + 1) * cap, cap, dtype=torch.int64, device=dev)
        self.d_out = torch.empty(chunks * cap, dtype=torch.uint8, device=dev)
        self.sizes = torch.zeros(chunks, dtype=torch.int64, device=dev)
        self.status = torch.zeros(chunks, dtype=torch.int32, device=dev)
        self.kernel_ms = []
        self.offsets = None

    def step(self):
        L = N.lib()
        N.check(L.lzma_amd_encode_chunks_dev(self.d_in.data_ptr(), self.in_off.data_ptr(),
                                             self.nchunks, self.n, C.byref(self.params),
                                             self.d_out.data_ptr(), self.out_off.data_ptr(),
                                             self.sizes.data_ptr(), self.status.data_ptr(), None))
        self.kernel_ms.append(N.last_timing())
        # container assembly: every rank learns every chunk's size -> global offsets
        allsz = parallel.gather_sizes(self.sizes, self.world * self.nchunks, self.rank, self.world)
        self.offsets = parallel.container_offsets(allsz)

    def units(self):  # uncompressed bytes per step on this rank
        return self.n

    def verify(self):
        assert bool((self.status == 0).all().item()), "encoder reported a failing chunk"
        # round trip of a few chunks through the GPU decoder as a cheap end-to-end check
        sizes = self.sizes.cpu().numpy()
        off = self.out_off.cpu().numpy()
        out = self.d_out
        streams = [out[int(off[i]): int(off[i]) + int(sizes[i])].cpu().numpy().tobytes()
                   for i in (0, self.nchunks - 1)]
        st, dec = codec.decode_batch(streams)
        assert (st == 0).all()
        assert dec[0] == self.host[: self.chunk].tobytes()
        assert dec[1] == self.host[-self.chunk:].tobytes()

    def roofline(self):
        # encode_kernel is launched once per batch of chunks and several launches run at the
        # same time: `achieved` is per launch (bytes of one launch / its duration, as a kernel
        # trace shows it); `aggregate_achieved` is all launches' bytes / the span they cover.
        last = self.kernel_ms[-max(1, len(self.kernel_ms) // 2):]
        launches = max(1, last[-1]["coder_launches"])
        avg_ms = sum(t["coder_ms"] for t in last) / len(last) / launches
        span_ms = sum(t["total_ms"] for t in last) / len(last)
        comp = int(self.sizes.sum().item())
        alg = self.n + comp  # N input bytes read once + C compressed bytes written
        gbs = alg / launches / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "encode_kernel", "achieved": round(gbs, 4),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 8),
                "traffic": pmc_traffic("encode_kernel", self.n // launches),
                "kernel_ms": round(avg_ms, 3), "launches_per_step": launches,
                "algorithmic_bytes_per_launch": alg // launches,
                "aggregate_achieved": round(alg / (span_ms * 1e-3) / 1e9, 4),
                "span_ms": round(span_ms, 3),
                "finder_ms_sum": round(last[-1]["finder_ms"], 3)}

    def cpu_baseline(self, budget_chunks=48):
        from oracle import oracle as O
        O.lib()
        cores = min(16, os.cpu_count() or 1)
        k = min(budget_chunks, self.nchunks)
        p = O.params(dict_log=self.dict_log, fb=self.fb, lc=3, lp=0, pb=2, cut=0, eos=False)
        chunks = [self.host[i * self.chunk:(i + 1) * self.chunk].tobytes() for i in range(k)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            outs = list(ex.map(lambda c: O.encode(c, p), chunks))
        dt = time.perf_counter() - t0
        gpu_sizes = self.sizes[:k].cpu().numpy()
        ref_total = sum(len(o) for o in outs)
        return {"value": round(k * self.chunk / dt / 1e6, 3), "unit": "MB/s", "cores": cores,
                "kind": "port",
                "sample": f"first {k} chunks of 1 MiB, one thread per chunk, reference default cut value",
                "size_ratio_gpu_over_cpu": round(float(gpu_sizes.sum()) / ref_total, 6)}

    def config(self):
        return {"workload": "configs[2]: encode, 1 MiB independent chunks of order-2 Markov text",
                "chunks_per_gpu": self.nchunks, "chunk_bytes": self.chunk,
                "bytes_per_gpu": self.n, "dict_log": self.dict_log, "fb": self.fb,
                "lc": 3, "lp": 0, "pb": 2, "parallelism": "chunks sharded by rank"}


class DecodeWorkload:
    """configs[3] shape: batch of small independent streams (4-64 KiB, EOS, level-3 params)."""
    name = "decode_c4"

    def __init__(self, rank, world, dev, streams):
        self.dev, self.world, self.ns = dev, world, streams
        sizes = synth.stream_sizes(streams * world)[rank * streams:(rank + 1) * streams].astype(np.int64)
        self.raw_sizes = sizes
        self.n = int(sizes.sum())
        nblocks = (self.n + MIB - 1) // MIB
        threads = min(16, os.cpu_count() or 1)
        self.host = synth.markov(nblocks * MIB, cfg=3, first_block=rank * nblocks, threads=threads)[: self.n]
        L = N.lib()
        raw_off = np.zeros(streams + 1, dtype=np.int64)
        np.cumsum(sizes, out=raw_off[1:])
        d_raw = torch.from_numpy(self.host).to(dev)
        d_raw_off = torch.from_numpy(raw_off).to(dev)
        caps = np.array([L.lzma_amd_encode_bound(int(s)) for s in sizes], dtype=np.int64)
        enc_off = np.zeros(streams + 1, dtype=np.int64)
        np.cumsum(caps, out=enc_off[1:])
        d_enc = torch.empty(int(enc_off[-1]), dtype=torch.uint8, device=dev)
        d_enc_off = torch.from_numpy(enc_off).to(dev)
        enc_sizes = torch.zeros(streams, dtype=torch.int64, device=dev)
        status = torch.zeros(streams, dtype=torch.int32, device=dev)
        p = codec.params_from_level(3, eos=True)
        # the streams are produced by the GPU encoder (parity-tested against the oracle)
        N.check(L.lzma_amd_encode_chunks_dev(d_raw.data_ptr(), d_raw_off.data_ptr(), streams, self.n,
                                             C.byref(p), d_enc.data_ptr(), d_enc_off.data_ptr(),
                                             enc_sizes.data_ptr(), status.data_ptr(), None))
        assert bool((status == 0).all().item())
        # compact the streams
        es = enc_sizes.cpu().numpy()
        coff = np.zeros(streams + 1, dtype=np.int64)
        np.cumsum(es, out=coff[1:])
        self.comp = int(coff[-1])
        d_c = torch.empty(self.comp, dtype=torch.uint8, device=dev)
        idx_src = torch.repeat_interleave(d_enc_off[:-1], enc_sizes) + (
            torch.arange(self.comp, device=dev) - torch.repeat_interleave(torch.from_numpy(coff[:-1]).to(dev), enc_sizes))
        d_c.copy_(d_enc[idx_src])
        del d_enc, idx_src
        self.d_in, self.in_off = d_c, torch.from_numpy(coff).to(dev)
        self.out_off = d_raw_off
        self.d_out = torch.empty(self.n, dtype=torch.uint8, device=dev)
        self.sizes = torch.zeros(streams, dtype=torch.int64, device=dev)
        self.status = status
        self.d_raw = d_raw
        self.kernel_ms = []

    def step(self):
        N.check(N.lib().lzma_amd_decode_batch_dev(self.d_in.data_ptr(), self.in_off.data_ptr(), self.ns,
                                                  self.d_out.data_ptr(), self.out_off.data_ptr(),
                                                  self.sizes.data_ptr(), self.status.data_ptr(), None))
        self.kernel_ms.append(N.last_timing())

    def units(self):
        return self.n

    def verify(self):
        assert bool((self.status == 0).all().item())
        assert torch.equal(self.d_out, self.d_raw), "decoded bytes differ from the raw input"

    def roofline(self):
        ms = [t["coder_ms"] for t in self.kernel_ms[-max(1, len(self.kernel_ms) // 2):]]
        avg_ms = sum(ms) / len(ms)
        alg = self.n + self.comp
        gbs = alg / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "decode_kernel", "achieved": round(gbs, 4),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 8),
                "traffic": pmc_traffic_decode(self.n), "kernel_ms": round(avg_ms, 3),
                "algorithmic_bytes_per_launch": alg}

    def cpu_baseline(self, budget_streams=3000):
        from oracle import oracle as O
        O.lib()
        cores = min(16, os.cpu_count() or 1)
        k = min(budget_streams, self.ns)
        off = self.in_off.cpu().numpy()
        blob = self.d_in[: int(off[k])].cpu().numpy()
        streams = [blob[int(off[i]):int(off[i + 1])].tobytes() for i in range(k)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            outs = list(ex.map(lambda s: O.decode(s), streams))
        dt = time.perf_counter() - t0
        nbytes = sum(len(o[1]) for o in outs)
        return {"value": round(nbytes / dt / 1e6, 3), "unit": "MB/s", "cores": cores, "kind": "port",
                "sample": f"first {k} streams, one thread per stream"}

    def config(self):
        return {"workload": "configs[3] shape: batched decode of independent 4-64 KiB streams (EOS, level-3 params)",
                "streams_per_gpu": self.ns, "raw_bytes_per_gpu": self.n, "compressed_bytes_per_gpu": self.comp,
                "parallelism": "streams sharded by rank"}


class MfWorkload:
    """configs[1]: BT4 match finder only, one stream, 1 MiB dictionary, exhaustive."""
    name = "mf_c2"

    def __init__(self, rank, world, dev, mib):
        self.dev = dev
        self.n = mib * MIB
        threads = min(16, os.cpu_count() or 1)
        self.host = synth.markov(self.n, cfg=1, first_block=rank * mib, threads=threads)
        self.d_in = torch.from_numpy(self.host).to(dev)
        self.counts = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.offs = torch.zeros(self.n, dtype=torch.int64, device=dev)
        self.cap = 6 * self.n
        self.pairs = torch.empty(2 * self.cap, dtype=torch.int32, device=dev)
        self.total = torch.zeros(1, dtype=torch.int64, device=dev)
        self.kernel_ms = []

    def step(self):
        N.check(N.lib().lzma_amd_bt4_matches_dev(self.d_in.data_ptr(), self.n, 1 << 20, 64,
                                                 self.counts.data_ptr(), self.offs.data_ptr(),
                                                 self.pairs.data_ptr(), self.cap, self.total.data_ptr(), None))
        self.kernel_ms.append(N.last_timing())

    def units(self):
        return self.n

    def verify(self):
        # size-independent properties: lengths strictly increase within a list unless the hash-2
        # pair survived a dropped hash-3 pair; distances are inside the dictionary
        tot = int(self.total.item())
        p = self.pairs[: 2 * tot].view(-1, 2)
        assert int(p[:, 1].max().item()) < (1 << 20) and int(p[:, 0].max().item()) <= 64
        assert int(p[:, 0].min().item()) >= 2

    def roofline(self):
        ms = [t["finder_ms"] for t in self.kernel_ms[-max(1, len(self.kernel_ms) // 2):]]
        avg_ms = sum(ms) / len(ms)
        alg = 5 * self.n + 8 * int(self.total.item())
        gbs = alg / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "mf_scan_kernel", "achieved": round(gbs, 4),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 8), "traffic": None,
                "kernel_ms": round(avg_ms, 3), "algorithmic_bytes_per_launch": alg,
                "total_ms_with_sorts": round(self.kernel_ms[-1]["total_ms"], 3)}

    def cpu_baseline(self, sample_mib=8):
        from oracle import oracle as O
        O.lib()
        k = min(sample_mib * MIB, self.n)
        data = self.host[:k].tobytes()
        t0 = time.perf_counter()
        O.bt4_matches(data, 1 << 20, 64, 1 << 30)
        dt = (time.perf_counter() - t0) / 2  # bt4_matches runs the finder twice (count, fill)
        return {"value": round(k / dt / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                "sample": f"first {k >> 20} MiB of the stream, exhaustive cut, single thread"}

    def config(self):
        return {"workload": "configs[1]: BT4 match finder only, one stream of order-2 Markov text, dict 1 MiB, fb 64, exhaustive",
                "bytes_per_gpu": self.n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="encode", choices=["encode", "decode", "mf"])
    ap.add_argument("--chunks", type=int, default=4096, help="1 MiB chunks per GPU (encode)")
    ap.add_argument("--streams", type=int, default=125000, help="streams per GPU (decode)")
    ap.add_argument("--mib", type=int, default=256, help="stream size in MiB (mf)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    rank, world, local = dist_init(args.gpus)
    dev = torch.device("cuda", local if world > 1 else 0)
    if N.lib().lzma_amd_device_count() < 1:
        raise SystemExit("no HIP device: the benchmark has no CPU fallback")
    N.check(N.lib().lzma_amd_set_device(dev.index))

    if args.workload == "encode":
        wl = EncodeWorkload(rank, world, dev, args.chunks)
        metric = "encode input MB/s per node (level-5 shape: dict 4 MiB, fb 128, 1 MiB chunks)"
    elif args.workload == "decode":
        wl = DecodeWorkload(rank, world, dev, args.streams)
        metric = "batched decode output MB/s per node (4-64 KiB streams)"
    else:
        wl = MfWorkload(rank, world, dev, args.mib)
        metric = "BT4 match-finder input MB/s (exhaustive, dict 1 MiB)"

    for _ in range(args.warmup):
        wl.step()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier_sync(world)
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world, dev)
    total_units = sum_over_ranks(float(wl.units()), world, dev) * args.steps
    if not args.no_verify:
        wl.verify()

    if rank == 0:
        value = total_units / dt / 1e6
        line = {
            "metric": metric, "value": round(value, 3), "unit": "MB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u16/u32 integer", "data": "synthetic",
            "config": wl.config(), "roofline": wl.roofline(),
        }
        # the CPU leg is a single-GPU report: with more ranks the others would sit in the final barrier
        line["cpu_baseline"] = wl.cpu_baseline() if (world == 1 and not args.no_cpu_baseline) else None
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
