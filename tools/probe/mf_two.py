"""Do two match-finder calls run side by side faster than one after the other?  Two threads, each
with its own stream and workspace, 256 MiB of text each (dict 4 MiB, fb 128: the encode benchmark's
finder shape).  python tools/probe/mf_two.py"""
import ctypes as C
import sys
import threading
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from lzma_amd import _native as N
from tests import synth

L = N.lib()
dev = torch.device("cuda:0")
n = 256 << 20
bufs = []
for k in range(2):
    d = torch.from_numpy(synth.markov(n + k)[k:k + n].copy()).to(dev)
    # counts only (no pair buffer: the lists stay in the library's arena), so tables + scan are what is timed
    bufs.append(dict(d=d, counts=torch.empty(n, dtype=torch.int32, device=dev),
                     total=torch.zeros(1, dtype=torch.int64, device=dev), stream=torch.cuda.Stream()))


def call(b):
    N.check(L.lzma_amd_bt4_matches_dev(b["d"].data_ptr(), n, 1 << 22, 128, b["counts"].data_ptr(), None, None,
                                       1 << 40, b["total"].data_ptr(), C.c_void_p(b["stream"].cuda_stream)))


def both():
    ts = [threading.Thread(target=call, args=(b,)) for b in bufs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()


for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    call(bufs[0])
    call(bufs[1])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"one after the other: {1e3 * (t1 - t0):.0f} ms", flush=True)
# persistent worker threads so that their workspaces are kept
import queue
qs = [queue.Queue() for _ in bufs]
done = queue.Queue()


def worker(b, q):
    while True:
        if q.get() is None:
            return
        call(b)
        done.put(1)


ws = [threading.Thread(target=worker, args=(b, q), daemon=True) for b, q in zip(bufs, qs)]
for w in ws:
    w.start()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for q in qs:
        q.put(1)
    done.get(), done.get()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"side by side (run {rep}): {1e3 * (t1 - t0):.0f} ms", flush=True)
for q in qs:
    q.put(None)
