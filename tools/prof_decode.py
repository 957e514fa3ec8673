"""One batched decode on synthetic streams, for rocprofv3 runs and timing experiments.

    python tools/prof_decode.py [nstreams] [stream_bytes] [markov|random|zeros]

random: incompressible bytes, every symbol a literal; zeros: one literal, then matches of the
longest length at distance 1; markov: the text the benchmark uses.  The three together separate
the cost of a literal, of a match and of the copy.
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from lzma_amd import _native as N, codec
from tests import synth
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
size = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
kind = sys.argv[3] if len(sys.argv) > 3 else "markov"
n = ns * size
if kind == "markov":
    host = synth.markov(n, cfg=3)
elif kind == "random":
    host = np.random.default_rng(5).integers(0, 256, n, dtype=np.uint8)
else:
    host = np.zeros(n, dtype=np.uint8)
chunks = [host[i * size:(i + 1) * size].tobytes() for i in range(ns)]
st, streams = codec.encode_chunks(chunks, codec.params_from_level(3, eos=True))
assert (st == 0).all()
comp = sum(len(s) for s in streams)
dev = torch.device("cuda:0")
L = N.lib()
in_off = np.zeros(ns + 1, dtype=np.int64)
np.cumsum([len(s) for s in streams], out=in_off[1:])
d_in = torch.from_numpy(np.frombuffer(b"".join(streams), dtype=np.uint8).copy()).to(dev)
d_in_off = torch.from_numpy(in_off).to(dev)
d_out_off = torch.arange(0, n + 1, size, dtype=torch.int64, device=dev)
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
sizes = torch.zeros(ns, dtype=torch.int64, device=dev)
status = torch.zeros(ns, dtype=torch.int32, device=dev)
for _ in range(2):
    N.check(L.lzma_amd_decode_batch_dev(d_in.data_ptr(), d_in_off.data_ptr(), ns, d_out.data_ptr(), d_out_off.data_ptr(),
                                        sizes.data_ptr(), status.data_ptr(), None))
    t = N.last_timing()
    print(kind, "out bytes", n, "compressed", comp, "ms", round(t["coder_ms"], 2), "GB/s", round(n / t["coder_ms"] / 1e6, 2),
          flush=True)
assert bool((status == 0).all().item()) and torch.equal(d_out.cpu(), torch.from_numpy(host))
