"""One encode call on synthetic data, for rocprofv3 runs and timing experiments.

    python tools/prof_encode.py [nchunks] [chunk_bytes] [dict_log] [fb] [markov|log] [verify|-] [pb]
"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from lzma_amd import _native as N, codec
from tests import synth
nchunks = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
L = N.lib()
dev = torch.device("cuda:0")
n = nchunks * chunk
dict_log = int(sys.argv[3]) if len(sys.argv) > 3 else 22
fb = int(sys.argv[4]) if len(sys.argv) > 4 else 128
kind = sys.argv[5] if len(sys.argv) > 5 else "markov"
host = synth.markov(n) if kind == "markov" else synth.logtext(n)
d_in = torch.from_numpy(host).to(dev)
in_off = torch.arange(0, n + 1, chunk, dtype=torch.int64, device=dev)
cap = L.lzma_amd_encode_bound(chunk)
out_off = torch.arange(0, (nchunks + 1) * cap, cap, dtype=torch.int64, device=dev)
d_out = torch.empty(nchunks * cap, dtype=torch.uint8, device=dev)
sizes = torch.zeros(nchunks, dtype=torch.int64, device=dev)
status = torch.zeros(nchunks, dtype=torch.int32, device=dev)
p = codec.make_params(dict_log, fb, pb=int(sys.argv[7]) if len(sys.argv) > 7 else 2)
N.check(L.lzma_amd_encode_chunks_dev(d_in.data_ptr(), in_off.data_ptr(), nchunks, n, C.byref(p), d_out.data_ptr(),
                                     out_off.data_ptr(), sizes.data_ptr(), status.data_ptr(), None))
print("positions", n, N.last_timing(), "ratio", sizes.sum().item() / n, flush=True)
if len(sys.argv) > 6 and sys.argv[6] == "verify":  # round trip of every stream through the GPU decoder
    off = out_off.cpu().numpy()
    sz = sizes.cpu().numpy()
    streams = [d_out[int(off[i]): int(off[i]) + int(sz[i])].cpu().numpy().tobytes() for i in range(nchunks)]
    st, dec = codec.decode_batch(streams)
    assert (st == 0).all() and b"".join(dec) == host.tobytes()
    print("round trip ok", flush=True)
