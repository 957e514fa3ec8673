import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests import synth
from lzma_amd import _native as N, codec
L = N.lib()
dev = torch.device('cuda:0')
def run(nchunks, chunk=1<<20, dict_log=22, fb=128, kind='markov'):
    n = nchunks*chunk
    host = synth.markov(n) if kind=='markov' else synth.logtext(n, block=chunk)
    d_in = torch.from_numpy(host).to(dev)
    in_off = torch.arange(0, n+1, chunk, dtype=torch.int64, device=dev)
    cap = L.lzma_amd_encode_bound(chunk)
    out_off = torch.arange(0, (nchunks+1)*cap, cap, dtype=torch.int64, device=dev)
    d_out = torch.empty(nchunks*cap, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(nchunks, dtype=torch.int64, device=dev); status = torch.zeros(nchunks, dtype=torch.int32, device=dev)
    p = codec.make_params(dict_log, fb)
    torch.cuda.synchronize()
    t=time.time()
    N.check(L.lzma_amd_encode_chunks_dev(d_in.data_ptr(), in_off.data_ptr(), nchunks, n, C.byref(p), d_out.data_ptr(), out_off.data_ptr(), sizes.data_ptr(), status.data_ptr(), None))
    dt=time.time()-t
    tm = N.last_timing()
    print(f'{kind} chunks={nchunks} x {chunk>>10}KiB dict=2^{dict_log} fb={fb}: wall {dt:.3f}s = {n/dt/1e6:.1f} MB/s; finder(tables) {tm["finder_ms"]:.1f} ms, coder {tm["coder_ms"]:.1f} ms; ratio {sizes.sum().item()/n:.4f} status_ok={(status==0).all().item()}', flush=True)
run(16); run(64); run(256)
pass
run(16, chunk=1<<24, dict_log=26, fb=273, kind='log')
