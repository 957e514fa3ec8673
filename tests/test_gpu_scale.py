"""GPU runs at sizes the CPU oracle cannot follow in a test, checked through properties
that do not depend on the size:

* every encoded chunk decodes back to its input (GPU decoder for all, oracle decoder and
  liblzma for a sample), and a sample of chunks equals the oracle encoder byte for byte;
* every reported match really is one: the bytes at distance+1 back agree for `length`
  bytes and (for lengths from the tree search) differ right after unless the length limit
  was reached; lengths within a list increase strictly; distances stay inside the dictionary.
"""
import lzma as pylzma

import numpy as np
import pytest

from oracle import oracle as O
from tests import synth

pytestmark = pytest.mark.gpu

MIB = 1 << 20


def test_encode_many_chunks_round_trip():
    import lzma_amd
    n_chunks, chunk = 96, 1 << 19  # several batches' worth of workgroups, ragged tail
    buf = synth.markov(n_chunks * chunk)
    chunks = [bytes(buf[i * chunk:(i + 1) * chunk]) for i in range(n_chunks)]
    chunks[-1] = chunks[-1][:12345]
    chunks[7] = b""
    p = lzma_amd.make_params(22, 128)
    st, outs = lzma_amd.encode_chunks(chunks, p)
    assert (st == 0).all()
    st2, dec = lzma_amd.decode_batch(outs)
    assert (st2 == 0).all()
    assert dec == chunks
    for i in (0, 7, 50, n_chunks - 1):  # the oracle agrees byte for byte on a sample
        want = O.encode(chunks[i], O.params(dict_log=22, fb=128, lc=3, lp=0, pb=2, cut=1 << 30, eos=False))
        assert outs[i] == want, i
        assert O.decode(outs[i]) == (0, chunks[i])
        assert pylzma.decompress(outs[i], format=pylzma.FORMAT_ALONE) == chunks[i]


@pytest.mark.parametrize("kind,dict_size,fb", [("markov", 1 << 20, 64), ("log", 1 << 22, 273)])
def test_match_lists_are_matches(kind, dict_size, fb):
    import lzma_amd
    n = 24 * MIB
    data = synth.markov(n, cfg=1) if kind == "markov" else synth.logtext(n)
    counts, pairs = lzma_amd.bt4_matches(data.tobytes(), dict_size, fb)
    assert counts.shape[0] == n and pairs.shape[0] == int(counts.sum())
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=offs[1:])
    lens, dists = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
    pos = np.repeat(np.arange(n, dtype=np.int64), counts)
    assert lens.min() >= 2 and lens.max() <= fb
    assert (dists < dict_size).all() and (dists + 1 <= pos).all()
    limit = np.minimum(fb, n - pos)
    assert (lens <= limit).all()
    # strictly increasing lengths inside a list
    same_list = pos[1:] == pos[:-1]
    assert (lens[1:][same_list] > lens[:-1][same_list]).all()
    # sampled pairs: equal for `len` bytes, different right after unless the limit was hit
    rng = np.random.default_rng(11)
    pick = rng.choice(pairs.shape[0], size=min(200000, pairs.shape[0]), replace=False)
    P, L, D = pos[pick], lens[pick], dists[pick] + 1
    first, last = data[P] == data[P - D], data[P + L - 1] == data[P - D + L - 1]
    assert first.all() and last.all()
    mid = data[P + L // 2] == data[P - D + L // 2]
    assert mid.all()
    # (pairs of length 2 and 3 come from the hash-2 / hash-3 heads, which the reference
    # reports at their nominal length without measuring them, lz_bin_tree.go:154-180)
    open_end = (L < limit[pick]) & (L >= 4)
    assert (data[(P + L)[open_end]] != data[(P - D + L)[open_end]]).all()
    # the first megabyte equals the oracle's lists exactly
    k = MIB
    c_ref, p_ref = O.bt4_matches(data[:k + fb].tobytes(), dict_size, fb, 1 << 30)
    # positions whose length limit is not cut by the end of the short buffer
    m = k - 1
    assert np.array_equal(counts[:m], c_ref[:m])
    assert np.array_equal(pairs[: int(offs[m])], p_ref[: int(c_ref[:m].sum())])


def test_arena_grow_and_retry():
    """The match-list arena starts from an estimate; when a batch needs more the scan is
    repeated with the exact size.  Reached here by starting from a deliberately small arena
    (in a child process: the setting is read when the arena is first allocated)."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import lzma_amd
from oracle import oracle as O
from tests import synth
n = 48 << 20
data = synth.markov(n, cfg=1)
counts, pairs = lzma_amd.bt4_matches(data.tobytes(), 1 << 20, 64)
assert pairs.shape[0] == int(counts.sum())
k = 1 << 19
c_ref, p_ref = O.bt4_matches(data[:k + 64].tobytes(), 1 << 20, 64, 1 << 30)
assert np.array_equal(counts[:k - 1], c_ref[:k - 1])
assert np.array_equal(pairs[: int(counts[:k - 1].sum())], p_ref[: int(c_ref[:k - 1].sum())])
chunks = [bytes(data[i << 20:(i + 1) << 20]) for i in range(40)]
st, outs = lzma_amd.encode_chunks(chunks, lzma_amd.make_params(22, 128))
assert (st == 0).all()
st2, dec = lzma_amd.decode_batch(outs)
assert (st2 == 0).all() and dec == chunks
want = O.encode(chunks[3], O.params(dict_log=22, fb=128, lc=3, lp=0, pb=2, cut=1 << 30, eos=False))
assert outs[3] == want
print("retry ok")
'''
    import os
    env = dict(os.environ, LZMA_AMD_ARENA_TEST="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "retry ok" in r.stdout, r.stdout + r.stderr
    assert r.stderr.count("arena grown") >= 2, r.stderr  # the finder call and the encoder call both grew it


@pytest.mark.timeout(600)
def test_c5_shape_full_chunk():
    """configs[4]: one full 16 MiB chunk of the redundant log, dict 64 MiB, fb 273 (the match
    lists take the two-word form: the dictionary is larger than the one-word form allows).
    Too long for the oracle's encoder at the exhaustive cut value, so: the stream decodes back
    through the oracle's decoder and liblzma; and on a 2 MiB chunk of the same data the bytes
    equal the oracle's (exhaustive cut) and the size is within 1 % of the oracle's at the
    reference's default cut value (16 + fb/2 = 152)."""
    import lzma_amd
    chunk = 1 << 24
    data = synth.logtext(chunk + (2 << 20), block=chunk)
    big, small = bytes(data[:chunk]), bytes(data[chunk:])
    p = lzma_amd.make_params(26, 273)
    st, outs = lzma_amd.encode_chunks([big, small], p)
    assert (st == 0).all()
    rc, back = O.decode(outs[0])
    assert rc == 0 and back == big
    assert pylzma.decompress(outs[0], format=pylzma.FORMAT_ALONE) == big
    st2, dec = lzma_amd.decode_batch(outs)
    assert (st2 == 0).all() and dec == [big, small]
    want = O.encode(small, O.params(dict_log=26, fb=273, lc=3, lp=0, pb=2, cut=1 << 30, eos=False))
    assert outs[1] == want
    ref = O.encode(small, O.params(dict_log=26, fb=273, lc=3, lp=0, pb=2, cut=0, eos=False))
    assert len(outs[1]) <= 1.01 * len(ref), (len(outs[1]), len(ref))


@pytest.mark.parametrize("co_run", ["0", "1", "3"])
def test_many_batches_every_launch_order(co_run):
    """A call cut into nine small batches (more than there are match-list slots, so slots are
    reused and waiting batches have to be launched to free them), with none, one or every
    finished batch launched beside the finder.  In a child process: the settings are read per call
    but the workspace is sized by the first one."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import lzma_amd
from oracle import oracle as O
from tests import synth
n_chunks, chunk = 36, 1 << 18
buf = synth.markov(n_chunks * chunk)
chunks = [bytes(buf[i * chunk:(i + 1) * chunk]) for i in range(n_chunks)]
chunks[5] = b""
chunks[-1] = chunks[-1][:777]
# keys that are all equal, and keys with no structure, for the sorts of the batches that hold them
chunks[9] = bytes(chunk)
chunks[10] = bytes(chunk)
chunks[22] = np.random.default_rng(3).integers(0, 256, chunk, dtype=np.uint8).tobytes()
p = lzma_amd.make_params(20, 64)
from lzma_amd import _native as N
st, outs = lzma_amd.encode_chunks(chunks, p)
assert (st == 0).all()
assert N.last_timing()["coder_launches"] == 9, N.last_timing()
st2, dec = lzma_amd.decode_batch(outs)
assert (st2 == 0).all() and dec == chunks
for i in (0, 4, 5, 9, 17, 22, 35):
    want = O.encode(chunks[i], O.params(dict_log=20, fb=64, lc=3, lp=0, pb=2, cut=1 << 30, eos=False))
    assert outs[i] == want, i
print("batches ok")
'''
    env = dict(os.environ, LZMA_AMD_BATCH_MIB="1", LZMA_AMD_CO_RUN=co_run)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "batches ok" in r.stdout, r.stdout + r.stderr


def test_finder_parity_with_its_own_sort_everywhere():
    """The finder's own radix sort normally runs only beside resident encode launches; here the
    whole match-finder parity suite runs with it in place of the library's sort (child process:
    the choice is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LZMA_AMD_SORT="own")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_matchfinder.py",
                        "tests/test_gpu_scale.py::test_match_lists_are_matches", "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_two_ranks_on_one_gpu():
    """bench.py --gpus 2 with both ranks on GPU 0 and their exchange over gloo: the ranks shard
    the chunks, encode their shares with the real kernels, verify them through the GPU decoder
    and agree on the container offsets."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rehearse", "--chunks", "65", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["chunks_total"] == 65 and line["config"]["chunks_this_rank"] == 33


def test_bench_single_rank_over_rccl():
    """bench.py under the launcher the multi-GPU runs use, with one rank: the process group is
    created with the RCCL backend on the device, and the barrier, the size all-gather (encode: even
    shares; decode: the padded form) and the reductions of a step run through it on device tensors."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    for extra in (["--chunks", "9"], ["--workload", "decode", "--streams", "300"]):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "1",
                            "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra", "--no-e2e"] + extra,
                           cwd=root, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
        assert line["n_gpus"] == 1 and line["value"] > 0
        assert "process group: nccl" in r.stderr, r.stderr[-2000:]


def _bench_line(args, timeout):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py"] + args, cwd=root, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])


@pytest.mark.timeout(600)
def test_match_finder_benchmark_stream_equals_oracle_digests():
    """The match-finder benchmark at its full size (256 MiB, one stream, dictionary 1 MiB): the
    lists of every 1 MiB block have the digest of the oracle's lists at the exhaustive cut value
    (tests/golden/c2_digests.json; bench.py's verification compares all 256 and fails otherwise)."""
    line = _bench_line(["--workload", "mf", "--mib", "256", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], 500)
    assert line["config"]["blocks_equal_to_oracle_digests"] == 256


@pytest.mark.timeout(900)
def test_match_finder_beyond_two_to_the_thirty_positions(tmp_path):
    """One finder call over more than 2^30 positions (the tables' rank words then carry a flag
    instead of a group-size class): the properties of bench.py's verification, the digests of the
    first 255 MiB blocks (same text as the benchmark stream), and the first MiB of positions pair
    for pair against the oracle."""
    dump = tmp_path / "first_mib.npz"
    line = _bench_line(["--workload", "mf", "--mib", "1040", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--dump-first-mib", str(dump)], 800)
    assert line["config"]["bytes_per_gpu"] == 1040 * MIB
    assert line["config"]["blocks_equal_to_oracle_digests"] == 255
    z = np.load(dump)
    c_ref, p_ref = O.bt4_matches(z["data"].tobytes(), 1 << 20, 64, 1 << 30)
    tot = int(c_ref[:MIB].sum())
    assert np.array_equal(z["counts"], c_ref[:MIB])
    assert np.array_equal(z["pairs"].astype(np.uint32), p_ref[:tot])
