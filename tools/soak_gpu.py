"""Randomised parity soak on the GPU, for as long as asked: draws of (data kind, length, dictionary,
fast bytes, lc, lp, pb, end marker), each encoded with one, two and three waves per chunk; every
stream must equal the oracle's at the exhaustive cut value byte for byte and decode back through
the GPU decoder.  Decoder draws: truncated and corrupted copies must give the oracle's status and
output.  Prints one line per draw (a run is never silent for long); exits non-zero at the first
difference with the draw's parameters.

    python tools/soak_gpu.py [--minutes 5] [--seed 1] [--finder] [--max-bytes 120000]

--finder: draws for the match finder alone instead (data kind, length, dictionary down to far
below the length, fast bytes): counts and pairs of every position equal to the oracle's tree
search with its cut value raised.
"""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

import lzma_amd
from oracle import oracle as O
from tests import synth


def mutated_repeats(r, n, block):
    base = r.integers(97, 123, block, dtype=np.uint8)
    out = np.tile(base, n // block + 1)[:n].copy()
    hits = r.integers(0, n, max(1, n // int(r.integers(20, 400))))
    out[hits] = r.integers(97, 123, hits.size, dtype=np.uint8)
    return out


def draw_data(r, n, markov, log):
    kind = int(r.integers(0, 7))
    if kind == 0:
        s = int(r.integers(0, len(markov) - n))
        return kind, markov[s:s + n]
    if kind == 1:
        s = int(r.integers(0, len(log) - n))
        return kind, log[s:s + n]
    if kind == 2:
        return kind, mutated_repeats(r, n, int(r.integers(2, 700)))
    if kind == 3:
        return kind, r.integers(0, 256, n, dtype=np.uint8)
    if kind == 4:  # runs of few symbols
        return kind, np.repeat(r.integers(0, 4, n // 7 + 1, dtype=np.uint8), r.integers(1, 14, n // 7 + 1))[:n]
    if kind == 5:  # text with binary islands
        s = int(r.integers(0, len(markov) - n))
        d = markov[s:s + n].copy()
        for _ in range(int(r.integers(1, 6))):
            a = int(r.integers(0, n))
            b = min(n, a + int(r.integers(1, 3000)))
            d[a:b] = r.integers(0, 256, b - a, dtype=np.uint8)
        return kind, d
    return kind, np.zeros(n, np.uint8)  # one long run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--finder", action="store_true")
    ap.add_argument("--max-bytes", type=int, default=120000, help="largest input drawn (up to 3 MiB)")
    a = ap.parse_args()
    O.lib()
    r = np.random.default_rng(a.seed)
    markov = synth.markov(4 << 20)
    log = synth.logtext(4 << 20, block=1 << 20)
    t_end = time.time() + a.minutes * 60
    it = 0
    while a.finder and time.time() < t_end:
        it += 1
        n = int(r.integers(1, a.max_bytes)) if it % 9 else int(r.integers(0, 40))
        kind, data = draw_data(r, max(n, 1), markov, log)
        data = bytes(np.asarray(data[:n], dtype=np.uint8))
        dict_size = int(1 << int(r.integers(12, 24))) if it % 3 else int(r.integers(4096, 1 << 18))
        fb = int(r.choice([5, 6, 8, 16, 32, 33, 64, 128, 255, 256, 272, 273]))
        what = dict(it=it, n=n, kind=kind, dict_size=dict_size, fb=fb, seed=a.seed)
        c_ref, p_ref = O.bt4_matches(data, dict_size, fb, cut=1 << 30)
        c_gpu, p_gpu = lzma_amd.bt4_matches(data, dict_size, fb)
        if not (np.array_equal(c_ref[:n], c_gpu[:n]) and np.array_equal(p_ref, p_gpu)):
            print("MATCH LISTS DIFFER", what, int(c_ref[:n].sum()), int(c_gpu[:n].sum()), flush=True)
            return 1
        print(f"ok {it} n={n} kind={kind} dict={dict_size} fb={fb} -> {len(p_ref)} pairs", flush=True)
    if a.finder:
        print(f"finder soak done: {it} draws, no difference", flush=True)
        return 0
    while time.time() < t_end:
        it += 1
        n = int(r.integers(1, a.max_bytes)) if it % 9 else int(r.integers(0, 40))
        kind, data = draw_data(r, max(n, 1), markov, log)
        data = bytes(np.asarray(data[:n], dtype=np.uint8))
        dict_log = int(r.integers(12, 27))
        fb = int(r.choice([5, 6, 8, 16, 32, 33, 64, 128, 255, 256, 272, 273]))
        lc, lp, pb, eos = int(r.integers(0, 9)), int(r.integers(0, 5)), int(r.integers(0, 5)), bool(r.integers(0, 2))
        what = dict(it=it, n=n, kind=kind, dict_log=dict_log, fb=fb, lc=lc, lp=lp, pb=pb, eos=eos, seed=a.seed)
        want = O.encode(data, O.params(dict_log=dict_log, fb=fb, lc=lc, lp=lp, pb=pb, cut=1 << 30, eos=eos))
        p = lzma_amd.make_params(dict_log, fb, lc, lp, pb, eos)
        for coop in ("0", "2", "3"):
            os.environ["LZMA_AMD_COOP"] = coop
            st, outs = lzma_amd.encode_chunks([data, data[: n // 2]], p)
            if not (st == 0).all() or outs[0] != want:
                print("ENCODE DIFFERS", what, "waves", coop, st, len(outs[0]), len(want), flush=True)
                return 1
        st2, dec = lzma_amd.decode_batch(outs)
        if not (st2 == 0).all() or dec != [data, data[: n // 2]]:
            print("DECODE DIFFERS", what, st2, flush=True)
            return 1
        # damaged copies of the stream: the GPU decoder must agree with the oracle's decoder
        bad = []
        for _ in range(3):
            b = bytearray(want)
            if len(b) > 14 and r.integers(0, 2):
                b[int(r.integers(13, len(b)))] ^= 1 << int(r.integers(0, 8))
            else:
                b = b[: int(r.integers(0, len(b) + 1))]
            bad.append(bytes(b))
        # (the same output capacity on both sides: a damaged stream's last match may run past the
        # size its header states, which the reference delivers and a tight buffer cannot hold)
        caps = [(n + 512) if not eos else 64 + 64 * len(b) for b in bad]
        st3, dec3 = lzma_amd.decode_batch(bad, out_caps=caps)
        for i, b in enumerate(bad):
            rc, back = O.decode(b, out_cap=caps[i])
            ok = rc == int(st3[i]) and back == dec3[i]
            if not ok:
                print("DAMAGED STREAM DIFFERS", what, i, rc, int(st3[i]), len(b), flush=True)
                return 1
        print(f"ok {it} n={n} kind={kind} dict=2^{dict_log} fb={fb} lc={lc} lp={lp} pb={pb} eos={int(eos)} -> {len(want)} B", flush=True)
    print(f"soak done: {it} draws, no difference", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
