"""Writes tests/golden/c2_digests.json: per 1 MiB block of the 256 MiB match-finder benchmark
stream (order-2 Markov text, generator configuration 1), a 64-bit digest of the oracle's match
lists at dictionary 1 MiB, 64 fast bytes, cut value 2^30 (exhaustive).

    python tests/golden/make_c2_digests.py [MiB]

Run in the build container (CPU only; about ten minutes -- the oracle walks the stream twice, once to count the pairs -- and 20 GB of memory for 256 MiB).
The GPU test and bench.py's match-finder workload compute the same digests from the GPU's
lists and compare every block.
"""
import hashlib
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402
from tests import c2_digest, synth  # noqa: E402

MIB = 1 << 20


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    data = synth.markov(mib * MIB, cfg=1, first_block=0)
    t0 = time.time()
    counts, pairs = O.bt4_matches(data.tobytes(), 1 << 20, 64, 1 << 30)
    print(f"oracle: {time.time() - t0:.0f} s, {pairs.shape[0]} pairs", flush=True)
    dig = c2_digest.digests_np(counts, pairs, MIB)
    out = {
        "what": "per-MiB digests (tests/c2_digest.py) of the oracle's BT4 match lists, cut value 2^30",
        "stream": f"tests.synth.markov({mib} MiB, cfg=1, first_block=0)",
        "stream_sha256": hashlib.sha256(data.tobytes()).hexdigest(),
        "mib": mib, "dict_size": 1 << 20, "fb": 64, "total_pairs": int(pairs.shape[0]),
        "digests": [f"{d:016x}" for d in dig],
    }
    name = "c2_digests.json" if mib == 256 else f"c2_digests_{mib}mib.json"
    (Path(__file__).resolve().parent / name).write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", name)


if __name__ == "__main__":
    main()
