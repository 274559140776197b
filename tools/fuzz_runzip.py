"""Corrupted-stream sweep for mrz_runzip_chunk on the GPU: random byte flips / truncations of a valid stream 0 must
give MRZ_E_CORRUPT / MRZ_E_ARG or a clean decode, never a fault or a hang.  usage: python tools/fuzz_runzip.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
from tests import _util

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
oracle = _util.Oracle(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
bases = [w.zipf_text(300000, seed=5) * 2, w.rep64k(40, seed=2, period=9000), w.tar_like(500000, seed=9)]
streams = [(d, oracle.rzip_chunk(d)) for d in bases]
ok = bad = 0
with m.RzipContext(level=7, max_chunk=1 << 20) as ctx:
    for c in range(ncases):
        d, r = streams[rng.randrange(len(streams))]
        s0 = bytearray(r["s0"])
        cb = m.chunk_bytes(len(d))
        mode = rng.randrange(4)
        if mode == 0:
            for _ in range(rng.randrange(1, 4)):
                s0[rng.randrange(len(s0))] ^= 1 << rng.randrange(8)
        elif mode == 1:
            s0 = s0[: rng.randrange(1, len(s0))]
        elif mode == 2:
            i = rng.randrange(len(s0) - 8)
            s0[i:i + 8] = bytes(rng.randrange(256) for _ in range(8))
        else:
            cb = rng.choice([1, 2, 3, 4, 5, 8])
        s1 = r["s1"] if rng.random() < 0.8 else r["s1"][: rng.randrange(len(r["s1"]) + 1)]
        try:
            back, n, cc, cs = ctx.runzip_chunk(bytes(s0), s1, cb, 4 * len(d))
            ok += 1
        except m.MrzError:
            bad += 1
print(f"done: {ncases} cases, {ok} decoded, {bad} rejected", flush=True)
