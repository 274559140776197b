"""Randomised sweep of the LZ4 size kernel and the gate on the GPU against the oracle (liblz4 1.9.3 restatement).
usage: python tools/fuzz_lz4.py [seconds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
from tests import _util

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = _util.Oracle(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))

def block():
    kind = rng.randrange(6)
    n = rng.choice([rng.randrange(0, 200), rng.randrange(0, 70000), rng.randrange(60000, 70000), rng.randrange(0, 600000)])
    s = rng.randrange(1 << 30)
    if kind == 0:
        return w.zipf_text(max(n, 1), seed=s)[:n]
    if kind == 1:
        return w.noise(max(n, 1), seed=s)[:n]
    if kind == 2:
        return (w.noise(rng.randrange(1, 300), seed=s) * (n // 50 + 1))[:n]
    if kind == 3:
        return bytes([rng.randrange(4)]) * n
    if kind == 4:
        return w.tar_like(max(n, 1024), seed=s)[:n]
    a = bytearray(w.zipf_text(max(n, 1), seed=s)[:n])
    for _ in range(len(a) // 97):
        a[rng.randrange(len(a))] = rng.randrange(256)
    return bytes(a)

t_end = time.time() + budget
cases = 0
with m.RzipContext(level=7) as ctx:
    while time.time() < t_end:
        blocks = [block() for _ in range(48)]
        got = ctx.lz4_sizes(blocks)
        want = [oracle.lz4_size(b) for b in blocks]
        assert got == want, [(len(b), g, x) for b, g, x in zip(blocks, got, want) if g != x][:3]
        thr = rng.choice([100, 98, 90, 50, 10])
        gate = ctx.lz4_compresses(blocks, threshold=thr)
        assert gate == [oracle.lz4_compresses(b, thr) for b in blocks]
        cases += len(blocks)
print(f"done: {cases} blocks ok", flush=True)
