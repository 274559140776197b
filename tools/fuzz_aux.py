"""Randomised sweep of the auxiliary kernels on the GPU: CRC-32 vs zlib, BLAKE2b (batch and streaming) vs hashlib,
rs-mrzip encoder vs the oracle.  usage: python tools/fuzz_aux.py [seconds] [seed]"""
import hashlib, os, random, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
from tests import _util

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = _util.Oracle(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
t_end = time.time() + budget
n_crc = n_b2 = n_rs = 0
with m.RzipContext(level=7) as ctx:
    while time.time() < t_end:
        n = rng.choice([rng.randrange(0, 300), rng.randrange(0, 200000), rng.randrange(0, 5 << 20)])
        d = w.noise(max(n, 1), seed=rng.randrange(1 << 30))[:n]
        assert ctx.crc32(d) == zlib.crc32(d)
        n_crc += 1
        msgs = [w.noise(max(k, 1), seed=rng.randrange(1 << 30))[:k] for k in (rng.randrange(0, 1000) for _ in range(rng.randrange(1, 40)))]
        outlen = rng.choice([16, 32, 64])
        assert ctx.blake2b_batch(msgs, outlen=outlen) == [hashlib.blake2b(x, digest_size=outlen).digest() for x in msgs]
        cut = sorted(rng.randrange(len(d) + 1) for _ in range(3))
        pieces = [d[:cut[0]], d[cut[0]:cut[1]], d[cut[1]:cut[2]], d[cut[2]:]]
        if len(d) < 300000:
            assert ctx.blake2b(d, pieces=pieces) == hashlib.blake2b(d).digest()
        n_b2 += len(msgs) + 1
        r = w.noise(max(n // 4, 1), seed=rng.randrange(1 << 30))[: n // 4]
        assert ctx.rs_encode(r) == oracle.rs_encode(r)
        n_rs += 1
print(f"done: crc {n_crc}, blake2b {n_b2}, rs {n_rs} ok", flush=True)
