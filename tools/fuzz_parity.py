"""Randomised parity sweep on the GPU: mrz_rzip_chunk (+ mrz_runzip_chunk) against the oracle on random shapes,
sizes, levels and victim_round values.  usage: [FUZZ_BIG=1] python tools/fuzz_parity.py [seconds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
from tests import _parity, _util

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
lib = m.load_library()
oracle = _util.Oracle(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
LEVELS = {1: (2, 2), 2: (4, 2), 3: (8, 2), 4: (16, 3), 5: (32, 4), 6: (32, 6), 7: (64, 16), 8: (64, 32), 9: (64, 128)}

def make(kind, n, s):
    if kind == "text":
        return w.zipf_text(n, seed=s)
    if kind == "noise":
        return w.noise(n, seed=s)
    if kind == "tar":
        return w.tar_like(n, seed=s)
    if kind == "rep":
        period = rng.choice([997, 4096, 30000, 65536, 200000])
        return w.rep64k(max(1, n // period), seed=s, period=period)
    if kind == "stride":
        seg = rng.choice([4096, 65536, 262144])
        return w.stride_stream(max(4, n // seg), seg, copy_bytes=rng.choice([100, seg // 4, seg - 7]), seed=s)
    if kind == "mix":
        parts = []
        while sum(map(len, parts)) < n:
            parts.append(make(rng.choice(["text", "noise", "rep"]), rng.randrange(1000, 200000), rng.randrange(1 << 30)))
            if parts and rng.random() < 0.4:
                parts.append(parts[rng.randrange(len(parts))])
        return b"".join(parts)[:n]
    raise ValueError(kind)

t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    kind = rng.choice(["text", "noise", "tar", "rep", "stride", "mix", "rep", "mix"])
    level = rng.choice([1, 2, 3, 4, 5, 6, 7, 7, 7, 8, 9])
    big = rng.random() < 0.25
    n = rng.randrange(1, 6 << 20) if big else rng.randrange(1, 300000)
    if os.environ.get("FUZZ_BIG"):  # few, large cases: culling, several segments, long farm rounds
        n = rng.randrange(4 << 20, 40 << 20)
    s = rng.randrange(1 << 30)
    data = make(kind, n, s)
    vr = rng.randrange(LEVELS[level][1])
    try:
        _parity.check_chunk(lib, oracle, data, level=level, victim_round=vr, table=rng.random() < 0.2)
    except AssertionError:
        print("MISMATCH", dict(kind=kind, level=level, n=len(data), seed=s, victim_round=vr, fuzz_seed=seed, case=cases), flush=True)
        raise
    cases += 1
    if cases % (2 if os.environ.get("FUZZ_BIG") else 25) == 0:
        print(f"{cases} cases ok, {time.time() - (t_end - budget):.0f} s", flush=True)
print(f"done: {cases} cases ok", flush=True)
