"""A/B of the per-segment engine choice: the same chunk with MRZ_DEEP_MIN_BITS = each of the given values (the number of
bits of minimum_tag_mask from which segments run on the deep engine; 99 = never), every value in its own process.
    python tools/sweep_deep.py SHAPE GIB BITS [BITS ...]      SHAPE: noise | tar | stride | text"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shape, gib, bits = sys.argv[1], sys.argv[2], sys.argv[3:]
for b in bits:
    env = dict(os.environ, MRZ_DEEP_MIN_BITS=b)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_big.py"), shape, gib], env=env,
                       capture_output=True, text=True)
    line = r.stdout.strip().splitlines()[0] if r.stdout.strip() else r.stderr[-500:]
    try:
        d = json.loads(line)
        print(json.dumps({"deep_min_bits": int(b), "shape": shape, "gib": d["gib"], "wall_s": d["wall_s"], "GiBps": d["GiBps"],
                          "launches": d["launches"], "inserts": d["inserts"], "misses": d["misses"], "matches": d["matches"],
                          "min_mask": d["min_mask"], "s0": d["s0"]}), flush=True)
    except Exception:
        print(json.dumps({"deep_min_bits": int(b), "error": line}), flush=True)
