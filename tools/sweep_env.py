"""A/B of one environment knob of the library (read at mrz_open): the same chunk with NAME = each of the given values,
every value in its own process.   python tools/sweep_env.py NAME SHAPE GIB VALUE [VALUE ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, shape, gib, vals = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
for v in vals:
    env = dict(os.environ, **{name: v})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_big.py"), shape, gib], env=env, capture_output=True, text=True)
    line = r.stdout.strip().splitlines()[0] if r.stdout.strip() else r.stderr[-500:]
    try:
        d = json.loads(line)
        print(json.dumps({name: v, "shape": shape, "gib": d["gib"], "wall_s": d["wall_s"], "GiBps": d["GiBps"], "inserts": d["inserts"],
                          "misses": d["misses"], "matches": d["matches"], "s0": d["s0"]}), flush=True)
    except Exception:
        print(json.dumps({name: v, "error": line}), flush=True)
