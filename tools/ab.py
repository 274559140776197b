"""A/B of library builds on a few shapes: python tools/ab.py LIB.so [LIB.so ...] -- shape names as tools/probe.py"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
shapes = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["noise64", "text32"]
data = {}
if "noise64" in shapes: data["noise64"] = w.noise(64 << 20)
if "text32" in shapes: data["text32"] = w.zipf_text(32 << 20)
if "tar64" in shapes: data["tar64"] = w.tar_like(64 << 20)
if "rep1g" in shapes: data["rep1g"] = w.rep64k_device(16384, "cuda")
if "rep4g" in shapes: data["rep4g"] = w.rep64k_device(65536, "cuda")
for lp in libs:
    lib = m.load_library(lp)
    row = {"lib": os.path.basename(lp)}
    for name, d in data.items():
        t = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda() if isinstance(d, bytes) else d
        with m.RzipContext(max_chunk=t.numel(), lib=lib) as ctx:
            ctx.rzip_chunk(t, fetch=False)
            ctx.victim_round = 0
            t0 = time.perf_counter()
            ctx.rzip_chunk(t, fetch=False)
            row[name] = round(time.perf_counter() - t0, 4)
    print(json.dumps(row), flush=True)
