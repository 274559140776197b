"""One rzip_chunk of a shape on the GPU (for profilers): python tools/one.py shape [LIB.so]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
shape = sys.argv[1]
lib = m.load_library(sys.argv[2]) if len(sys.argv) > 2 else m.load_library()
d = {"noise64": lambda: w.noise(64 << 20), "text32": lambda: w.zipf_text(32 << 20), "tar64": lambda: w.tar_like(64 << 20)}[shape]()
t = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
with m.RzipContext(max_chunk=t.numel(), lib=lib) as ctx:
    ctx.rzip_chunk(t, fetch=False)
