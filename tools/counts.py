"""Lane-level counters of the wide sequencer (a -DMRZ_SEQ_STATS -DMRZ_SEQ_COUNTS build): python tools/counts.py LIB.so shape"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRZ_PRINT_PROF"] = "1"
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
lib = m.load_library(sys.argv[1])
shape = sys.argv[2]
d = {"noise64": lambda: w.noise(64 << 20), "text32": lambda: w.zipf_text(32 << 20), "tar64": lambda: w.tar_like(64 << 20)}[shape]()
t = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
with m.RzipContext(max_chunk=t.numel(), lib=lib) as ctx:
    ctx.rzip_chunk(t, fetch=False)
