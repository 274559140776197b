"""S independent rep64k streams on ONE GPU at once (one ctx + host thread each): aggregate GiB/s.
usage: python tools/probe_multistream.py S GIB_PER_STREAM   (helpers are split 224 / S per stream)"""
import sys, os, time, json, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w

S = int(sys.argv[1]); gib = float(sys.argv[2])
nper = int(gib * (1 << 30)) // 65536
n = nper * 65536
data = [w.rep64k_device(nper, "cuda", seed=1234 + i) for i in range(S)]
ctxs = [m.RzipContext(level=7, max_chunk=n) for _ in range(S)]
for c in ctxs:
    c.set_farm_helpers(224 // S)
res = [None] * S
def work(i):
    ctxs[i].victim_round = 0
    res[i], _, _ = ctxs[i].rzip_chunk(data[i], fetch=False)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
print(json.dumps({"streams": S, "gib_per_stream": gib, "farm_wgs": os.environ.get("MRZ_FARM_WGS"), "wall_s": round(dt, 3),
                  "agg_GiBps": round(S * n / (1 << 30) / dt, 3), "crc": [f"{r.crc32:08x}" for r in res][:3]}), flush=True)
