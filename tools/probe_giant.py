"""One giant match: a 256 MiB rep64k block followed by K exact copies of itself (the copies are one match).
Times the rzip stage; the difference to the 256 MiB block alone is the bulk compare."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
K = int(sys.argv[1]) if len(sys.argv) > 1 else 15
base = w.rep64k_device(4096, "cuda")
for k in (0, K):
    t = base.repeat(k + 1)
    n = t.numel()
    with m.RzipContext(level=7, max_chunk=n) as ctx:
        ctx.set_profiling(True)
        ctx.rzip_chunk(t, fetch=False)
        ctx.victim_round = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, _, _ = ctx.rzip_chunk(t, fetch=False)
        dt = time.perf_counter() - t0
        tm = ctx.timings()
    print(json.dumps({"copies": k, "GiB": n / 2**30, "s": round(dt, 4), "seq_ms": round(tm.sequencer_ms, 2), "tag_ms": round(tm.tagscan_ms, 2), "enc_ms": round(tm.encode_ms, 2), "crc_ms": round(tm.crc_ms, 2), "nseg": tm.n_segments, "events": res.n_events, "match_bytes": res.stats.match_bytes}), flush=True)
