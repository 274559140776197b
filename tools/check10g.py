"""Full-size parity check: GPU streams vs oracle on the 10 GiB benchmark stream (needs ~25 GiB host RAM)."""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
from tests import _util
LIB = None
if "--prof" in sys.argv:
    sys.argv.remove("--prof")
    LIB = m.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libmrzgpu_prof.so"))
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
nper = int(gib * (1 << 30)) // 65536
t = w.rep64k_device(nper, "cuda")
with m.RzipContext(max_chunk=t.numel(), lib=LIB) as ctx:
    for i in range(2):
        ctx.victim_round = 0
        t0 = time.time(); res, s0, s1 = ctx.rzip_chunk(t); dt = time.time() - t0
        print("gpu run", i, "%.2fs" % dt, res.s0_len, res.s1_len, res.stats.as_dict(), hashlib.sha256(s0).hexdigest()[:16], hashlib.sha256(s1).hexdigest()[:16], "vr", ctx.victim_round, flush=True)
host = t.cpu().numpy().tobytes()
del t
o = _util.Oracle(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
t0 = time.time(); r = o.rzip_chunk(host); dt = time.time() - t0
print("oracle  %.2fs" % dt, len(r["s0"]), len(r["s1"]), r["stats"], hashlib.sha256(r["s0"]).hexdigest()[:16], hashlib.sha256(r["s1"]).hexdigest()[:16], "vr", r["victim_round"], flush=True)
print("MATCH" if (r["s0"], r["s1"]) == (s0, s1) else "MISMATCH")
