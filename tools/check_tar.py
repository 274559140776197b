"""GiB-scale parity + determinism check: the tar-like stream through the GPU path (several runs on one ctx, optional
second library build) against the oracle.  usage: python tools/check_tar.py [GiB] [runs] [--prof] [--no-oracle]"""
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
ORACLE = "--no-oracle" not in sys.argv
if not ORACLE:
    sys.argv.remove("--no-oracle")
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t = w.tar_like_device(int(gib * (1 << 30)), "cuda")
sigs = []
with m.RzipContext(max_chunk=t.numel(), lib=LIB) as ctx:
    for i in range(runs):
        ctx.victim_round = 0
        t0 = time.time(); res, s0, s1 = ctx.rzip_chunk(t); dt = time.time() - t0
        sig = (res.s0_len, res.s1_len, hashlib.sha256(s0).hexdigest()[:16], hashlib.sha256(s1).hexdigest()[:16])
        sigs.append(sig)
        print("gpu run", i, "%.2fs" % dt, sig, res.stats.as_dict(), "vr", ctx.victim_round, flush=True)
print("DETERMINISTIC" if len(set(sigs)) == 1 else "NONDETERMINISTIC", flush=True)
if ORACLE:
    host = t.cpu().numpy().tobytes()
    del t
    o = _util.Oracle(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
    t0 = time.time(); r = o.rzip_chunk(host); dt = time.time() - t0
    osig = (len(r["s0"]), len(r["s1"]), hashlib.sha256(r["s0"]).hexdigest()[:16], hashlib.sha256(r["s1"]).hexdigest()[:16])
    print("oracle  %.2fs" % dt, osig, r["stats"], "vr", r["victim_round"], flush=True)
    print("MATCH" if all(s == osig for s in sigs) else "MISMATCH", flush=True)
