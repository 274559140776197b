"""Quick timing probe of the rzip stage on a few workload shapes (GPU box)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w

PROF = "--prof" in sys.argv
if PROF:
    sys.argv.remove("--prof")
    os.environ.setdefault("MRZ_PRINT_PROF", "1")
    LIB = m.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libmrzgpu_prof.so"))
else:
    LIB = m.load_library()

def run(name, data, level=7):
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda() if isinstance(data, bytes) else data
    n = t.numel()
    with m.RzipContext(level=level, max_chunk=n, lib=LIB) as ctx:
        ctx.set_profiling(True)
        if not PROF:
            ctx.rzip_chunk(t, fetch=False)
        t0 = time.perf_counter()
        res, _, _ = ctx.rzip_chunk(t, fetch=False)
        dt = time.perf_counter() - t0
        tm = ctx.timings()
        print(json.dumps({"name": name, "n": n, "wall_s": round(dt, 4), "MBps": round(n / dt / 1e6, 1),
                          "seq_ms": round(tm.sequencer_ms, 2), "tag_ms": round(tm.tagscan_ms, 3),
                          "enc_ms": round(tm.encode_ms, 3), "crc_ms": round(tm.crc_ms, 3), "nseg": tm.n_segments,
                          "inserts": res.stats.inserts, "hits": res.stats.tag_hits, "matches": res.stats.matches,
                          "events": res.n_events, "s0": res.s0_len, "s1": res.s1_len}), flush=True)

which = sys.argv[1:] or ["noise1", "text4", "rep64", "rep1g"]
if "noise1" in which: run("noise-1MiB", w.noise(1 << 20))
if "noise8" in which: run("noise-8MiB", w.noise(8 << 20))
if "text4" in which: run("text-4MiB", w.zipf_text(4 << 20))
if "text32" in which: run("text-32MiB", w.zipf_text(32 << 20))
if "text100" in which: run("text-100MB", w.zipf_text(100_000_000))
if "noise64" in which: run("noise-64MiB", w.noise(64 << 20))
if "noise1g" in which: run("noise-1GiB", w.noise_device(1 << 30, "cuda"))
if "noise2g" in which: run("noise-2GiB", w.noise_device(2 << 30, "cuda"))
if "noise8g" in which: run("noise-8GiB", w.noise_device(8 << 30, "cuda"))
if "noise32g" in which: run("noise-32GiB", w.noise_device(32 << 30, "cuda"))
if "tar64g" in which: run("tar-64GiB", w.tar_like_device(64 << 30, "cuda"))
if "stride64g" in which: run("stride-64GiB", w.stride_stream_device(16, 4 << 30, "cuda"))
if "tar8g" in which: run("tar-8GiB", w.tar_like_device(8 << 30, "cuda"))
if "rep64" in which: run("rep64k-64MiB", w.rep64k_device(1024, "cuda"))
if "rep1g" in which: run("rep64k-1GiB", w.rep64k_device(16384, "cuda"))
if "rep10g" in which: run("rep64k-10GiB", w.rep64k_device(163840, "cuda"))
