"""Decoder timing probe (GPU box): encode a stream on the GPU, decode it in HBM, report GB/s of output."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w

def run(name, t):
    if isinstance(t, bytes):
        t = torch.frombuffer(bytearray(t), dtype=torch.uint8).cuda()
    n = t.numel()
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    with m.RzipContext(level=7, max_chunk=n) as ctx:
        res, _, _ = ctx.rzip_chunk(t, fetch=False)
        cb = m.chunk_bytes(n)
        args = ((res.d_s0, res.s0_len), (res.d_s1, res.s1_len), cb, n)
        ctx.runzip_chunk(*args, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, got, cc, cs = ctx.runzip_chunk(*args, out=out)
        dt = time.perf_counter() - t0
    ok = bool(torch.equal(out, t)) and cc == cs
    print(json.dumps({"name": name, "n": n, "s0": res.s0_len, "s1": res.s1_len, "decode_s": round(dt, 4),
                      "GBps": round(n / dt / 1e9, 2), "ok": ok}), flush=True)

which = sys.argv[1:] or ["noise", "text", "rep1g"]
if "noise" in which: run("noise-64MiB", w.noise(64 << 20))
if "text" in which: run("text-32MiB", w.zipf_text(32 << 20))
if "rep1g" in which: run("rep64k-1GiB", w.rep64k_device(16384, "cuda"))
if "rep10g" in which: run("rep64k-10GiB", w.rep64k_device(163840, "cuda"))
