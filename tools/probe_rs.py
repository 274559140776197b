"""Throughput probe of the RS(255,223) encode+interleave kernel (device in, device out)."""
import sys, os, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
args = [a for a in sys.argv[1:] if not a.endswith(".so")]
lib = m.load_library(libs[0]) if libs else m.load_library()
gib = float(args[0]) if args else 1.0
n = int(gib * (1 << 30))
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
total = lib.mrz_rs_encoded_size(n)
dst = torch.empty(total, dtype=torch.uint8, device="cuda")
with m.RzipContext() as ctx:
    ctx.set_profiling(True)
    for i in range(3):
        rc = lib.mrz_rs_encode(ctx.ctx, ctypes.c_void_p(src.data_ptr()), n, 1, ctypes.c_void_p(dst.data_ptr()), 1, total)
        assert rc == 0, rc
        t = ctx.timings()
        rows = (n // (223 * 8176) + 1) * 8176
        alg = rows * (223 + 255)
        print(json.dumps({"lib": os.path.basename(libs[0]) if libs else "product", "n": n, "kernel_ms": round(t.encode_ms, 3), "input_GBps": round(n / t.encode_ms / 1e6, 1),
                          "alg_GBps": round(alg / t.encode_ms / 1e6, 1), "frac_of_8TBps": round(alg / t.encode_ms / 1e6 / 8000, 4)}), flush=True)
