"""Throughput probes: LZ4 gate (batch of blocks), BLAKE2b batch, CRC-32."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
with m.RzipContext() as ctx:
    # LZ4 gate: 64 blocks of 4 MiB each, text / noise / mixed, resident on the device
    for kind in ("text", "noise"):
        blk = w.zipf_text(4 << 20, seed=3) if kind == "text" else w.noise(4 << 20, seed=3)
        ts = [torch.frombuffer(bytearray(blk), dtype=torch.uint8).cuda() for _ in range(64)]
        ctx.lz4_compresses(ts[:2])
        torch.cuda.synchronize()
        t0 = time.perf_counter(); r = ctx.lz4_compresses(ts, 100); dt = time.perf_counter() - t0
        print(json.dumps({"lz4_gate": kind, "blocks": 64, "block_MiB": 4, "verdict0": r[0], "s": round(dt, 4),
                          "agg_GBps": round(64 * len(blk) / dt / 1e9, 2)}), flush=True)
    # BLAKE2b batch: 4096 messages of 1 MiB
    msg = torch.randint(0, 256, (1 << 20,), dtype=torch.uint8, device="cuda")
    msgs = [msg] * 4096
    ctx.blake2b_batch(msgs[:64])
    t0 = time.perf_counter(); ctx.blake2b_batch(msgs); dt = time.perf_counter() - t0
    print(json.dumps({"blake2b_batch": 4096, "msg_MiB": 1, "s": round(dt, 4), "agg_GBps": round(4096 * (1 << 20) / dt / 1e9, 2)}), flush=True)
    one = torch.randint(0, 256, (64 << 20,), dtype=torch.uint8, device="cuda")
    t0 = time.perf_counter(); ctx.blake2b(one); dt = time.perf_counter() - t0
    print(json.dumps({"blake2b_stream_MiB": 64, "s": round(dt, 4), "GBps": round((64 << 20) / dt / 1e9, 3)}), flush=True)
    big = torch.randint(0, 256, (4 << 30,), dtype=torch.uint8, device="cuda")
    ctx.crc32(big)
    t0 = time.perf_counter(); ctx.crc32(big); dt = time.perf_counter() - t0
    print(json.dumps({"crc32_GiB": 4, "s": round(dt, 4), "GBps": round((4 << 30) / dt / 1e9, 1)}), flush=True)
