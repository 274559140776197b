#!/usr/bin/env python3
"""Large single chunks on the GPU: wall time, kernel split, launches, result counters; optional decode check.
    python tools/probe_big.py SHAPE GIB [--verify] [--oracle-gib X]
SHAPE: noise | tar | stride | rep64k | text"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GIB = 1 << 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape")
    ap.add_argument("gib", type=float)
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--oracle-gib", type=float, default=0.0)
    ap.add_argument("--level", type=int, default=7)
    ap.add_argument("--reps", type=int, default=1)
    args = ap.parse_args()
    import torch
    import modern_rzip_amd as m
    from modern_rzip_amd import workloads
    dev = torch.device("cuda", 0)
    n = int(args.gib * GIB)
    t0 = time.perf_counter()
    if args.shape == "noise":
        data = workloads.noise_device(n, dev)
    elif args.shape == "tar":
        data = workloads.tar_like_device(n, dev)
    elif args.shape == "stride":
        seg = max(n // 16, 1 << 20)
        data = workloads.stride_stream_device(16, seg, dev)
        n = data.numel()
    elif args.shape == "rep64k":
        data = workloads.rep64k_device(n // 65536, dev)
        n = data.numel()
    elif args.shape == "text":
        data = torch.frombuffer(bytearray(workloads.zipf_text(n)), dtype=torch.uint8).to(dev)
    else:
        raise SystemExit("shape?")
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    lib = m.load_library()
    with m.RzipContext(level=args.level, max_chunk=n, lib=lib) as ctx:
        ctx.set_profiling(True)
        for rep in range(args.reps):
            ctx.victim_round = 0
            t0 = time.perf_counter()
            res, _, _ = ctx.rzip_chunk(data, fetch=False)
            dt = time.perf_counter() - t0
            tm = ctx.timings()
            out = {"shape": args.shape, "gib": round(n / GIB, 3), "gen_s": round(gen_s, 1), "wall_s": round(dt, 3),
                   "GiBps": round(n / GIB / dt, 3), "sequencer_ms": round(tm.sequencer_ms, 1),
                   "frontend_ms": round(tm.tagscan_ms, 1), "encode_ms": round(tm.encode_ms, 1), "crc_ms": round(tm.crc_ms, 1),
                   "launches": tm.n_segments, "narrow": tm.n_narrow, "matches": res.stats.matches,
                   "inserts": res.stats.inserts, "hits": res.stats.tag_hits, "misses": res.stats.tag_misses,
                   "min_mask": res.min_mask, "s0": res.s0_len, "s1": res.s1_len, "events": res.n_events}
            print(json.dumps(out), flush=True)
        if args.verify:
            back = torch.empty(n, dtype=torch.uint8, device=dev)
            _, got, crc_calc, crc_stored = ctx.runzip_chunk((res.d_s0, res.s0_len), (res.d_s1, res.s1_len),
                                                            m.chunk_bytes(n, lib=lib), n, out=back)
            same = got == n and crc_calc == crc_stored == res.crc32
            for a in range(0, n, 1 << 30):
                same = same and bool(torch.equal(back[a:a + (1 << 30)], data[a:a + (1 << 30)]))
            print(json.dumps({"decoded_equals_input": bool(same)}), flush=True)
        if args.oracle_gib > 0:
            from tests import _util
            o = _util.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
            k = int(min(args.oracle_gib * GIB, n))
            host = data[:k].cpu().numpy().tobytes()
            t0 = time.perf_counter()
            r = o.rzip_chunk(host, level=args.level)
            odt = time.perf_counter() - t0
            ctx.victim_round = 0
            t0 = time.perf_counter()
            res2, s0, s1 = ctx.rzip_chunk(data[:k])
            gdt = time.perf_counter() - t0
            print(json.dumps({"oracle_gib": round(k / GIB, 3), "oracle_GiBps": round(k / GIB / odt, 4),
                              "gpu_same_prefix_GiBps": round(k / GIB / gdt, 4),
                              "identical": (s0, s1) == (r["s0"], r["s1"]) and res2.stats.as_dict() == r["stats"]}), flush=True)


if __name__ == "__main__":
    main()
