#!/usr/bin/env python3
"""bench.py -- rzip-stage throughput of libmrzgpu on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload tar|rep64k] [--gib G] [--mode streams|window]

A "step" is one pass of the rzip stage (mrz_rzip_chunk: front end -> sequencer ->
record encoder, + CRC-32) over one chunk that is already resident in HBM.  The
metric of BASELINE.json is quoted on a 256 GiB window, which does not fit one
GPU beside its literal stream; the workload is therefore the largest single-GPU
configuration, configs[2]: the rzip stage of `mrzip -L7` on a 64 GiB synthetic
tar (S3 mix of SURVEY.md 8d: 60 % text members, 25 % noise members, 15 % exact
duplicates, 512-byte aligned), one chunk (-m pinned so max_chunk >= file size).
--workload rep64k runs configs[1] instead (10 GiB of text with 64 KiB-period
repeats; it is also one of the `shapes`).  With N > 1 every rank runs its own
stream on its own GPU (chunks / files are independent units of the reference: N
`mrzip` processes), so scaling is weak and there is no data-path collective.
--mode window (N > 1) instead splits ONE window over the ranks: range-sharded tag
scan, tags sent to rank 0 over RCCL, exact matcher on rank 0 (shard.rzip_chunk_window);
strong scaling of a front end whose back end (the matcher) does not shard.

Prints ONE JSON line on rank 0 (see the contract in the task description):
value = whole-job GiB/s of input consumed; roofline = algorithmic bytes
(N + literal bytes + stream-0 bytes) over the measured time of the
dominant kernel (the sequencer), against the 8 TB/s HBM peak; cpu_baseline = the
oracle (bit-exact C restatement of the reference's single-threaded rzip stage)
timed on a bounded prefix of the same stream on this host, 8-byte and byte-wise
match extension, and checked against a GPU run of the same prefix; shapes = the
other workload shapes (text, noise, tar mix) on GPU and oracle, measured once,
outside the timed region.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = 1 << 30
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def _oracle():
    import subprocess
    from tests import _util
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    return _util.Oracle(path)


def _sha(b):
    return hashlib.sha256(b).hexdigest()[:16]


def cpu_baseline(ctx, sample_bytes, data=None, what="rep64k"):
    """Times the oracle's rzip stage (1 core) on the first sample_bytes of the workload, with the 8-byte and the
    byte-wise match extension, and checks that a GPU run of the same prefix gives the same two streams."""
    from modern_rzip_amd import workloads
    o = _oracle()
    if data is None:
        data = workloads.rep64k(sample_bytes // 65536, seed=1234)
    else:
        data = data[:sample_bytes].cpu().numpy().tobytes()  # the same bytes the GPU has sequenced
    t0 = time.perf_counter()
    r = o.rzip_chunk(data, level=7)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    rb = o.rzip_chunk(data, level=7, bytewise=True)
    dtb = time.perf_counter() - t0
    ctx.victim_round = 0
    t0 = time.perf_counter()
    res, s0, s1 = ctx.rzip_chunk(data)
    gdt = time.perf_counter() - t0
    same = (s0, s1) == (r["s0"], r["s1"]) == (rb["s0"], rb["s1"]) and res.stats.as_dict() == r["stats"]
    if not same:
        raise SystemExit("cpu_baseline: the GPU streams of the prefix differ from the oracle's")
    return {"value": round(len(data) / GIB / dt, 5), "unit": "GiB/s", "cores": 1, "kind": "port",
            "sample": f"first {len(data) / GIB:.2f} GiB of the same {what} stream, oracle/liboracle.so "
                      f"mrzo_rzip_chunk (matcher + CRC32, no MD5), {dt:.1f} s",
            "bytewise_value": round(len(data) / GIB / dtb, 5),
            "bytewise_note": "the same with single_match_len's byte-at-a-time compare (src/rzip.c:378) instead of 8 bytes",
            "gpu_same_prefix_GiBps": round(len(data) / GIB / gdt, 5),
            "prefix_sha256": {"oracle_s0": _sha(r["s0"]), "gpu_s0": _sha(s0), "oracle_s1": _sha(r["s1"]), "gpu_s1": _sha(s1)},
            "matches": r["stats"]["matches"]}


def pmc_traffic():
    """HBM bytes per sequencer launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 +
    WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes); None if no summary is present."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["sequencer_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def shape_line(ctx, name, data, oracle_sample=None):
    """One workload shape: GPU GiB/s (one chunk, profiling events on), oracle GiB/s on the same bytes (or on a
    prefix of them), the roofline fraction recomputed the same way as for the headline."""
    o = _oracle()
    ctx.victim_round = 0
    ctx.rzip_chunk(data, fetch=False)  # warm
    ctx.victim_round = 0
    t0 = time.perf_counter()
    res, _, _ = ctx.rzip_chunk(data, fetch=False)
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    sample = data if oracle_sample is None else data[:oracle_sample]
    if hasattr(sample, "data_ptr"):
        sample = sample.cpu().numpy().tobytes()
    t0 = time.perf_counter()
    o.rzip_chunk(sample, level=7)
    odt = time.perf_counter() - t0
    alg = len(data) + res.s1_len + res.s0_len
    seq_s = tm.sequencer_ms / 1e3
    return {"name": name, "bytes": len(data), "gpu_GiBps": round(len(data) / GIB / dt, 5),
            "oracle_GiBps": round(len(sample) / GIB / odt, 5), "oracle_sample_bytes": len(sample),
            "gpu_over_oracle": round((len(data) / dt) / (len(sample) / odt), 3),
            "sequencer_ms": round(tm.sequencer_ms, 2), "launches": tm.n_segments, "narrow_launches": tm.n_narrow,
            "deep_launches": tm.n_deep,
            "roofline_frac": round(alg / seq_s / 1e9 / HBM_PEAK_GBPS, 6) if seq_s > 0 else None,
            "matches": res.stats.matches}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="tar", choices=["tar", "rep64k"],
                    help="tar: BASELINE configs[2] (64 GiB synthetic tar, the largest single-GPU configuration); rep64k: "
                         "configs[1] (10 GiB text with 64 KiB-period repeats)")
    ap.add_argument("--gib", type=float, default=0.0, help="chunk size per GPU in GiB (default: 64 for tar, 10 for rep64k)")
    ap.add_argument("--level", type=int, default=7)
    ap.add_argument("--cpu-sample-gib", type=float, default=2.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shapes", action="store_true", help="skip the other workload shapes (text, noise, tar mix)")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed GPU decode of the result")
    ap.add_argument("--mode", default="streams", choices=["streams", "window"],
                    help="streams: one independent stream per GPU (weak scaling); window: ONE window range-sharded over "
                         "the GPUs, matcher on rank 0 (strong scaling of the front end)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (rehearsal of the "
                    "multi-process path with several ranks on one GPU: ranks wrap around the visible devices)")
    args = ap.parse_args()

    import torch
    import modern_rzip_amd as m
    from modern_rzip_amd import workloads

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dev_index = local_rank
    if args.dist_backend != "nccl":
        dev_index = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.dist_backend)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    lib = m.load_library()

    if args.mode == "window" and world > 1:
        return window_mode(args, m, workloads, lib, dist, rank, world, dev_index)

    if args.gib <= 0:
        args.gib = 64.0 if args.workload == "tar" else 10.0
    nper = int(args.gib * GIB) // 65536
    n = nper * 65536
    if args.workload == "tar":
        data = workloads.tar_like_device(n, dev, seed=5 + rank)  # every rank its own stream
        wl_name = (f"tar-{args.gib:g}G: rzip stage of mrzip -L{args.level} on a {args.gib:g} GiB synthetic tar (60 % text members, 25 % "
                   f"noise, 15 % exact duplicates, 512-byte aligned; BASELINE configs[2], the largest single-GPU "
                   f"configuration), one chunk per GPU, input resident in HBM")
    else:
        data = workloads.rep64k_device(nper, dev, seed=1234 + rank)
        wl_name = (f"rep64k-{args.gib:g}G: mrzip -n -L{args.level}, {args.gib:g} GiB synthetic text with 64 KiB-period "
                   f"repeats (BASELINE configs[1]), one chunk per GPU, input resident in HBM")
    torch.cuda.synchronize()

    ctx = m.RzipContext(level=args.level, max_chunk=n, device=dev_index, lib=lib)
    ctx.set_profiling(True)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        ctx.victim_round = 0
        res, _, _ = ctx.rzip_chunk(data, fetch=False)
    barrier()
    t0 = time.perf_counter()
    seq_ms = tag_ms = enc_ms = crc_ms = 0.0
    nseg = nnarrow = ndeep = 0
    for _ in range(args.steps):
        ctx.victim_round = 0  # each step = a fresh `mrzip` process on the same file
        res, _, _ = ctx.rzip_chunk(data, fetch=False)  # returns after the device work has completed
        t = ctx.timings()
        seq_ms += t.sequencer_ms
        tag_ms += t.tagscan_ms
        enc_ms += t.encode_ms
        crc_ms += t.crc_ms
        nseg, nnarrow, ndeep = t.n_segments, t.n_narrow, t.n_deep
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # untimed self-check on every rank: the streams of the last step decode back to the input (GPU runzip)
    verify = None
    if not args.no_verify and res is not None:
        back = torch.empty(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        v0 = time.perf_counter()
        _, got, crc_calc, crc_stored = ctx.runzip_chunk((res.d_s0, res.s0_len), (res.d_s1, res.s1_len),
                                                        m.chunk_bytes(n, lib=lib), n, out=back)
        vdt = time.perf_counter() - v0
        same = got == n and crc_calc == crc_stored == res.crc32
        for a in range(0, n, 1 << 30):  # slice-wise: torch.equal materialises a temporary of the operand size
            same = same and bool(torch.equal(back[a:a + (1 << 30)], data[a:a + (1 << 30)]))
        if not same:
            raise SystemExit(f"rank {rank}: decode of the benchmark result does not reproduce the input")
        verify = {"decoded_equals_input": True, "decode_GBps": round(n / vdt / 1e9, 2)}
        del back

    if rank == 0:
        steps = args.steps
        total_in = n * steps * world
        alg_bytes = n + res.s1_len + res.s0_len  # per step per GPU
        seq_s = seq_ms / 1e3 / steps             # sequencer time per step (sum over its launches)
        achieved = alg_bytes / seq_s / 1e9 if seq_s > 0 else 0.0
        out = {
            "metric": "rzip-stage GiB/s (mrzip -n)",
            "value": round(total_in / GIB / dt, 4),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl_name, "level": args.level, "chunk_bytes_per_gpu": n, "streams": world, "mode": "streams"},
            "roofline": {"bound": "hbm", "kernel": "mrz_seq_narrow_kernel" if nnarrow * 2 > nseg else
                         ("mrz_seq_deep_kernel" if ndeep * 2 > nseg else "mrz_sequencer_kernel"),
                         "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6),
                         "traffic": pmc_traffic(), "traffic_source": "profiles (rocprofv3 --pmc passes, not this run)",
                         "launches_per_step": nseg, "narrow_launches_per_step": nnarrow, "deep_launches_per_step": ndeep,
                         "alg_bytes_per_step": alg_bytes,
                         "avg_launch_ms": round(seq_ms / steps / max(nseg, 1), 4)},
            "kernel_ms_per_step": {"sequencer": round(seq_ms / steps, 2), "tagscan": round(tag_ms / steps, 2),
                                   "encode": round(enc_ms / steps, 2), "crc32": round(crc_ms / steps, 2)},
            "result": {"s0_len": res.s0_len, "s1_len": res.s1_len, "matches": res.stats.matches,
                       "inserts": res.stats.inserts, "crc32": f"{res.crc32:08x}"},
        }
        if verify:
            out["verify"] = verify
        if args.workload == "tar" and world == 1 and not args.no_verify:
            # configs[2] is "GPU rzip + LZ4 test -> host back-end": the LZ4 compressibility gate (lz4_compresses,
            # src/stream.c:1685-1733) over the blocks the two streams of this chunk would be flushed in (128 MiB stream buffers
            # at -L7, src/stream.c:907-912), on the device, untimed -- what the gate adds to a step
            blk = 128 << 20
            blocks = [(res.d_s1 + a, min(blk, res.s1_len - a)) for a in range(0, res.s1_len, blk)]
            blocks += [(res.d_s0 + a, min(blk, res.s0_len - a)) for a in range(0, res.s0_len, blk)]
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            verdicts = ctx.lz4_compresses(blocks, threshold=100)
            gdt = time.perf_counter() - g0
            out["lz4_gate"] = {"blocks": len(blocks), "seconds": round(gdt, 3), "skip_backend": sum(1 for v in verdicts if v == 0),
                               "note": "mrz_lz4_compresses_batch over the stream blocks of the last step, device-resident, untimed"}
        if world == 1 and args.gib > 4.5 and args.workload == "rep64k":
            # the S2 generator repeats itself exactly after 4 GiB (period i + 65536 equals period i): the tail of the
            # chunk is ONE match.  Time the part that carries the matcher's work on its own.
            n4 = 4 * GIB
            ctx.victim_round = 0
            ctx.rzip_chunk(data[:n4], fetch=False)
            ctx.victim_round = 0
            t4 = time.perf_counter()
            r4, _, _ = ctx.rzip_chunk(data[:n4], fetch=False)
            d4 = time.perf_counter() - t4
            step_s = dt / steps
            out["split"] = {"first_4GiB_GiBps": round(4.0 / d4, 4), "first_4GiB_matches": r4.stats.matches,
                            "us_per_emitted_match_first_4GiB": round(d4 * 1e6 / max(r4.n_events, 1), 2),
                            "tail_GiBps": round((n - n4) / GIB / max(step_s - d4, 1e-9), 2),
                            "note": "beyond 4 GiB the stream repeats itself exactly: the tail is one match"}
        base = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            if args.workload == "tar":
                base = cpu_baseline(ctx, int(min(args.cpu_sample_gib, args.gib) * GIB), data=data, what="tar")
            else:
                base = cpu_baseline(ctx, int(min(args.cpu_sample_gib, args.gib) * GIB))
        if not args.no_shapes and world == 1:
            del data
            torch.cuda.empty_cache()
            shapes = []
            with m.RzipContext(level=args.level, max_chunk=1 << 30, device=dev_index, lib=lib) as c2:
                c2.set_profiling(True)
                shapes.append(shape_line(c2, "text-100M (S1, shape of configs[0])", workloads.zipf_text(100_000_000)))
                shapes.append(shape_line(c2, "noise-64M", workloads.noise(64 << 20)))
                shapes.append(shape_line(c2, "noise-1G", workloads.noise_device(1 << 30, dev, seed=11)))
                shapes.append(shape_line(c2, "tar-like-128M (S3 mix, shape of configs[2])", workloads.tar_like(128 << 20)))
            if args.workload == "tar":
                ctx.close()  # (frees the 64 GiB chunk's scratch)
                with m.RzipContext(level=args.level, max_chunk=10 * GIB, device=dev_index, lib=lib) as c3:
                    c3.set_profiling(True)
                    shapes.append(shape_line(c3, "rep64k-10G (BASELINE configs[1]; beyond 4 GiB the stream repeats itself: the tail "
                                                 "is one match)", workloads.rep64k_device(10 * GIB // 65536, dev, seed=1234),
                                             oracle_sample=2 * GIB))
            out["shapes"] = shapes
        if base is not None:
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def window_mode(args, m, workloads, lib, dist, rank, world, dev_index):
    """ONE window over the ranks (BASELINE configs[3], scaled to --gib): S4 stride stream.  Every rank keeps its byte range
    in a shareable allocation in its own HBM and maps all ranges into one address range (shard.window_map: HIP virtual
    memory management, peer loads over xGMI); every rank runs the front end over the stretches of its range and ships
    compacted candidate records to rank 0 (send/recv: RCCL moves device tensors GPU to GPU; gloo takes the host path);
    the exact matcher runs on rank 0 and reads the other ranks' bytes through the mapping."""
    import torch
    from modern_rzip_amd import shard
    dev = torch.device("cuda", dev_index)
    use_dev = args.dist_backend == "nccl"
    g = m.window_granularity(dev_index, lib)
    seg_bytes = max(int(args.gib * GIB) // 16 // g, 1) * g
    total = 16 * seg_bytes
    ctx = m.RzipContext(level=args.level, max_chunk=total if rank == 0 else 0, device=dev_index, lib=lib)
    # the same bytes on every rank (seeded generator on the device); a rank keeps its range only
    win = workloads.stride_stream_device(16, seg_bytes, dev, seed=99)
    wmap, part, ranges = shard.window_map(lambda o, n: win[o:o + n], total, rank, world, dist, ctx, device=dev_index)
    del win
    torch.cuda.empty_cache()
    off, size = ranges[rank]
    mine = (wmap.ptr + off, min(size + 48, total - off))
    times = []
    res = None
    for it in range(args.warmup + args.steps):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = shard.rzip_chunk_window(ctx, mine, off, total, rank, world, dist, window=(wmap.ptr, total),
                                      device=dev if use_dev else None, cap=8 << 20, ranges=ranges)
        torch.cuda.synchronize()
        dist.barrier()
        if it >= args.warmup:
            times.append(time.perf_counter() - t0)
        if rank == 0:
            res = out[0]
    if rank == 0:
        dt = sum(times)
        print(json.dumps({"metric": "rzip-stage GiB/s (mrzip -n)", "value": round(total * args.steps / GIB / dt, 4),
                          "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                          "config": {"workload": f"stride-{total / GIB:g}G: ONE window (-U) of noise segments with planted "
                                                 f"repeats (BASELINE configs[3], scaled), byte ranges in the ranks' own HBM "
                                                 f"mapped into one address range, front end range-sharded, candidate "
                                                 f"records to rank 0 over {args.dist_backend}",
                                     "mode": "window", "level": args.level, "window_bytes": total,
                                     "served": getattr(ctx, "window_served", None)},
                          "result": {"s0_len": res.s0_len, "s1_len": res.s1_len, "matches": res.stats.matches,
                                     "crc32": f"{res.crc32:08x}"}}), flush=True)
    dist.barrier()
    wmap.close()
    part.close()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
