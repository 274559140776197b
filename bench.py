#!/usr/bin/env python3
"""bench.py -- rzip-stage throughput of libmrzgpu on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--gib G]

A "step" is one pass of the rzip stage (mrz_rzip_chunk: tag scan -> sequencer ->
record encoder, + CRC-32) over one chunk that is already resident in HBM.  The
workload is BASELINE.json configs[1]: `mrzip -n -L7` on 10 GiB of synthetic text
with 64 KiB-period repeats (S2 "rep64k-10G"), one chunk (-m pinned
so max_chunk >= file size).  With N > 1 every rank runs its own 10 GiB stream on
its own GPU (chunks / files are independent units of the reference: N `mrzip`
processes), so scaling is weak and there is no data-path collective.

Prints ONE JSON line on rank 0 (see the contract in the task description):
value = whole-job GiB/s of input consumed; roofline = algorithmic bytes
(N + literal bytes + stream-0 bytes) over the measured time of the
dominant kernel (the sequencer), against the 8 TB/s HBM peak; cpu_baseline = the
oracle (bit-exact C restatement of the reference's single-threaded rzip stage)
timed on a bounded prefix of the same stream on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = 1 << 30
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(sample_bytes):
    """Times the oracle's rzip stage (1 core) on the first sample_bytes of the workload."""
    import subprocess
    from modern_rzip_amd import workloads
    from tests import _util
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    o = _util.Oracle(path)
    data = workloads.rep64k(sample_bytes // 65536, seed=1234)
    t0 = time.perf_counter()
    r = o.rzip_chunk(data, level=7)
    dt = time.perf_counter() - t0
    return {"value": round(len(data) / GIB / dt, 5), "unit": "GiB/s", "cores": 1, "kind": "port",
            "sample": f"first {len(data) / GIB:.2f} GiB of the same rep64k stream, oracle/liboracle.so "
                      f"mrzo_rzip_chunk (matcher + CRC32, no MD5), {dt:.1f} s",
            "matches": r["stats"]["matches"]}


def pmc_traffic(nseg):
    """HBM bytes per sequencer launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 +
    WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes); None if no summary is present."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["sequencer_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gib", type=float, default=10.0, help="chunk size per GPU in GiB (BASELINE config: 10)")
    ap.add_argument("--level", type=int, default=7)
    ap.add_argument("--cpu-sample-gib", type=float, default=2.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed GPU decode of the result")
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

    nper = int(args.gib * GIB) // 65536
    n = nper * 65536
    data = workloads.rep64k_device(nper, dev, seed=1234 + rank)  # every rank its own stream
    torch.cuda.synchronize()

    lib = m.load_library()
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
    nseg = 0
    for _ in range(args.steps):
        ctx.victim_round = 0  # each step = a fresh `mrzip` process on the same file
        res, _, _ = ctx.rzip_chunk(data, fetch=False)  # returns after the device work has completed
        t = ctx.timings()
        seq_ms += t.sequencer_ms
        tag_ms += t.tagscan_ms
        enc_ms += t.encode_ms
        crc_ms += t.crc_ms
        nseg = t.n_segments
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
            "config": {"workload": f"rep64k-{args.gib:g}G: mrzip -n -L{args.level}, {args.gib:g} GiB synthetic text with "
                                   f"64 KiB-period repeats (BASELINE configs[1]), one chunk per GPU, input resident in HBM",
                       "level": args.level, "chunk_bytes_per_gpu": n, "streams": world},
            "roofline": {"bound": "hbm", "kernel": "mrz_sequencer_kernel", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6),
                         "traffic": pmc_traffic(nseg),
                         "launches_per_step": nseg, "alg_bytes_per_step": alg_bytes,
                         "avg_launch_ms": round(seq_ms / steps / max(nseg, 1), 4)},
            "kernel_ms_per_step": {"sequencer": round(seq_ms / steps, 2), "tagscan": round(tag_ms / steps, 2),
                                   "encode": round(enc_ms / steps, 2), "crc32": round(crc_ms / steps, 2)},
            "result": {"s0_len": res.s0_len, "s1_len": res.s1_len, "matches": res.stats.matches,
                       "inserts": res.stats.inserts, "crc32": f"{res.crc32:08x}"},
        }
        if verify:
            out["verify"] = verify
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(int(min(args.cpu_sample_gib, args.gib) * GIB))
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
