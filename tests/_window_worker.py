"""One rank of the window-sharded rzip stage (BASELINE configs[3], scaled), as a FRESH process:

    python tests/_window_worker.py RANK WORLD PORT gpu|emu SEGMENTS SEG_BYTES [backend]

Every rank keeps its byte range of ONE window in a shareable allocation of its own (mrz_window_part_create), all ranks
map all parts into one address range (mrz_window_map_create, descriptors passed over a Unix socket), run the front end
over the stretches of their range and ship compacted candidate records to rank 0, which runs the exact matcher and reads
the other ranks' bytes through the mapping.  Rank 0 compares streams, counters, CRC and victim_round with the oracle
and prints "window ok".  Launched by tests/test_gpu_parity.py (both ranks on GPU 0, gloo transport) and usable by hand.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    which, segments, seg_bytes = sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    backend = sys.argv[7] if len(sys.argv) > 7 else "gloo"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import modern_rzip_amd as m
    from modern_rzip_amd import shard, workloads
    from tests import _util
    dev = None
    if which == "gpu":
        lib = m.load_library()
        dev_index = rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dev = torch.device("cuda", dev_index)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        lib = m.load_library(os.path.join(ROOT, "tests", "emu", "libmrzgpu_emu.so"))
        dev_index = 0
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        win = workloads.stride_stream(segments, seg_bytes, seed=50)  # the same bytes on every rank (seeded generator)
        total = len(win)
        with m.RzipContext(lib=lib, max_chunk=total if rank == 0 else 0, device=dev_index) as ctx:
            wmap, part, ranges = shard.window_map(lambda off, size: win[off:off + size], total, rank, world, dist, ctx,
                                                  device=dev_index)
            try:
                off, size = ranges[rank]
                mine = (wmap.ptr + off, min(size + 48, total - off))  # own range + halo, read through the mapping
                out = shard.rzip_chunk_window(ctx, mine, off, total, rank, world, dist, victim_round=3,
                                              window=(wmap.ptr, total), device=dev, cap=1 << 20, ranges=ranges)
                if rank == 0:
                    served = ctx.window_served
                    res, s0, s1 = out
                    oracle = _util.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
                    want = oracle.rzip_chunk(win, victim_round=3)
                    assert (s0, s1) == (want["s0"], want["s1"]), "streams differ from the oracle's"
                    assert res.stats.as_dict() == want["stats"] and res.crc32 == want["crc"]
                    assert ctx.victim_round == want["victim_round"]
                    assert res.stats.matches >= 1 and served["remote"] >= 1 and served["n"] > served["remote"], served
                    print("window ok", total, served, flush=True)
            finally:
                dist.barrier()
                wmap.close()
                part.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
