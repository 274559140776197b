"""CPU tier: the N>1 path over gloo with world_size 2 (the emulated library does
the per-rank compute; what is under test is the sharding / chaining / gathering)."""
import hashlib
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import modern_rzip_amd as m
    from modern_rzip_amd import shard
    from tests import _util
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = m.load_library(os.path.join(ROOT, "tests", "emu", "libmrzgpu_emu.so"))
        oracle = _util.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))

        # (1) independent streams: no collective on the data path
        streams = [_util.rep64k(3, seed=40 + i, period=2048) for i in range(4)]
        mine = shard.streams_of_rank(len(streams), rank, world)
        digests = {}
        for i in mine:
            mrz, _, _ = m.rzip_buffer(streams[i], lib=lib)
            digests[i] = hashlib.sha256(mrz).hexdigest()
        gathered = [None] * world
        dist.all_gather_object(gathered, digests)
        if rank == 0:
            merged = {}
            for g in gathered:
                merged.update(g)
            want = {i: hashlib.sha256(oracle.compress(s)[0]).hexdigest() for i, s in enumerate(streams)}
            assert merged == want

        # (2) one file, chunks dealt round-robin, victim_round chained rank to rank
        data = _util.rep64k(10, seed=77, period=2048)
        max_chunk = 8192  # page-rounded as src/rzip.c:888 requires: 3 chunks
        ramsize = max_chunk // 2 * 3
        ctx = m.RzipContext(lib=lib, max_chunk=max_chunk)

        def run(chunk, vr_in):
            ctx.victim_round = vr_in
            res, s0, s1 = ctx.rzip_chunk(chunk)
            return s0, s1, ctx.victim_round

        parts = shard.rzip_file_chunk_chain(data, max_chunk, rank, world, run, dist)
        spec = shard.rzip_file_chunk_chain_speculative(data, max_chunk, rank, world, run, dist)
        noise = _util.xorshift_noise(3 * max_chunk, seed=12)  # no evictions: victim_round never moves
        spec_noise = shard.rzip_file_chunk_chain_speculative(noise, max_chunk, rank, world, run, dist)
        ctx.close()
        if rank == 0:
            assert spec[0] == parts  # the same streams whether or not the predictions held
            assert spec_noise[1] == 0  # ... and on a stream without evictions no chunk had to be run twice
            want_n, _, _ = oracle.compress(noise, ramsize=ramsize)
            assert oracle.frame(len(noise), spec_noise[0], hashlib.md5(noise).digest(), ramsize=ramsize) == want_n
            assert len(parts) == 3
            md5 = hashlib.md5(data).digest()
            got = oracle.frame(len(data), parts, md5, ramsize=ramsize)
            want, _, _ = oracle.compress(data, ramsize=ramsize)
            assert oracle.plan(len(data), ramsize=ramsize)[0] == max_chunk
            assert got == want
            rc, back = oracle.decompress(got)
            assert rc == 0 and back == data
        # (3) one chunk, CRC of byte ranges combined on rank 0 (range-sharded front-end); the per-range CRC is the
        # library's kernel
        import zlib
        big = _util.zipf_text(300000, seed=5) + _util.xorshift_noise(70001, seed=6)
        with m.RzipContext(lib=lib) as c2:
            crc = shard.chunk_crc_sharded(big, rank, world, c2.crc32, dist)
        if rank == 0:
            assert crc == zlib.crc32(big)
        # (4) ONE window over the two ranks (BASELINE configs[3], scaled): every rank runs the front end over the
        # stretches in its byte range (compacted candidate records), rank 0 runs the exact matcher over them; streams
        # identical to the single-process oracle
        from modern_rzip_amd import workloads
        win = workloads.stride_stream(8, 96 * 1024, seed=50)  # noise, every 4th segment repeats an earlier one
        ranges = shard.window_ranges(len(win), world)
        off, size = ranges[rank]
        mine = win[off:off + size + 48]  # own range + halo
        with m.RzipContext(lib=lib, max_chunk=len(win)) as c3:
            c3.set_segment_positions(64 * 1024)  # several stretches per range
            out = shard.rzip_chunk_window(c3, mine, off, len(win), rank, world, dist, victim_round=5, cap=20000)
            crc = shard.chunk_crc_sharded(win, rank, world, c3.crc32, dist)
            if rank == 0:
                assert c3.window_served["remote"] >= 2 and c3.window_served["n"] > c3.window_served["remote"]
        if rank == 0:
            res, s0, s1 = out
            want = oracle.rzip_chunk(win, victim_round=5)
            assert (s0, s1) == (want["s0"], want["s1"])
            assert res.stats.as_dict() == want["stats"] and res.crc32 == want["crc"] == crc
            assert res.stats.matches >= 2
        q.put((rank, "ok"))
    except BaseException as e:  # surface the failure in the parent
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_gloo(emu_lib, oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_provider_path_when_the_wide_engine_hands_over_to_the_deep_one(emu_lib, oracle, monkeypatch):
    """Candidate provider + engine change inside a stretch: at level 1 the mask tightens quickly; the wide engine ends
    its launch where the mask reaches MRZ_DEEP_MIN_BITS and takes scan_next back to its position -- the host (which
    drives the geometry in this mode) must ask for the next stretch from THERE, not from where the last one ended."""
    import modern_rzip_amd as m
    from tests import _util
    monkeypatch.setenv("MRZ_DEEP_MIN_BITS", "5")
    data = _util.xorshift_noise(3400000, seed=3)
    want = oracle.rzip_chunk(data, level=1)
    assert want["min_mask"] >= 31
    with m.RzipContext(lib=emu_lib, level=1, max_chunk=len(data)) as ctx, m.RzipContext(lib=emu_lib, level=1, max_chunk=len(data)) as other:
        def provider(seg_start, span, min_mask, p_done, cap):
            return other.window_scan(data, 0, len(data), seg_start, span, min_mask, p_done, cap=cap)
        ctx.set_cand_provider(provider)
        res, s0, s1 = ctx.rzip_chunk(data)
        ctx.set_cand_provider(None)
        assert (s0, s1) == (want["s0"], want["s1"]) and res.stats.as_dict() == want["stats"]
        assert ctx.timings().n_deep >= 1 and ctx.timings().n_deep < ctx.timings().n_segments


def run_window_workers(which, segments, seg_bytes, timeout=600, backend="gloo"):
    """Two FRESH processes (started before anything in them touches a device), one rank each, over the window-sharded
    path with peer-mapped window memory: tests/_window_worker.py.  Returns rank 0's output."""
    import subprocess
    port = 29500 + (os.getpid() * 7 + 13) % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_window_worker.py"), str(r), "2", str(port), which,
                               str(segments), str(seg_bytes), backend], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs) and "window ok" in outs[0], "\n----\n".join(o[-3000:] for o in outs)
    return outs[0]


def test_window_over_two_processes_with_mapped_ranges(emu_lib, oracle):
    """BASELINE configs[3], scaled, on the emulator: every rank's byte range in a shareable allocation of its own, all
    parts mapped back to back in both processes (the emulator's VMM subset is memfd + mmap), candidates shipped to rank 0,
    whose matcher reads the other rank's bytes through the mapping; equal to the oracle."""
    run_window_workers("emu", 8, 128 * 1024)


def test_chunk_splitting_rules():
    from modern_rzip_amd import shard
    assert shard.split_chunks(0, 4096) == [(0, 0)]
    assert shard.split_chunks(10, 4096) == [(0, 10)]
    assert shard.split_chunks(8192, 4096) == [(0, 4096), (4096, 4096)]
    assert shard.split_chunks(9000, 4096) == [(0, 4096), (4096, 4096), (8192, 808)]
    assert shard.streams_of_rank(5, 1, 2) == [1, 3]


def test_crc32_combine_and_ranges():
    import random
    import zlib
    from modern_rzip_amd import shard
    rng = random.Random(4)
    for _ in range(50):
        a = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 3000)))
        b = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 3000)))
        assert shard.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    for total, world in ((0, 2), (1, 8), (4096, 2), (100000, 3), (10 << 30, 8)):
        r = shard.byte_ranges(total, world)
        assert len(r) == world and sum(n for _, n in r) == total
        assert all(o % 4096 == 0 or n == 0 for o, n in r)
        assert all(r[i][0] + r[i][1] == r[i + 1][0] or r[i + 1][1] == 0 for i in range(world - 1))


def test_window_scan_of_ranges_equals_local_scan(emu_lib, oracle):
    """The candidate provider path on one process: stretches scanned from byte ranges (with halo) by another context
    give the same streams; the matcher asks for one stretch after another, each from where the last pass ended (or
    further on, at the tile of the matcher's position, when an emitted match has swallowed what lies between)."""
    import modern_rzip_amd as m
    from modern_rzip_amd import shard, workloads
    from tests import _util
    data = _util.zipf_text(150000, seed=9) + workloads.stride_stream(4, 40000, seed=3)
    want = oracle.rzip_chunk(data)
    with m.RzipContext(lib=emu_lib, max_chunk=len(data)) as ctx, m.RzipContext(lib=emu_lib, max_chunk=len(data)) as other:
        ranges = shard.window_ranges(len(data), 3)
        assert sum(n for _, n in ranges) == len(data) and all(o % 4096 == 0 for o, _ in ranges)
        calls = []

        def provider(seg_start, span, min_mask, p_done, cap):
            r = max(i for i in range(3) if ranges[i][0] <= seg_start and ranges[i][1])
            off, size = ranges[r]
            end = off + size if r < 2 else -(-len(data) // 4096) * 4096
            span = max(4096, min(span, end - seg_start))
            out = other.window_scan(data[off:off + size + 48], off, len(data), seg_start, span, min_mask, p_done, cap=cap)
            calls.append((seg_start, span, out[3], out[4]))
            return out

        ctx.set_segment_positions(32 * 1024)
        ctx.set_candidate_capacity(6000)
        ctx.set_cand_provider(provider)
        res, s0, s1 = ctx.rzip_chunk(data)
        ctx.set_cand_provider(None)
        assert (s0, s1) == (want["s0"], want["s1"]) and res.stats.as_dict() == want["stats"]
        assert len(calls) > 10 and all(nc <= 6000 for _, _, _, nc in calls)
        assert all(calls[i][2] <= calls[i + 1][0] for i in range(len(calls) - 1))
        assert sum(calls[i][2] == calls[i + 1][0] for i in range(len(calls) - 1)) > len(calls) // 2
