"""N exact matchers on ONE GPU at once: N independent streams (files / chunks are independent units of the reference; a
chunk's matcher is one dependency chain, so streams are what scales with the CU count under bit-exactness), one mrz_ctx
and host thread each, ctx i on XCD i % 8 (mrz_set_xcd: its sequencer workgroups, and the deep engine's scan helpers, sit
on that XCD), the compare farm's helper workgroups split between them.
    python tools/probe_multictx.py SHAPE MIB N [N ...] [--check]        SHAPE: noise | tar | text
Prints one JSON line per N: aggregate GiB/s, speed-up over N = 1, and whether every stream's result (CRC, lengths and
sha256 of both streams) equals its single-ctx run; --check also compares every stream with the oracle."""
import hashlib
import json
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # read by the HIP runtime when it initialises: >= 2 x ctxs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv):
    check = "--check" in argv
    argv = [a for a in argv if a != "--check"]
    shape, mib, ns = argv[0], int(argv[1]), [int(x) for x in argv[2:]]
    import torch
    import modern_rzip_amd as m
    from modern_rzip_amd import workloads as w
    lib = m.load_library()
    nmax = max(ns)
    n = mib << 20
    gen = {"noise": lambda i: w.noise(n, seed=100 + i), "tar": lambda i: w.tar_like_fast(n, seed=200 + i, pool_bytes=8 << 20),
           "text": lambda i: w.zipf_text(n, seed=300 + i)}[shape]
    host = [gen(i) for i in range(nmax)]
    data = [torch.frombuffer(bytearray(h), dtype=torch.uint8).cuda() for h in host]
    ctxs = [m.RzipContext(level=7, max_chunk=n, lib=lib) for _ in range(nmax)]
    for i, c in enumerate(ctxs):
        c.set_xcd(i % 8)

    def sig(c, res):
        s0 = (res.s0_len, res.s1_len, res.crc32, res.stats.inserts, res.stats.matches, res.stats.tag_misses)
        return s0

    solo, lines = {}, []
    for N in [1] + [x for x in ns if x != 1]:
        for c in ctxs[:N]:
            c.set_farm_helpers(max(224 // N - 8, 8))
        res = [None] * N
        streams = [None] * N

        def work(i):
            ctxs[i].victim_round = 0
            r, s0, s1 = ctxs[i].rzip_chunk(data[i], fetch=True)
            res[i] = r
            streams[i] = (hashlib.sha256(s0).hexdigest(), hashlib.sha256(s1).hexdigest())
        best = None
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(i,)) for i in range(N)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        if N == 1:
            # every stream alone, on ctx 0: the reference results the concurrent runs are compared with
            for i in range(nmax):
                ctxs[0].victim_round = 0
                r, s0, s1 = ctxs[0].rzip_chunk(data[i], fetch=True)
                solo[i] = (sig(ctxs[0], r), (hashlib.sha256(s0).hexdigest(), hashlib.sha256(s1).hexdigest()))
            t1 = best
        same = all((sig(ctxs[i], res[i]), streams[i]) == solo[i] for i in range(N))
        line = {"shape": shape, "mib_per_stream": mib, "ctxs": N, "wall_s": round(best, 3),
                "agg_GiBps": round(N * n / (1 << 30) / best, 4), "speedup_over_1": round(N * t1 / best, 2),
                "identical_to_solo": same}
        lines.append(line)
        print(json.dumps(line), flush=True)
    if check:
        from tests import _util
        o = _util.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
        okc = True
        for i in range(nmax):
            want = o.rzip_chunk(host[i], level=7)
            okc = okc and solo[i][1] == (hashlib.sha256(want["s0"]).hexdigest(), hashlib.sha256(want["s1"]).hexdigest()) \
                and solo[i][0][2] == want["crc"]
        print(json.dumps({"oracle_equal": okc, "all_identical": all(l["identical_to_solo"] for l in lines)}), flush=True)
    for c in ctxs:
        c.close()
    return lines


def chunks_of_one_file(mib_per_chunk=32, n_chunks=8, n_ctxs=8):
    """One file of n_chunks chunks (noise: victim_round never moves, every prediction holds) through n_ctxs ctxs on n_ctxs
    XCDs of one process: the archive equals the oracle's; returns the JSON line."""
    import hashlib as hl
    import torch  # noqa: F401
    import modern_rzip_amd as m
    from modern_rzip_amd import shard, workloads as w
    from tests import _util
    lib = m.load_library()
    o = _util.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    max_chunk = mib_per_chunk << 20
    data = w.noise(n_chunks * max_chunk - 12345, seed=77)
    ramsize = max_chunk // 2 * 3
    out = {}
    for nc in (1, n_ctxs):
        ctxs = [m.RzipContext(level=7, max_chunk=max_chunk, lib=lib) for _ in range(nc)]
        for i, c in enumerate(ctxs):
            c.set_xcd(i % 8)
            c.set_farm_helpers(max(224 // nc - 8, 8))
        t0 = time.perf_counter()
        parts, reruns = shard.rzip_file_chunks_on_ctxs(data, max_chunk, ctxs)
        dt = time.perf_counter() - t0
        for c in ctxs:
            c.close()
        arch = o.frame(len(data), parts, hl.md5(data).digest(), ramsize=ramsize)
        out[nc] = (dt, hl.sha256(arch).hexdigest(), reruns)
    want, _, _ = o.compress(data, ramsize=ramsize)
    line = {"file_mib": len(data) >> 20, "chunks": n_chunks, "ctxs": n_ctxs, "wall_s_1ctx": round(out[1][0], 3),
            "wall_s": round(out[n_ctxs][0], 3), "speedup": round(out[1][0] / out[n_ctxs][0], 2), "reruns": out[n_ctxs][2],
            "archive_equals_oracle": out[n_ctxs][1] == hl.sha256(want).hexdigest() == out[1][1]}
    print(json.dumps(line), flush=True)
    return line


if __name__ == "__main__":
    if sys.argv[1] == "chunks":
        chunks_of_one_file(*[int(x) for x in sys.argv[2:]])
    else:
        main(sys.argv[1:])
