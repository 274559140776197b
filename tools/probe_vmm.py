"""Where does the window mode's time go?  One process, one GPU: the same stride stream sequenced (a) from an ordinary
device tensor, (b) from a shareable allocation (mrz_window_part_create: HIP VMM memory), (c) through the candidate
provider (front end run by a second ctx, stretch by stretch) from the VMM memory.  python tools/probe_vmm.py [MiB]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(mib=256):
    import torch
    import modern_rzip_amd as m
    from modern_rzip_amd import shard, workloads
    lib = m.load_library()
    dev = torch.device("cuda", 0)
    g = m.window_granularity(0, lib)
    seg = max((mib << 20) // 16 // g, 1) * g
    total = 16 * seg
    win = workloads.stride_stream_device(16, seg, dev, seed=99)
    out = {"bytes": total}
    with m.RzipContext(lib=lib, max_chunk=total) as ctx:
        def run(name, chunk, **kw):
            best = None
            for _ in range(2):
                ctx.victim_round = 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if kw:
                    res = shard.rzip_chunk_window(ctx, kw["mine"], 0, total, 0, 1, None, window=chunk, device=dev, cap=8 << 20)[0]
                else:
                    res, _, _ = ctx.rzip_chunk(chunk, fetch=False)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[name] = {"s": round(best, 3), "MiBps": round(total / best / 2**20, 1), "crc": f"{res.crc32:08x}", "matches": res.stats.matches}
        run("tensor", win)
        part = m.WindowPart(total, device=0, lib=lib)
        ctx.copy_to(part.ptr, win)
        run("vmm", (part.ptr, total))
        run("vmm_provider", (part.ptr, total), mine=(part.ptr, total))
        run("tensor_provider", win, mine=win)
        part.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
