"""Co-residency on ONE GPU (BASELINE configs[4]: BLAKE2b + rs-mrzip beside the rzip stage): the rzip stage of one
stream on its ctx while a second host thread hashes with the BLAKE2b kernels and a third encodes rs-mrzip bursts, each
on its own ctx / streams.  Prints solo and concurrent times and checks that every output is the same as solo.
    python tools/probe_coresident.py [rzip MiB] [blake2b messages x KiB] [rs MiB]"""
import sys, os, time, json, threading, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modern_rzip_amd as m
from modern_rzip_amd import workloads as w


def main(rz_mib=512, b2_msgs=2048, b2_kib=256, rs_mib=256):
    lib = m.load_library()
    stream = w.rep64k_device(rz_mib * 16, "cuda")
    msgs = [w.noise(b2_kib << 10, seed=i) for i in range(8)] * (b2_msgs // 8)
    rs_in = w.noise(rs_mib << 20, seed=77)
    c1 = m.RzipContext(max_chunk=stream.numel(), lib=lib)
    c2 = m.RzipContext(lib=lib)
    c3 = m.RzipContext(lib=lib)
    out = {}

    def rzip():
        c1.victim_round = 0
        t0 = time.perf_counter()
        res, s0, s1 = c1.rzip_chunk(stream)
        out["rzip"] = (time.perf_counter() - t0, hashlib.sha256(s0 + s1).hexdigest())

    def blake():
        t0 = time.perf_counter()
        d = c2.blake2b_batch(msgs)
        one = c2.blake2b(msgs[0] * 8)  # the streaming triple as well
        out["blake2b"] = (time.perf_counter() - t0, hashlib.sha256(b"".join(d) + one).hexdigest())

    def rs():
        t0 = time.perf_counter()
        enc = c3.rs_encode(rs_in)
        out["rs"] = (time.perf_counter() - t0, hashlib.sha256(enc).hexdigest())

    for f in (rzip, blake, rs):  # warm
        f()
    solo = {}
    for name, f in (("rzip", rzip), ("blake2b", blake), ("rs", rs)):
        f()
        solo[name] = out[name]
    th = [threading.Thread(target=f) for f in (rzip, blake, rs)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    same = all(out[k][1] == solo[k][1] for k in solo)
    assert hashlib.sha256(msgs[0]).digest() is not None
    assert c2.blake2b(msgs[0]) == hashlib.blake2b(msgs[0]).digest()
    line = {"rzip_MiB": rz_mib, "blake2b": f"{b2_msgs} x {b2_kib} KiB", "rs_MiB": rs_mib,
            "solo_s": {k: round(v[0], 4) for k, v in solo.items()}, "together_s": {k: round(out[k][0], 4) for k in solo},
            "together_wall_s": round(wall, 4), "sum_of_solo_s": round(sum(v[0] for v in solo.values()), 4),
            "slowdown": {k: round(out[k][0] / solo[k][0], 3) for k in solo}, "outputs_identical": same}
    print(json.dumps(line), flush=True)
    for c in (c1, c2, c3):
        c.close()
    return line


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    main(*a)
