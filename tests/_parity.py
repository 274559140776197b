"""Shared parity checks: libmrzgpu (real or emulated) against the oracle."""
import hashlib

import modern_rzip_amd as m


def check_chunk(lib, oracle, data, level=7, victim_round=0, table=False, seg_positions=None, cand_cap=None, xcd=None):
    """mrz_rzip_chunk vs mrzo_rzip_chunk: both streams, CRC, the seven counters,
    victim_round, final mask / hash_count (and optionally the whole table).  seg_positions / cand_cap shrink the
    front end's passes (several segments per chunk; lists that fill up)."""
    want = oracle.rzip_chunk(data, level=level, victim_round=victim_round, want_table=table)
    with m.RzipContext(level=level, max_chunk=len(data), lib=lib) as ctx:
        ctx.victim_round = victim_round
        if seg_positions:
            ctx.set_segment_positions(seg_positions)
        if cand_cap:
            ctx.set_candidate_capacity(cand_cap)
        if xcd is not None:
            ctx.set_xcd(xcd)
        res, s0, s1 = ctx.rzip_chunk(data)
        assert res.crc32 == want["crc"]
        assert res.stats.as_dict() == want["stats"]
        assert ctx.victim_round == want["victim_round"]
        assert res.min_mask == want["min_mask"]
        assert res.hash_count == want["hash_count"]
        assert s1 == want["s1"]
        assert s0 == want["s0"]
        if table:
            assert ctx.fetch_table() == want["table"]
        check_runzip(ctx, data, want["s0"], want["s1"])
    return want


def check_runzip(ctx, data, s0, s1, chunk_bytes_=None):
    """mrz_runzip_chunk on the ORACLE's streams gives the input back; computed == stored == zlib CRC."""
    import zlib
    cb = chunk_bytes_ or m.chunk_bytes(len(data), lib=ctx.lib)
    back, n, crc_calc, crc_stored = ctx.runzip_chunk(s0, s1, cb, len(data))
    assert n == len(data)
    assert back == data
    assert crc_calc == crc_stored == (zlib.crc32(data) & 0xFFFFFFFF)


def check_file(lib, oracle, data, level=7, **kw):
    """Host driver (rzip_fd mirror) vs oracle whole-file output, plus decode."""
    want, wstats, wmd5 = oracle.compress(data, level=level, **kw)
    got, gstats, gmd5 = m.rzip_buffer(data, level=level, lib=lib, **kw)
    assert gmd5 == wmd5 == hashlib.md5(data).digest()
    assert gstats.as_dict() == wstats
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()
    rc, back = oracle.decompress(got)
    assert rc == 0 and back == data
    assert m.runzip_buffer(want, lib=lib) == data  # mrzip -d of the ORACLE's archive through the GPU decoder
    return got


def check_stream(lib, oracle, data, to_stdout, ramsize, level=7):
    """The STDIN form of rzip_fd's chunk loop (mrz_rzip_stream_buffer) vs the oracle's restatement: archive bytes,
    counters, MD5; both decoders give the input back (an archive written to STDOUT in several chunks carries no size)."""
    want, wstats, wmd5, nch = oracle.compress_stream(data, to_stdout=to_stdout, level=level, ramsize=ramsize)
    got, gstats, gmd5 = m.rzip_stream_buffer(data, to_stdout=to_stdout, level=level, ramsize=ramsize, lib=lib)
    assert gmd5 == wmd5 == hashlib.md5(data).digest()
    assert gstats.as_dict() == wstats
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()
    size_field = int.from_bytes(got[6:14], "little")
    assert size_field == (len(data) if (not to_stdout or nch == 1) else 0)
    assert m.runzip_buffer(got, lib=lib) == data
    if size_field or not data:
        rc, back = oracle.decompress(got)
        assert rc == 0 and back == data
    return got, nch


def check_fd(lib, oracle, data, use_pipe, ramsize, tmp_path, level=7):
    """mrz_rzip_fd on real descriptors: a regular file (FILE form) or a pipe (STDIN form) into a file."""
    import os
    import threading
    outp = os.path.join(str(tmp_path), "out.mrz")
    fd_out = os.open(outp, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
    try:
        if use_pipe:
            r, w = os.pipe()

            def feed():
                with os.fdopen(w, "wb") as f:
                    for a in range(0, len(data), 37000):  # ragged writes: read() returns short counts
                        f.write(data[a:a + 37000])
            t = threading.Thread(target=feed)
            t.start()
            try:
                m.rzip_fd(r, fd_out, level=level, ramsize=ramsize, lib=lib)
            finally:
                t.join()
                os.close(r)
            want = oracle.compress_stream(data, to_stdout=False, level=level, ramsize=ramsize)[0]
        else:
            inp = os.path.join(str(tmp_path), "in.bin")
            with open(inp, "wb") as f:
                f.write(data)
            fd_in = os.open(inp, os.O_RDONLY)
            try:
                m.rzip_fd(fd_in, fd_out, level=level, ramsize=ramsize, lib=lib)
            finally:
                os.close(fd_in)
            want = oracle.compress(data, level=level, ramsize=ramsize)[0]
    finally:
        os.close(fd_out)
    with open(outp, "rb") as f:
        got = f.read()
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()
    return got


def ref_rs_decode(enc):
    """`rs-mrzip -d` on `enc` with the REFERENCE's own rsd32 / gather (oracle/_ref/librs_ref.so = rs-mrzip/reed-solomon.c
    compiled in place) and decode()'s loop (rs-mrzip/rs-mrzip.c:37-117) around them: (bytes, report)."""
    import ctypes
    import os
    ref = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                                   "librs_ref.so"))
    ROWS, BURST = 8176, 8176 * 255
    nb = len(enc) // BURST
    tail = enc[nb * BURST:]
    rows_out = bytearray()
    corrected = uncorrectable = 0
    for b in range(nb):
        tr = ctypes.create_string_buffer(enc[b * BURST:(b + 1) * BURST], BURST)
        ec = ctypes.create_string_buffer(BURST)
        ref.gather(tr, ec, ROWS, 255)
        eras = (ctypes.c_int * 32)()
        base = ctypes.addressof(ec)
        for i in range(ROWS):
            f = ref.rsd32(ctypes.c_void_p(base + i * 255), eras, 0)
            if f > 0:
                corrected += f
            elif f == -1:
                uncorrectable += 1
        raw = ec.raw
        for i in range(ROWS):
            rows_out += raw[i * 255:i * 255 + 223]
    rep = dict(corrected=corrected, uncorrectable=uncorrectable, checksum_ok=False, truncated=len(tail) != 68)
    out = bytes(rows_out)
    if len(tail) == 68:
        rep["checksum_ok"] = hashlib.blake2b(out).digest() == tail[:64]
        k_i, k_j = tail[64] | tail[65] << 8, tail[66] | tail[67] << 8
        if k_i < ROWS:
            out = out[:(nb - 1) * ROWS * 223 + k_i * 223 + k_j]
    return out, rep


def check_rs_decode(ctx, data, damage):
    """encode (GPU) -> damage -> decode (GPU) == the reference's decoder on the same damaged bytes."""
    enc = bytearray(ctx.rs_encode(data))
    for at, n, val in damage:
        for k in range(n):
            enc[at + k] ^= val
    enc = bytes(enc)
    got, rep = ctx.rs_decode(enc)
    want, wrep = ref_rs_decode(enc)
    assert rep == wrep, (rep, wrep)
    assert got == want
    return got, rep


def _blocks_of_archive(mrz):
    """(stream, payload) of every block of a -n archive in FILE order = the reference's flush order
    (src/stream.c:1199-1293): walks the chunks, finds every block header by following both chains."""
    out = []
    at = 20 + mrz[19]
    while True:
        cb, eof = mrz[at], mrz[at + 1]
        at += 2 + cb
        initial = at
        heads = {}  # file position of a block header -> stream
        end_max = initial + 2 * (1 + 3 * cb)
        for s in (0, 1):
            pos = initial + s * (1 + 3 * cb)
            first = True
            while True:
                c_len = int.from_bytes(mrz[pos + 1:pos + 1 + cb], "little")
                nxt = int.from_bytes(mrz[pos + 1 + 2 * cb:pos + 1 + 3 * cb], "little")
                if not first:
                    heads[pos] = s
                    end_max = max(end_max, pos + 1 + 3 * cb + c_len)
                first = False
                if not nxt:
                    break
                pos = initial + nxt
        for pos in sorted(heads):
            c_len = int.from_bytes(mrz[pos + 1:pos + 1 + cb], "little")
            out.append((heads[pos], mrz[pos + 1 + 3 * cb:pos + 1 + 3 * cb + c_len]))
        at = end_max
        if eof:
            return out


def check_pipeline(lib, oracle, data, level=7, **kw):
    """mrz_rzip_pipeline hands over exactly the blocks of the reference-identical archive, in file order."""
    okw = {k: v for k, v in kw.items() if k not in ("lz4_test", "threshold")}
    want, wstats, wmd5 = oracle.compress(data, level=level, **okw)
    got = []
    st, md5 = m.rzip_pipeline(data, lambda info, payload: got.append((info, payload)) and None, level=level, lib=lib, **kw)
    assert md5 == wmd5 and st.as_dict() == wstats
    blocks = _blocks_of_archive(want)
    assert [(i["stream"], p) for i, p in got] == blocks
    chunks = sorted({i["chunk_index"] for i, _ in got})
    assert chunks == list(range(len(chunks)))
    assert [i["eof"] for i, _ in got if i["first_of_chunk"]] == [0] * (len(chunks) - 1) + [1]
    if kw.get("lz4_test"):
        # the gate's verdict for every block = lz4_compresses of the oracle on the same bytes (blocks of >= 64 bytes)
        for i, p in got:
            assert i["lz4_verdict"] == (oracle.lz4_compresses(p, kw.get("threshold", 100)) if len(p) >= 64 else -1)
    else:
        assert all(i["lz4_verdict"] == -1 for i, _ in got)
    return got
