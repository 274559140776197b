"""Shared parity checks: libmrzgpu (real or emulated) against the oracle."""
import hashlib

import modern_rzip_amd as m


def check_chunk(lib, oracle, data, level=7, victim_round=0, table=False):
    """mrz_rzip_chunk vs mrzo_rzip_chunk: both streams, CRC, the seven counters,
    victim_round, final mask / hash_count (and optionally the whole table)."""
    want = oracle.rzip_chunk(data, level=level, victim_round=victim_round, want_table=table)
    with m.RzipContext(level=level, max_chunk=len(data), lib=lib) as ctx:
        ctx.victim_round = victim_round
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
    want, wstats, wmd5 = oracle.compress(data, level=level, **kw)
    got = []
    st, md5 = m.rzip_pipeline(data, lambda info, payload: got.append((info, payload)) and None, level=level, lib=lib, **kw)
    assert md5 == wmd5 and st.as_dict() == wstats
    blocks = _blocks_of_archive(want)
    assert [(i["stream"], p) for i, p in got] == blocks
    chunks = sorted({i["chunk_index"] for i, _ in got})
    assert chunks == list(range(len(chunks)))
    assert [i["eof"] for i, _ in got if i["first_of_chunk"]] == [0] * (len(chunks) - 1) + [1]
    return got
