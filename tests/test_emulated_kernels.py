"""CPU tier: the gfx950 kernel SOURCES compiled by g++ against the wave64
emulator (tests/emu) and driven through the same C ABI, checked against the
oracle.  This exercises the kernels' logic (ballot/shuffle algorithms, indexing,
bounds) where sanitizers and debuggers work; the GPU tier repeats the same
checks on the real library at larger sizes."""
import hashlib
import json
import os
import zlib

import pytest

import modern_rzip_amd as m
from tests import _parity, _util

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "vectors.json")) as f:
    GOLD = json.load(f)


@pytest.fixture(scope="module")
def inputs():
    return _util.golden_inputs()


@pytest.mark.parametrize("name", ["empty", "range30", "a1000", "range256x64", "seed42x64"])
def test_golden_files_through_host_driver(emu_lib, oracle, inputs, name):
    got = _parity.check_file(emu_lib, oracle, inputs[name])
    assert hashlib.sha256(got).hexdigest() == GOLD["files"][name]["sha256"]


@pytest.mark.parametrize("n", [0, 1, 30, 31, 32, 33, 62, 63, 64, 95, 4095, 4096, 4097, 4127])
def test_ragged_sizes(emu_lib, oracle, n):
    _parity.check_chunk(emu_lib, oracle, (_util.zipf_text(max(n, 1), seed=n + 1) * 2)[:n])


def test_periodic_text_levels(emu_lib, oracle):
    data = _util.rep64k(3, seed=5, period=8192)
    for level in (1, 4, 6, 7, 9):
        _parity.check_chunk(emu_lib, oracle, data, level=level)


def test_victim_round_chain(emu_lib, oracle):
    data = _util.rep64k(24, seed=9, period=2048)
    for vr in (0, 3, 15):
        _parity.check_chunk(emu_lib, oracle, data, victim_round=vr)


def test_noise_and_table_state(emu_lib, oracle):
    _parity.check_chunk(emu_lib, oracle, _util.xorshift_noise(40000, seed=3), table=True)


def test_backward_extension_near_start(emu_lib, oracle):
    # a repeat whose backward extension runs into offset 0 and into last_match
    blk = _util.xorshift_noise(700, seed=8)
    _parity.check_chunk(emu_lib, oracle, blk + blk + b"xyz" + blk[5:] + blk)


def test_long_match_pieces(emu_lib, oracle):
    # one match longer than 0xFFFF: put_match splits it (src/rzip.c:183)
    blk = _util.xorshift_noise(70000, seed=12)
    _parity.check_chunk(emu_lib, oracle, blk + blk)


def test_batch_engine_culling_and_mask_promotion(emu_lib, oracle):
    """Level 1 (2 MiB table): the table fills, the first cull switches the insert mask,
    sweeps complete and minimum_tag_mask is promoted -- all inside batch mode."""
    want = _parity.check_chunk(emu_lib, oracle, _util.xorshift_noise(3 << 20, seed=3), level=1, table=True)
    assert want["min_mask"] > 15 and want["stats"]["inserts"] > want["hash_count"]
    _parity.check_chunk(emu_lib, oracle, _util.tar_like(1 << 20, seed=4), level=1, table=True)


@pytest.mark.parametrize("wgs", [2, 3])
def test_sequencer_workgroups_take_turns(emu_lib, oracle, wgs, monkeypatch):
    """Several sequencer workgroups (the emulator runs them interleaved): batches prepared ahead are re-validated
    against the write log at their turn -- stale lanes, lanes dropped by a match emitted meanwhile, new epochs after
    a mask promotion and continuation inside a window all occur on these inputs; results stay the reference's."""
    monkeypatch.setenv("MRZ_EMU_CORESIDENT", "1")
    monkeypatch.setenv("MRZ_SEQ_WGS", str(wgs))
    _parity.check_chunk(emu_lib, oracle, _util.zipf_text(160000, seed=4), table=True)
    _parity.check_chunk(emu_lib, oracle, _util.xorshift_noise(100000, seed=3), table=True)
    _parity.check_chunk(emu_lib, oracle, _util.rep64k(24, seed=9, period=2048), victim_round=3)
    blk = _util.xorshift_noise(70000, seed=12)
    _parity.check_chunk(emu_lib, oracle, blk + blk)
    _parity.check_chunk(emu_lib, oracle, _util.tar_like(1 << 19, seed=4), level=1, table=True)


def _deep_mix():
    """Level-1 input (2 MiB table) that takes the deep engine through its states: noise until the table is nearly full
    (no lane is scanned before the first cull: 2.6 MB leave hash_count at 83.7 k of the 87.4 k limit -- the 1.7 MB this
    test used until round 3 never got there, and the bulk rounds never ran in it), then text, during which the culling
    begins (tag-equal entries, real matches: record replay and cooperative path between bulk commits, culls in every
    round, lanes of one run settled inside a round), a repeat of earlier noise (one long match, lanes dropped behind
    it) and noise again; the sweep has wrapped once by the end (min_mask 31)."""
    noise = _util.xorshift_noise(2600000, seed=3)
    return noise + _util.zipf_text(420000, seed=4) + noise[300000:420000] + _util.xorshift_noise(150000, seed=8)


def test_deep_engine_alone(emu_lib, oracle, monkeypatch):
    """MRZ_SEQ_ENGINE=deep: every segment on the deep engine (run scans by all waves, conflicts from the lanes' planned
    writes, bulk commit with prefix sums, stale lanes scanned again, record replay / cooperative path for real
    matches); the whole table is compared too."""
    monkeypatch.setenv("MRZ_SEQ_ENGINE", "deep")
    want = _parity.check_chunk(emu_lib, oracle, _deep_mix(), level=1, table=True, victim_round=1)
    assert want["stats"]["inserts"] > want["hash_count"] and want["stats"]["matches"] >= 2 and want["stats"]["tag_hits"] > 10
    assert want["min_mask"] == 31  # the cull sweep has been through the table: the scanned (bulk) rounds have run
    _parity.check_chunk(emu_lib, oracle, _util.rep64k(40, seed=9, period=997), level=2, table=True)
    blk = _util.xorshift_noise(70000, seed=12)
    _parity.check_chunk(emu_lib, oracle, blk + blk + b"xyz" + blk[5:])
    _parity.check_chunk(emu_lib, oracle, _util.zipf_text(30000, seed=4), seg_positions=4096, cand_cap=4096)


def test_deep_engine_with_scan_helpers(emu_lib, oracle, monkeypatch):
    """The same with two scan helper workgroups beside the committer (the emulator runs them interleaved): batches are
    dealt out through device memory, the helpers' records are copied back; and the per-segment choice (wide engine first,
    ending its launch where the mask reaches MRZ_DEEP_MIN_BITS, then the deep engine)."""
    monkeypatch.setenv("MRZ_EMU_CORESIDENT", "1")
    monkeypatch.setenv("MRZ_DEEP_SCANNERS", "2")
    monkeypatch.setenv("MRZ_DEEP_MIN_BITS", "5")
    with m.RzipContext(level=1, max_chunk=4 << 20, lib=emu_lib) as ctx:
        noise = _util.xorshift_noise(3300000, seed=3)
        want = oracle.rzip_chunk(noise, level=1)
        res, s0, s1 = ctx.rzip_chunk(noise)
        assert (s0, s1) == (want["s0"], want["s1"]) and res.stats.as_dict() == want["stats"] and res.min_mask == 31
        t = ctx.timings()
        assert 1 <= t.n_deep < t.n_segments


@pytest.mark.parametrize("engine", ["wide", "narrow"])
def test_front_end_passes_and_full_lists(emu_lib, oracle, engine, monkeypatch):
    """The candidate list of the front end: several passes per chunk (small spans), passes cut short because the list is
    full (small capacity, down to one tile), a run of one byte whose tag passes the mask at EVERY position (all
    positions are candidates), matches that jump over whole passes; each engine alone."""
    monkeypatch.setenv("MRZ_SEQ_ENGINE", engine)
    text = _util.zipf_text(70000, seed=4)
    _parity.check_chunk(emu_lib, oracle, text, seg_positions=4096)
    _parity.check_chunk(emu_lib, oracle, text, seg_positions=16384, cand_cap=4096, table=True)
    noise = _util.xorshift_noise(50000, seed=3)
    _parity.check_chunk(emu_lib, oracle, noise, seg_positions=8192, cand_cap=5000)
    blk = _util.xorshift_noise(30000, seed=12)
    _parity.check_chunk(emu_lib, oracle, blk + blk + text[:9000] + blk, seg_positions=4096, victim_round=2)
    # runs of one byte: the tag of a run is hash_index[b] (31 equal terms); for half the byte values it passes the 1-bit
    # mask, and then EVERY position of the run is in the list (each pass is cut after one tile); the matcher swallows
    # the run as one match
    swallowed = 0
    for b in (0, 1, 2, 3):
        run = bytes([b]) * 21000 + text[:3000] + bytes([b]) * 6000
        # (capacity 8192 -> passes of 3 tiles under the 1-bit mask: two tiles of a run fill the list, the third is cut off)
        swallowed += _parity.check_chunk(emu_lib, oracle, run, seg_positions=16384, cand_cap=8192)["stats"]["matches"] >= 2
    assert swallowed >= 1
    _parity.check_chunk(emu_lib, oracle, _util.rep64k(5, seed=5, period=8192), level=4, seg_positions=8192, xcd=5)


def test_crc32_kernel(emu_lib):
    with m.RzipContext(lib=emu_lib) as ctx:
        for n in (0, 1, 15, 16, 17, 1000, 65535, 65536, 65537, 3 * 65536 + 77):
            d = _util.xorshift_noise(n, seed=n + 5)
            assert ctx.crc32(d) == zlib.crc32(d), n


def test_blake2b_kernels(emu_lib, oracle):
    with m.RzipContext(lib=emu_lib) as ctx:
        msgs = [_util.xorshift_noise(n, seed=n + 1) for n in (0, 1, 127, 128, 129, 256, 257, 1000)]
        for outlen in (64, 32):
            got = ctx.blake2b_batch(msgs, outlen)
            assert got == [hashlib.blake2b(x, digest_size=outlen).digest() for x in msgs]
        d = _util.xorshift_noise(223 * 9 + 40, seed=2)
        rows = [d[i:i + 223] for i in range(0, len(d), 223)]
        assert ctx.blake2b(d, pieces=rows) == hashlib.blake2b(d).digest()
        assert ctx.blake2b(b"") == hashlib.blake2b(b"").digest()
        assert ctx.blake2b(d, pieces=[d[:128], d[128:256], d[256:]]) == oracle.blake2b(d)


def test_lz4_kernel_sizes(emu_lib, oracle):
    with open(os.path.join(HERE, "golden", "lz4_sizes.json")) as f:
        gold = json.load(f)["sizes"]
    from tests.golden import make_lz4_golden
    small = [(k, d) for k, d in make_lz4_golden.cases() if len(d) <= 70000]
    with m.RzipContext(lib=emu_lib) as ctx:
        got = ctx.lz4_sizes([d for _, d in small])
        for (name, d), g in zip(small, got):
            assert g == gold[name] == oracle.lz4_size(d), name
        extra = [bytes(n) for n in (1, 12, 13, 14, 64)] + [_util.zipf_text(n, seed=n) for n in (13, 20, 100, 3000)]
        assert ctx.lz4_sizes(extra) == [oracle.lz4_size(d) for d in extra]


def test_lz4_gate(emu_lib, oracle):
    blocks = [_util.zipf_text(30000, seed=4), _util.xorshift_noise(30000, seed=4), bytes(5000),
              _util.xorshift_noise(20000, seed=5) + _util.zipf_text(10000, seed=6)]
    with m.RzipContext(lib=emu_lib) as ctx:
        for thr in (100, 60, 10):
            assert ctx.lz4_compresses(blocks, thr) == [oracle.lz4_compresses(b, thr) for b in blocks]
        assert ctx.lz4_compresses(blocks[0]) == oracle.lz4_compresses(blocks[0], 100)


def test_two_chunks_through_host_driver(emu_lib, oracle):
    data = _util.rep64k(6, seed=13, period=4096)
    # ramsize chosen so that max_chunk (= ramsize/3*2, page-rounded) splits the file in two
    _parity.check_file(emu_lib, oracle, data, ramsize=3 * 16384 // 2 + 3000)


def test_rs_encoder(emu_lib, oracle):
    with m.RzipContext(lib=emu_lib) as ctx:
        for n in (0, 1, 222, 223, 224, 100000):
            d = _util.xorshift_noise(n, seed=n + 3)
            assert ctx.rs_encode(d) == oracle.rs_encode(d), n


def test_rs_decoder_against_the_reference(emu_lib):
    """rsd32 / gather / decode() (rs-mrzip/reed-solomon.c:143-333, rs-mrzip.c:37-117): undamaged input, bursts of errors
    that the interleave spreads over many codewords, a codeword with exactly 16 and with more than 16 errors."""
    data = _util.xorshift_noise(300000, seed=5)
    with m.RzipContext(lib=emu_lib) as ctx:
        got, rep = _parity.check_rs_decode(ctx, data, [])
        assert got == data and rep == dict(corrected=0, uncorrectable=0, checksum_ok=True, truncated=False)
        # a burst of 4 * 8176 damaged bytes = 4 errors in every codeword; 16 errors in row 7; 17 in row 9
        dmg = [(1000, 4 * 8176, 0x5a)] + [((20 + 3 * k) * 8176 + 7, 1, 0x11) for k in range(16)]
        dmg += [((100 + 2 * k) * 8176 + 9, 1, 0x80) for k in range(17)]
        got, rep = _parity.check_rs_decode(ctx, data, dmg)
        assert rep["uncorrectable"] >= 1 and rep["corrected"] >= 4 * 8170 and not rep["checksum_ok"]
        got, rep = _parity.check_rs_decode(ctx, data, dmg[1:17])  # 16 errors in one codeword: the limit
        assert got == data and rep["checksum_ok"] and rep == dict(corrected=16, uncorrectable=0, checksum_ok=True,
                                                                   truncated=False)
        # trailer damaged / missing
        _parity.check_rs_decode(ctx, data, [(2084880 + 3, 1, 1)])
        enc = ctx.rs_encode(data)
        got, rep = ctx.rs_decode(enc[:-68])
        assert rep["truncated"] and got[:len(data)] == data and len(got) == 8176 * 223


def test_runzip_overlapping_and_corrupt_streams(emu_lib, oracle):
    # hand-made streams: literal "abc", match len 10 dist 3 (the 3 history bytes repeat, src/runzip.c:182-199),
    # match len 4 dist 13 (plain copy), terminator + CRC
    want = b"abc" + b"abcabcabca" + b"abca"
    crc = zlib.crc32(want) & 0xFFFFFFFF
    s0 = (bytes([0, 3, 0]) + bytes([1, 10, 0, 3]) + bytes([1, 4, 0, 13]) + bytes([0, 0, 0]) + crc.to_bytes(4, "big"))
    with m.RzipContext(level=7, max_chunk=64, lib=emu_lib) as ctx:
        back, n, cc, cs = ctx.runzip_chunk(s0, b"abc", 1, 64)
        assert back == want and cc == cs == crc
        # distance beyond the history / zero distance / empty match / literals beyond stream 1 / no terminator
        for bad in (bytes([0, 3, 0, 1, 4, 0, 4, 0, 0, 0]) + bytes(4),
                    bytes([0, 3, 0, 1, 4, 0, 0, 0, 0, 0]) + bytes(4),
                    bytes([0, 3, 0, 1, 0, 0, 1, 0, 0, 0]) + bytes(4),
                    bytes([0, 9, 0, 0, 0, 0]) + bytes(4),
                    bytes([0, 3, 0, 1, 4, 0, 3])):
            with pytest.raises(m.MrzError):
                ctx.runzip_chunk(bad, b"abc", 1, 64)


def test_runzip_many_tiles(emu_lib, oracle):
    # output of several 32 KiB decode tiles, stream 0 of several 1 KiB parse tiles, matches reaching back across tiles
    words = [_util.xorshift_noise(48, seed=100 + i) for i in range(40)]
    order = _util.xorshift_noise(3000, seed=7)
    data = _util.rep64k(5, seed=3, period=40000) + b"".join(words[b % 40] + bytes([b]) for b in order)
    want = oracle.rzip_chunk(data, level=7)
    assert len(want["s0"]) > 3000
    with m.RzipContext(level=7, max_chunk=len(data), lib=emu_lib) as ctx:
        _parity.check_runzip(ctx, data, want["s0"], want["s1"])


def test_stride_repeats(emu_lib, oracle):
    # BASELINE configs[3] shape in miniature: noise segments, planted repeats 1, 3 and 7 segments back
    from modern_rzip_amd import workloads as w
    data = w.stride_stream(12, 4096, copy_bytes=1500)
    want = _parity.check_chunk(emu_lib, oracle, data)
    assert want["stats"]["matches"] >= 2


def test_backend_handoff_pipeline(emu_lib, oracle):
    # three chunks; blocks must come in the reference's flush order with its block sizes
    data = _util.rep64k(10, seed=13, period=4096) + _util.zipf_text(9000, seed=2)
    got = _parity.check_pipeline(emu_lib, oracle, data, ramsize=3 * 16384 // 2 + 3000)
    assert max(i["chunk_index"] for i, _ in got) >= 2
    _parity.check_pipeline(emu_lib, oracle, b"")
    # the LZ4 gate wired in: a verdict per block (compthread -> lzma_compress_buf -> lz4_compresses)
    _parity.check_pipeline(emu_lib, oracle, data[:60000], ramsize=3 * 16384 // 2 + 3000, lz4_test=True, threshold=90)
    # a callback error aborts the run and comes back
    with pytest.raises(m.MrzError):
        m.rzip_pipeline(data, lambda info, payload: -6, lib=emu_lib, ramsize=3 * 16384 // 2 + 3000)


@pytest.mark.parametrize("to_stdout", [False, True])
def test_stdin_chunking(emu_lib, oracle, to_stdout):
    """SURVEY a-11: mmap_stdin / the STDIN chunk loop (src/rzip.c:700-732,915-1061, src/util.c:156-164).  ramsize is
    pinned so small that the input spans several chunks: maxram = ramsize / 3 (/ 6 towards STDOUT)."""
    ram = 6 * 8192  # chunks of 16 KiB (8 KiB towards STDOUT)
    chunk = ram // (6 if to_stdout else 3)
    text = _util.zipf_text(3 * chunk + 777, seed=21)
    for data in (b"", text[:100], text[:chunk - 1], text[:chunk], text[:chunk + 1], text[:2 * chunk], text):
        got, nch = _parity.check_stream(emu_lib, oracle, data, to_stdout, ram)
        # a length that is a multiple of the chunk size ends in one more, EMPTY chunk carrying the eof flag
        assert nch == len(data) // chunk + 1


def test_rzip_fd_on_a_pipe_and_on_a_file(emu_lib, oracle, tmp_path):
    data = _util.rep64k(5, seed=3, period=4096) + _util.zipf_text(50000, seed=5)
    _parity.check_fd(emu_lib, oracle, data, use_pipe=True, ramsize=3 * 16384, tmp_path=tmp_path)
    _parity.check_fd(emu_lib, oracle, data, use_pipe=False, ramsize=3 * 16384, tmp_path=tmp_path)


def test_corrupt_archive_with_huge_length_field(emu_lib):
    """A block length of 2^63 - 1 in an 8-byte-wide chunk must be rejected, not overflow a bounds check."""
    hdr = bytearray(b"MRZI" + bytes([0, 9]) + (100).to_bytes(8, "little") + bytes([1, 0, 0, 0, 0x77, 0]))
    chunk = bytes([8, 1]) + (4096).to_bytes(8, "little")
    head = lambda c, u, nx: bytes([3]) + c.to_bytes(8, "little") + u.to_bytes(8, "little") + nx.to_bytes(8, "little")
    big = (1 << 63) - 1
    body = head(0, 0, 50) + head(0, 0, 0) + head(big, big, 0)
    with pytest.raises(m.MrzError):
        m.runzip_buffer(bytes(hdr) + chunk + body + bytes(64), lib=emu_lib)
