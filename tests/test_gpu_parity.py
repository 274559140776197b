"""GPU tier (-m gpu): libmrzgpu.so on a real MI355X through the C ABI, bit-exact
against the oracle on seeded inputs, against the reference's golden vectors, and
-- at sizes the oracle cannot reach quickly -- through size-independent
properties (decode round trip, CRC/MD5 equality, stream accounting)."""
import hashlib
import json
import os
import zlib

import pytest

import modern_rzip_amd as m
from modern_rzip_amd import workloads as w
from tests import _parity, _util

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
with open(os.path.join(HERE, "golden", "vectors.json")) as f:
    GOLD = json.load(f)


@pytest.fixture(scope="module")
def inputs():
    return _util.golden_inputs()


def test_native_library_is_loaded(gpu_lib):
    import torch
    assert torch.cuda.is_available()
    assert gpu_lib.mrz_abi_version() == 4
    assert os.path.basename(m.lib_path()) == "libmrzgpu.so"
    with open("/proc/self/maps") as f:
        assert "libmrzgpu.so" in f.read()


@pytest.mark.parametrize("name", list(GOLD["files"].keys()))
def test_reference_golden_vectors(gpu_lib, oracle, inputs, name):
    """Whole `mrzip -n -L7` output == the reference's."""
    g = GOLD["files"][name]
    got, st, md5 = m.rzip_buffer(inputs[name], level=7, lib=gpu_lib)
    assert len(got) == g["mrz_len"]
    assert hashlib.sha256(got).hexdigest() == g["sha256"]
    for k, v in g["stats"].items():
        assert getattr(st, k) == v, k
    rc, back = oracle.decompress(got)
    assert rc == 0 and back == inputs[name]


@pytest.mark.parametrize("n", [0, 1, 30, 31, 32, 33, 63, 64, 65, 4095, 4096, 4097, 65535, 65536, 65537, 100003])
def test_ragged_sizes(gpu_lib, oracle, n):
    _parity.check_chunk(gpu_lib, oracle, (_util.zipf_text(max(n, 1), seed=n + 1) * 2)[:n])


@pytest.mark.parametrize("level", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_levels_on_periodic_text(gpu_lib, oracle, level):
    _parity.check_chunk(gpu_lib, oracle, _util.rep64k(40, seed=5), level=level)


def test_syn64_chunk_state(gpu_lib, oracle, inputs):
    want = _parity.check_chunk(gpu_lib, oracle, inputs["syn64"], table=True)
    assert len(want["s0"]) == 23707 and len(want["s1"]) == 67521


@pytest.mark.parametrize("vr", [0, 1, 7, 15])
def test_victim_round_in_out(gpu_lib, oracle, vr):
    _parity.check_chunk(gpu_lib, oracle, _util.rep64k(48, seed=9, period=4096), victim_round=vr)


def test_text_with_culling(gpu_lib, oracle):
    # 24 MB of Zipf text: the table passes its 2/3 limit, several cull sweeps complete
    want = _parity.check_chunk(gpu_lib, oracle, _util.zipf_text(12 << 20, seed=7), table=True)
    assert want["min_mask"] > 1


def test_noise_with_culling(gpu_lib, oracle):
    want = _parity.check_chunk(gpu_lib, oracle, _util.xorshift_noise(8 << 20, seed=3))
    # the table reached its 2/3 limit (src/rzip.c:529,583): inserts beyond it cull entries
    assert want["stats"]["inserts"] > want["hash_count"] > 2_700_000 and want["stats"]["matches"] == 0


def test_tar_like_mix(gpu_lib, oracle):
    _parity.check_chunk(gpu_lib, oracle, _util.tar_like(8 << 20, seed=5))


@pytest.mark.parametrize("wgs", ["1", "2", "4"])
def test_sequencer_workgroup_counts(gpu_lib, oracle, wgs, monkeypatch):
    """The wide engine with 1, 2 and 4 sequencer workgroups taking turns (3 is the default every other test runs with):
    same streams, counters and table as the oracle, on text, noise (culling, mask promotion) and the tar mix."""
    monkeypatch.setenv("MRZ_SEQ_WGS", wgs)
    _parity.check_chunk(gpu_lib, oracle, _util.zipf_text(6 << 20, seed=11), table=True)
    _parity.check_chunk(gpu_lib, oracle, _util.xorshift_noise(24 << 20, seed=12), table=True)
    _parity.check_chunk(gpu_lib, oracle, _util.tar_like(12 << 20, seed=13), level=4)


def test_crowded_chains_run_the_entry_pool_dry(gpu_lib, oracle):
    """A 997-byte period at level 7: every look-up finds max_chain_len tag-equal entries, so a 512-lane batch wants
    7-8 k pool entries where the LDS pool holds 4 k; which lanes get chunks depends on the order of the atomics.  A lane
    that got none must not store through a chunk id of an earlier batch (it once did: a fuzz find, hits counted as
    misses in some runs only).  Several runs, several workgroup counts: every one must equal the oracle."""
    data = _util.rep64k(4750705 // 997, seed=628395130, period=997)
    import os
    for wgs in ("1", "2", "3"):
        os.environ["MRZ_SEQ_WGS"] = wgs
        try:
            for _ in range(3):
                _parity.check_chunk(gpu_lib, oracle, data, level=7, victim_round=2)
        finally:
            del os.environ["MRZ_SEQ_WGS"]


def test_stride_repeats_unlimited_window(gpu_lib, oracle):
    """BASELINE configs[3] shape, scaled: noise segments with planted repeats 1, 3 and 7 segments back, one chunk
    (-U).  Matches are a quarter segment long and reach back up to 7 segments."""
    data = w.stride_stream(48, 512 << 10)
    want = _parity.check_chunk(gpu_lib, oracle, data)
    assert want["stats"]["match_bytes"] >= 10 * (128 << 10)
    got = _parity.check_file(gpu_lib, oracle, data[: 12 << 20], unlimited=True)
    assert len(got) < 12 << 20


@pytest.mark.parametrize("farm_wgs", ["0", "40"])
def test_small_or_no_compare_farm(farm_wgs):
    """The helper-workgroup count is a launch parameter: without helpers (local striped compares only) and with
    a two-row farm the streams must be the same bits.  Own process: the count is read once per process."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import modern_rzip_amd as m\n"
        "from tests import _parity, _util\n"
        "import os\n"
        "o = _util.Oracle(os.path.join(%r, 'oracle', 'liboracle.so'))\n"
        "_parity.check_chunk(m.load_library(), o, _util.rep64k(160, seed=21))\n"
        "print('ok')\n" % (ROOT, ROOT))
    env = dict(os.environ, MRZ_FARM_WGS=farm_wgs)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("wgs", ["2", "3"])
def test_no_farm_mailbox_with_several_sequencer_workgroups(gpu_lib, oracle, wgs, monkeypatch):
    """MRZ_NO_HELPER_WGS=1: no mailbox is allocated, yet the grid still has blocks between the sequencer workgroups
    (0, 8, 16): they must leave at once instead of taking a farm ticket from a null mailbox."""
    monkeypatch.setenv("MRZ_NO_HELPER_WGS", "1")
    monkeypatch.setenv("MRZ_SEQ_WGS", wgs)
    _parity.check_chunk(gpu_lib, oracle, _util.rep64k(160, seed=21))
    _parity.check_chunk(gpu_lib, oracle, _util.zipf_text(3 << 20, seed=5))


def test_segment_boundaries(gpu_lib, oracle):
    # > 16 Mi positions: the chunk spans two tag-scan segments, with matches crossing the seam
    blk = _util.zipf_text(9 << 20, seed=31)
    _parity.check_chunk(gpu_lib, oracle, blk + blk[: 8 << 20] + _util.xorshift_noise(1 << 20, seed=2))


def test_long_match_pieces_and_backward(gpu_lib, oracle):
    blk = _util.xorshift_noise(300000, seed=12)
    _parity.check_chunk(gpu_lib, oracle, blk + blk + b"xyz" + blk[5:] + blk)


def test_farm_multi_round_matches(gpu_lib, oracle):
    """Matches far longer than one farm round covers (224 KiB per entry and round): several tag-equal entries that
    are all ~700 KB long (rounds continue until the last entry stops), a lone long entry (local round, then the
    farm), and a match that runs into the end of the chunk."""
    blk = _util.xorshift_noise(700000, seed=31)
    tail = _util.xorshift_noise(900, seed=32)
    data = blk + b"#" + blk + b"##" + blk + b"###" + blk[:650000] + tail + blk[1000:]
    want = _parity.check_chunk(gpu_lib, oracle, data)
    assert want["stats"]["match_bytes"] > 3 * 600000


def test_exact_repeat_is_one_giant_match(gpu_lib, oracle):
    """A block followed by exact copies of itself (what the S2 stream does beyond 4 GiB, what duplicate tar
    members do): the copies are ONE match of ~88 MiB -- bulk farm rounds (every helper, 64-128 KiB each), host
    skipping of the segments the match covers, ~1400 0xFFFF pieces written by a whole workgroup."""
    blk = _util.rep64k(128, seed=41)  # 8 MiB
    data = blk * 12
    want = _parity.check_chunk(gpu_lib, oracle, data)
    assert want["stats"]["matches"] > 1300 and want["stats"]["match_bytes"] > 11 * len(blk) - 100000
    with m.RzipContext(lib=gpu_lib, max_chunk=len(data)) as ctx:
        ctx.set_profiling(True)
        ctx.rzip_chunk(data, fetch=False)
        assert ctx.timings().n_segments < 6  # 96 MiB = 6 segments of 16 Mi positions; most are never launched


def test_backend_handoff_pipeline(gpu_lib, oracle):
    """mrz_rzip_pipeline: blocks of stream_bufsize bytes in the reference's flush order, several per stream
    (24 MiB of literals against a 10 MiB buffer) and several chunks, GPU work overlapping the consumer."""
    _parity.check_pipeline(gpu_lib, oracle, _util.xorshift_noise(24 << 20, seed=17) + _util.rep64k(40, seed=3))
    # the LZ4 gate wired in (default mode: compthread -> lzma_compress_buf -> lz4_compresses): a verdict per block
    _parity.check_pipeline(gpu_lib, oracle, _util.tar_like(40 << 20, seed=9), ramsize=60 << 20, lz4_test=True)


def test_pipeline_hands_blocks_over_before_the_chunk_is_done(gpu_lib, oracle):
    """Within-chunk hand-off (write_sbstream -> flush_buffer during hash_search, src/rzip.c:197-211): on a ONE-chunk
    input of several segments the first block reaches the consumer while later segments have not been sequenced."""
    # every 1 MiB of noise is followed by a copy of itself: half of the input is literal bytes (stream 1 fills a
    # 10 MiB block every 20 MiB of input), and a match is emitted every 2 MiB, so bytes become final as the segments
    # go by (hash_search only writes literals when a match is emitted, src/rzip.c:593: match-free input has nothing to
    # hand over before its end, in the reference as here).  6 segments of 16 Mi positions.
    noise = _util.xorshift_noise(48 << 20, seed=23)
    data = b"".join(noise[a:a + (1 << 20)] * 2 for a in range(0, len(noise), 1 << 20))
    got = _parity.check_pipeline(gpu_lib, oracle, data, ramsize=30 << 20, unlimited=True)
    assert {i["chunk_index"] for i, _ in got} == {0}
    first = got[0][0]
    assert first["input_final"] < len(data) // 2, first  # cut long before the chunk's last segment was launched
    assert got[-1][0]["input_final"] == len(data)
    got = _parity.check_pipeline(gpu_lib, oracle, _util.rep64k(96, seed=13), ramsize=3 << 20)
    assert max(i["chunk_index"] for i, _ in got) >= 2


def test_concurrent_contexts(gpu_lib, oracle):
    """Independent streams on one GPU at once: one ctx and host thread each, the helper workgroups split between
    them (mrz_set_farm_helpers).  Every stream must come out exactly as it does alone."""
    import threading
    datas = [_util.rep64k(48, seed=61), _util.zipf_text(3 << 20, seed=62), _util.rep64k(20, seed=63, period=30000) * 3,
             _util.xorshift_noise(2 << 20, seed=64)]
    wants = [oracle.rzip_chunk(d) for d in datas]
    ctxs = [m.RzipContext(lib=gpu_lib, max_chunk=len(d)) for d in datas]
    got = [None] * len(datas)

    def work(i):
        ctxs[i].set_farm_helpers(224 // len(datas))
        res, s0, s1 = ctxs[i].rzip_chunk(datas[i])
        got[i] = (s0, s1, res.crc32, res.stats.as_dict())

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(datas))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in ctxs:
        c.close()
    for g, wnt in zip(got, wants):
        assert g == (wnt["s0"], wnt["s1"], wnt["crc"], wnt["stats"])


def test_multi_chunk_file(gpu_lib, oracle):
    data = _util.rep64k(96, seed=13)  # 6 MiB, chunks of 2 MiB+
    _parity.check_file(gpu_lib, oracle, data, ramsize=3 * (2 << 20) // 2 + 5000)
    _parity.check_file(gpu_lib, oracle, _util.tar_like(3 << 20, seed=8), level=9)


def test_device_resident_input(gpu_lib, oracle):
    import torch
    data = _util.rep64k(64, seed=17)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    want = oracle.rzip_chunk(data)
    with m.RzipContext(lib=gpu_lib, max_chunk=len(data)) as ctx:
        res, s0, s1 = ctx.rzip_chunk(t)
        assert (s0, s1, res.crc32) == (want["s0"], want["s1"], want["crc"])
        # second chunk on the same ctx: table is reset, victim_round is chained
        want2 = oracle.rzip_chunk(data, victim_round=want["victim_round"])
        res2, s0b, s1b = ctx.rzip_chunk(t)
        assert (s0b, s1b) == (want2["s0"], want2["s1"])
        assert ctx.victim_round == want2["victim_round"]


def test_crc32_kernel(gpu_lib):
    with m.RzipContext(lib=gpu_lib) as ctx:
        for n in (0, 1, 15, 17, 65535, 65536, 65537, 10 * 65536 + 77, (64 << 20) + 13):
            d = _util.xorshift_noise(n, seed=n % 1000 + 5)
            assert ctx.crc32(d) == zlib.crc32(d), n


def test_large_property_roundtrip(gpu_lib, oracle):
    """1 GiB S2-style stream (too long for the oracle's matcher in a test): the
    archive must decode back to the input with matching CRC/MD5, and the stream
    accounting must add up."""
    import numpy as np
    nper = 16384
    data = _util.rep64k(nper, seed=1234)
    with m.RzipContext(lib=gpu_lib, max_chunk=len(data)) as ctx:
        res, s0, s1 = ctx.rzip_chunk(data)
    assert res.crc32 == zlib.crc32(data)
    st = res.stats
    assert st.literal_bytes + st.match_bytes == len(data)
    assert st.literal_bytes == len(s1)
    md5 = hashlib.md5(data).digest()
    mrz = oracle.frame(len(data), [(len(data), s0, s1)], md5)
    rc, back = oracle.decompress(mrz)
    assert rc == 0
    assert hashlib.sha256(back).digest() == hashlib.sha256(data).digest()


def test_runzip_device_roundtrip_1gib(gpu_lib):
    """Encode -> decode entirely in HBM: 1 GiB S2 stream, streams and output never leave the device."""
    import torch
    t = w.rep64k_device(16384, "cuda")
    n = t.numel()
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    with m.RzipContext(lib=gpu_lib, max_chunk=n) as ctx:
        res, _, _ = ctx.rzip_chunk(t, fetch=False)
        _, got, cc, cs = ctx.runzip_chunk((res.d_s0, res.s0_len), (res.d_s1, res.s1_len), m.chunk_bytes(n, lib=gpu_lib),
                                          n, out=out)
    assert got == n and cc == cs == res.crc32
    assert torch.equal(out, t)


def test_bench_config_roundtrip_10gib(gpu_lib):
    """BASELINE configs[1] at full size (10 GiB, chunk_bytes 5): encode and decode in HBM, outputs must agree
    with the input, the CRCs with each other, and the stream accounting must add up."""
    import torch
    t = w.rep64k_device(163840, "cuda")
    n = t.numel()
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    with m.RzipContext(lib=gpu_lib, max_chunk=n) as ctx:
        res, _, _ = ctx.rzip_chunk(t, fetch=False)
        assert m.chunk_bytes(n, lib=gpu_lib) == 5
        _, got, cc, cs = ctx.runzip_chunk((res.d_s0, res.s0_len), (res.d_s1, res.s1_len), 5, n, out=out)
        assert ctx.crc32(t) == res.crc32
    assert got == n and cc == cs == res.crc32
    assert res.stats.literal_bytes + res.stats.match_bytes == n and res.stats.literal_bytes == res.s1_len
    assert torch.equal(out, t)


@pytest.mark.parametrize("kind", ["text", "noise", "tar"])
def test_runzip_roundtrip_shapes(gpu_lib, kind):
    """GPU encoder -> GPU decoder on non-periodic shapes (many short records / no matches / mixed)."""
    data = {"text": lambda: w.zipf_text(24 << 20, seed=3), "noise": lambda: w.noise(8 << 20, seed=4),
            "tar": lambda: w.tar_like(32 << 20, seed=6)}[kind]()
    with m.RzipContext(lib=gpu_lib, max_chunk=len(data)) as ctx:
        res, s0, s1 = ctx.rzip_chunk(data)
        back, got, cc, cs = ctx.runzip_chunk(s0, s1, m.chunk_bytes(len(data), lib=gpu_lib), len(data))
    assert got == len(data) and cc == cs == zlib.crc32(data)
    assert back == data


def test_runzip_golden_archives(gpu_lib, oracle, inputs):
    """`mrzip -d` of the reference-identical golden archives through the GPU decoder."""
    for name, data in inputs.items():
        mrz, _, _ = oracle.compress(data)
        assert hashlib.sha256(mrz).hexdigest() == GOLD["files"][name]["sha256"]
        assert m.runzip_buffer(mrz, lib=gpu_lib) == data
        if len(mrz) > 40:  # a flipped payload byte must be caught (MD5 / CRC / record validation)
            bad = bytearray(mrz)
            bad[len(bad) // 2] ^= 0x40
            with pytest.raises(m.MrzError):
                m.runzip_buffer(bytes(bad), lib=gpu_lib)


def test_blake2b_kernels(gpu_lib, oracle):
    import torch
    with m.RzipContext(lib=gpu_lib) as ctx:
        msgs = [_util.xorshift_noise(n, seed=n + 1) for n in
                (0, 1, 127, 128, 129, 256, 257, 1000, 65536, 1 << 20)] + [b"abc"] * 70
        for outlen in (64, 32):
            got = ctx.blake2b_batch(msgs, outlen)
            assert got == [hashlib.blake2b(x, digest_size=outlen).digest() for x in msgs]
        d = _util.xorshift_noise(223 * 999 + 40, seed=2)
        rows = [d[i:i + 223 * 100] for i in range(0, len(d), 223 * 100)]
        assert ctx.blake2b(d, pieces=rows) == hashlib.blake2b(d).digest() == oracle.blake2b(d)
        t = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        assert ctx.blake2b(t) == hashlib.blake2b(d).digest()


def test_lz4_sizes_and_gate(gpu_lib, oracle):
    with open(os.path.join(HERE, "golden", "lz4_sizes.json")) as f:
        gold = json.load(f)["sizes"]
    from tests.golden import make_lz4_golden
    cases = list(make_lz4_golden.cases())
    with m.RzipContext(lib=gpu_lib) as ctx:
        got = ctx.lz4_sizes([d for _, d in cases])
        for (name, d), g in zip(cases, got):
            assert g == gold[name], name
        edge = [bytes(n) for n in (1, 12, 13, 14, 64)] + [_util.zipf_text(n, seed=n) for n in (13, 20, 100, 3000)]
        assert ctx.lz4_sizes(edge) == [oracle.lz4_size(d) for d in edge]
        blocks = [_util.zipf_text(3 << 20, seed=4), _util.xorshift_noise(3 << 20, seed=4), bytes(500000),
                  _util.xorshift_noise(11 << 20, seed=5) + _util.zipf_text(2 << 20, seed=6),
                  _util.tar_like(4 << 20, seed=3)]
        for thr in (100, 60):
            assert ctx.lz4_compresses(blocks, thr) == [oracle.lz4_compresses(b, thr) for b in blocks]


def test_gate_on_gpu_preprocessed_streams(gpu_lib, oracle):
    """cfg3 shape: GPU rzip -> LZ4 gate on both streams of the chunk."""
    data = _util.tar_like(8 << 20, seed=21)
    with m.RzipContext(lib=gpu_lib, max_chunk=len(data)) as ctx:
        res, s0, s1 = ctx.rzip_chunk(data)
        got = ctx.lz4_compresses([s0, s1], 100)
    assert got == [oracle.lz4_compresses(s0, 100), oracle.lz4_compresses(s1, 100)]


def test_rs_encoder(gpu_lib, oracle):
    """rs-mrzip encode: parity + interleave on the GPU, byte-identical to the oracle
    (itself pinned to the reference's reed-solomon.c)."""
    import torch
    burst = 223 * 8176
    with m.RzipContext(lib=gpu_lib) as ctx:
        for n in (0, 1, 223, burst - 1, burst, burst + 1, 3 * burst + 12345):
            d = _util.xorshift_noise(n, seed=n % 977 + 3)
            assert hashlib.sha256(ctx.rs_encode(d)).digest() == hashlib.sha256(oracle.rs_encode(d)).digest(), n
        d = _util.zipf_text(2 * burst + 77, seed=5)
        t = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        assert ctx.rs_encode(t) == oracle.rs_encode(d)


# ---- round 2: full-size bit-exactness, both sequencer engines, the STDIN form ---------------------------------

def _sha(b):
    return hashlib.sha256(b).hexdigest()


def _exact_vs_oracle(gpu_lib, oracle, data, victim_round=0):
    """streams (sha256), CRC, the seven counters, victim_round, final mask and hash_count -- all equal."""
    want = oracle.rzip_chunk(data, level=7, victim_round=victim_round)
    with m.RzipContext(lib=gpu_lib, max_chunk=len(data)) as ctx:
        ctx.victim_round = victim_round
        res, s0, s1 = ctx.rzip_chunk(data)
        assert (_sha(s0), _sha(s1)) == (_sha(want["s0"]), _sha(want["s1"]))
        assert res.crc32 == want["crc"] and res.stats.as_dict() == want["stats"]
        assert ctx.victim_round == want["victim_round"]
        assert res.min_mask == want["min_mask"] and res.hash_count == want["hash_count"]
        return ctx.timings()


def test_bench_config_10gib_bit_exact_vs_oracle(gpu_lib, oracle):
    """BASELINE configs[1] at FULL size against the oracle (which takes ~5 s for it): both streams, the counters
    and the chained victim_round, not only a round trip."""
    data = w.rep64k_device(163840, "cuda").cpu().numpy().tobytes()
    t = _exact_vs_oracle(gpu_lib, oracle, data)
    assert t.n_narrow > 0  # the regime hint sent the stream to the narrow engine


def test_text_100m_bit_exact_vs_oracle(gpu_lib, oracle):
    """S1 text-100M (the shape of BASELINE configs[0]) against the oracle."""
    _exact_vs_oracle(gpu_lib, oracle, w.zipf_text(100_000_000))


def test_tar_like_1gib_bit_exact_vs_oracle(gpu_lib, oracle):
    """1 GiB of the S3 tar mix (text and noise members, exact duplicates, repeats at distances of hundreds of MiB)
    against the oracle."""
    _exact_vs_oracle(gpu_lib, oracle, w.tar_like_fast(1 << 30, seed=11))


@pytest.mark.parametrize("engine", ["wide", "narrow", "deep"])
def test_both_engines_pinned(gpu_lib, oracle, engine, monkeypatch):
    """Every shape through each of the three sequencer kernels alone (MRZ_SEQ_ENGINE pins the per-segment choice)."""
    monkeypatch.setenv("MRZ_SEQ_ENGINE", engine)
    _parity.check_chunk(gpu_lib, oracle, _util.zipf_text(12 << 20, seed=3), table=True)
    _parity.check_chunk(gpu_lib, oracle, _util.xorshift_noise(24 << 20, seed=5), table=True)
    _parity.check_chunk(gpu_lib, oracle, _util.tar_like(8 << 20, seed=7), table=True, victim_round=9)
    _parity.check_chunk(gpu_lib, oracle, _util.rep64k(512, seed=1234), table=True)
    _parity.check_chunk(gpu_lib, oracle, _util.zipf_text(3 << 20, seed=8), level=1, table=True)
    _parity.check_chunk(gpu_lib, oracle, _util.zipf_text(3 << 20, seed=8), level=9, table=True)


@pytest.mark.parametrize("scanners", ["0", "3"])
def test_deep_engine_scan_helper_counts(gpu_lib, oracle, scanners, monkeypatch):
    """The deep engine without scan helpers and with three (default: up to 15): the lanes of a batch are dealt to however
    many helpers have checked in; the result does not depend on it.  512 MiB of noise reaches a 7-bit mask (wide engine
    first, which ends its launch where the mask reaches six bits); the tar mix adds real matches inside deep segments."""
    monkeypatch.setenv("MRZ_DEEP_SCANNERS", scanners)
    monkeypatch.setenv("MRZ_DEEP_MIN_BITS", "4")
    data = w.noise(192 << 20, seed=21) + w.tar_like_fast(64 << 20, seed=22, pool_bytes=8 << 20)
    want = oracle.rzip_chunk(data)
    with m.RzipContext(lib=gpu_lib, max_chunk=len(data)) as ctx:
        res, s0, s1 = ctx.rzip_chunk(data)
        assert (s0, s1) == (want["s0"], want["s1"]) and res.stats.as_dict() == want["stats"] and res.crc32 == want["crc"]
        assert res.min_mask == want["min_mask"] and res.hash_count == want["hash_count"]
        t = ctx.timings()
        assert 1 <= t.n_deep < t.n_segments


def test_engines_alternate_within_one_chunk(gpu_lib, oracle):
    """Text, then a long stretch of one match after another, then noise: the per-segment choice changes engine in the
    middle of the chunk, the two kernels continue each other through the matcher state."""
    data = _util.zipf_text(40 << 20, seed=2) + _util.rep64k(3072, seed=1234) + _util.xorshift_noise(40 << 20, seed=4)
    t = _exact_vs_oracle(gpu_lib, oracle, data)
    assert 0 < t.n_narrow < t.n_segments


@pytest.mark.parametrize("to_stdout", [False, True])
def test_stdin_chunking(gpu_lib, oracle, to_stdout):
    """SURVEY a-11 / BASELINE configs[4] (scaled): the STDIN form of the chunk loop, several chunks incl. the empty
    eof chunk of an input whose length is a multiple of the chunk size."""
    ram = 6 << 20
    chunk = ram // (6 if to_stdout else 3)
    text = _util.tar_like(3 * chunk + 12345, seed=21)
    for data in (b"", text[:chunk - 1], text[:chunk], text[:2 * chunk], text):
        got, nch = _parity.check_stream(gpu_lib, oracle, data, to_stdout, ram)
        assert nch == len(data) // chunk + 1


def test_rzip_fd_on_a_pipe_and_on_a_file(gpu_lib, oracle, tmp_path):
    data = _util.rep64k(40, seed=3) + _util.zipf_text(3 << 20, seed=5)
    _parity.check_fd(gpu_lib, oracle, data, use_pipe=True, ramsize=3 << 20, tmp_path=tmp_path)
    _parity.check_fd(gpu_lib, oracle, data, use_pipe=False, ramsize=3 << 20, tmp_path=tmp_path)


def test_window_sharded_over_two_processes(gpu_lib):
    """BASELINE configs[3] in its defining form, scaled to one box: ONE 256 MiB window (noise segments with planted
    repeats at 1-, 3- and 7-segment strides) whose halves live in shareable allocations of two fresh processes (both on
    GPU 0, gloo transport); both map both halves into one address range (mrz_window_map_create), each scans the
    stretches of its half (mrz_window_scan from the mapping, halo included) and ships compacted candidates to rank 0,
    whose matcher, compare farm, CRC and literal gather read rank 1's bytes through the mapping.  Rank 0 compares both
    streams, the counters, the CRC and victim_round with the oracle."""
    from tests.test_distributed import run_window_workers
    out = run_window_workers("gpu", 16, 16 << 20, timeout=900)
    assert "window ok 268435456" in out


def test_eight_matchers_on_eight_xcds(gpu_lib):
    """Eight independent streams at once, one ctx each on its own XCD (mrz_set_xcd; own process: the HIP runtime has to
    open enough hardware queues): every stream's result equals its solo run and the oracle."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_multictx.py"), "tar", "24", "8", "--check"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert last == {"oracle_equal": True, "all_identical": True}, r.stdout[-2000:]
    # ... and the chunks of ONE file through eight ctxs (victim_round chained speculatively): the oracle's archive
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_multictx.py"), "chunks", "16", "8", "8"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert last["archive_equals_oracle"] and last["reruns"] == 0, r.stdout[-2000:]


def test_c_caller_program(gpu_lib, tmp_path):
    """tests/c/capi_test.c: a plain C99 program (gcc, no ctypes) linked against libmrzgpu.so drives the chunk call and
    mrz_rzip_fd (file and pipe) and compares every byte with the oracle itself."""
    import subprocess
    exe = str(tmp_path / "capi_test")
    subprocess.run(["gcc", "-std=c99", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
                    os.path.join(ROOT, "tests", "c", "capi_test.c"), "-o", exe,
                    "-L" + os.path.join(ROOT, "modern-rzip_amd"), "-lmrzgpu", "-L" + os.path.join(ROOT, "oracle"), "-loracle"],
                   check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([os.path.join(ROOT, "modern-rzip_amd"), os.path.join(ROOT, "oracle"),
                                                            os.environ.get("LD_LIBRARY_PATH", "")]))
    out = subprocess.run([exe], check=True, env=env, stdout=subprocess.PIPE, text=True).stdout
    assert "capi_test ok" in out


def test_coresident_blake2b_and_rs_beside_rzip(gpu_lib):
    """BASELINE configs[4] (scaled): the BLAKE2b kernels and the rs-mrzip encoder run from other host threads, on
    their own ctxs / streams, WHILE a chunk is being sequenced -- every output identical to its solo run."""
    sys_path_tools = os.path.join(ROOT, "tools")
    import importlib.util
    spec = importlib.util.spec_from_file_location("probe_coresident", os.path.join(sys_path_tools, "probe_coresident.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    line = mod.main(rz_mib=128, b2_msgs=512, b2_kib=256, rs_mib=64)
    assert line["outputs_identical"]
    assert line["together_wall_s"] < line["sum_of_solo_s"] * 1.05  # they overlap instead of queueing behind each other


def test_rs_decoder_against_the_reference(gpu_lib):
    """mrz_rs_decode vs the reference's own rsd32 / gather (oracle/_ref/librs_ref.so) on damaged encodings."""
    data = _util.xorshift_noise(2 * 1823248 + 777, seed=15)  # three bursts
    with m.RzipContext(lib=gpu_lib) as ctx:
        got, rep = _parity.check_rs_decode(ctx, data, [])
        assert got == data and rep == dict(corrected=0, uncorrectable=0, checksum_ok=True, truncated=False)
        dmg = [(5000, 16 * 8176, 0xa5), (2084880 + 999, 3 * 8176 + 100, 0x3c)]  # 16 errors per codeword: the limit
        got, rep = _parity.check_rs_decode(ctx, data, dmg)
        assert got == data and rep["checksum_ok"] and rep["corrected"] == 16 * 8176 + 3 * 8176 + 100
        dmg.append((2 * 2084880, 17 * 8176, 0x77))  # 17 per codeword in the last burst: beyond repair
        got, rep = _parity.check_rs_decode(ctx, data, dmg)
        assert rep["uncorrectable"] > 0 and not rep["checksum_ok"]
