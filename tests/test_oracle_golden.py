"""CPU tier: pins the oracle (oracle/liboracle.so) against the reference's golden
vectors (tests/golden/vectors.json), standard known answers and
-- where present -- oracle/_ref (the reference's own blake2b.c compiled in place)."""
import ctypes
import hashlib
import json
import os
import zlib

import pytest

from tests import _util

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

with open(os.path.join(HERE, "golden", "vectors.json")) as f:
    GOLD = json.load(f)


@pytest.fixture(scope="module")
def inputs():
    return _util.golden_inputs()


def test_hash_index_matches_reference_probe(oracle):
    H = oracle.hash_index()
    g = GOLD["hash_index"]
    assert H[0] == int(g["H0"], 16) and H[1] == int(g["H1"], 16) and H[2] == int(g["H2"], 16)
    assert H[3] == int(g["H3"], 16) and H[255] == int(g["H255"], 16)
    x = 0
    for v in H:
        x ^= v
    assert x == int(g["xor_all"], 16)


def test_hash_index_is_glibc_random():
    """src/rzip.c:672: (random() << 16) ^ random() at glibc's default seed."""
    import subprocess
    import sys
    code = ("import ctypes;c=ctypes.CDLL('libc.so.6');c.random.restype=ctypes.c_long;"
            "print([ (c.random()<<16)^c.random() for _ in range(256)])")
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout
    libc = eval(out)
    o = _util.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    assert o.hash_index() == libc


@pytest.mark.parametrize("name", list(GOLD["files"].keys()))
def test_whole_file_golden(oracle, inputs, name):
    g = GOLD["files"][name]
    mrz, stats, md5 = oracle.compress(inputs[name], level=7)
    assert len(mrz) == g["mrz_len"]
    assert hashlib.sha256(mrz).hexdigest() == g["sha256"]
    for k, v in g["stats"].items():
        assert stats[k] == v, (k, stats[k], v)
    assert md5 == hashlib.md5(inputs[name]).digest()
    if "md5" in g:
        assert md5.hex() == g["md5"]
    rc, back = oracle.decompress(mrz)
    assert rc == 0 and back == inputs[name]


def test_syn64_table_distribution(oracle, inputs):
    g = GOLD["files"]["syn64"]
    L = oracle.L
    m = ctypes.c_void_p(L.mrzo_matcher_new(7))
    s0, s1 = _util.Buf(), _util.Buf()
    data = inputs["syn64"]
    assert L.mrzo_rzip_chunk(m, data, len(data), 4, ctypes.byref(s0), ctypes.byref(s1), None) == 0
    tot, pri = ctypes.c_int64(), ctypes.c_int64()
    L.mrzo_matcher_distrib(m, ctypes.byref(tot), ctypes.byref(pri))
    assert (tot.value, pri.value) == (g["table_total"], g["table_primary"])
    assert s0.len == 23707 and s1.len == 67521  # stream sizes recorded in SURVEY 8c
    L.mrzo_buf_free(ctypes.byref(s0))
    L.mrzo_buf_free(ctypes.byref(s1))
    L.mrzo_matcher_free(m)


def test_crc32_md5_known_answers(oracle):
    for d in (b"", b"a", b"123456789", bytes(range(256)) * 37, _util.xorshift_noise(100003)):
        assert oracle.crc32(d) == zlib.crc32(d)
    assert oracle.crc32(b"123456789") == 0xCBF43926
    # MD5 via the whole-file path
    for d in (b"", b"abc", b"x" * 55, b"x" * 56, b"x" * 64, _util.xorshift_noise(1000)):
        _, _, md5 = oracle.compress(d)
        assert md5 == hashlib.md5(d).digest()


def test_blake2b_known_answers(oracle):
    # RFC 7693 appendix A
    assert oracle.blake2b(b"abc").hex() == (
        "ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
        "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    for n in (0, 1, 127, 128, 129, 255, 256, 257, 1000, 4096):
        d = _util.xorshift_noise(n, seed=n + 1)
        for outlen in (64, 32, 20):
            assert oracle.blake2b(d, outlen) == hashlib.blake2b(d, digest_size=outlen).digest()
    d = _util.xorshift_noise(1000)
    assert oracle.blake2b(d, 64, pieces=[d[:1], d[1:128], d[128:129], d[129:900], d[900:]]) == \
        hashlib.blake2b(d).digest()


def test_blake2b_against_reference_build(oracle):
    """oracle/_ref/libblake2b_ref.so = the reference's common/blake2b.c compiled in place."""
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libblake2b_ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    R = ctypes.CDLL(ref_path)

    class St(ctypes.Structure):
        _fields_ = [("h", ctypes.c_uint64 * 8), ("t", ctypes.c_uint64 * 2), ("f", ctypes.c_uint64 * 2),
                    ("buf", ctypes.c_uint8 * 128), ("buflen", ctypes.c_size_t), ("outlen", ctypes.c_size_t),
                    ("last_node", ctypes.c_uint8)]

    def ref(pieces, outlen=64):
        s = St()
        R.blake2b_init(ctypes.byref(s), ctypes.c_size_t(outlen))
        for p in pieces:
            R.blake2b_update(ctypes.byref(s), p, ctypes.c_size_t(len(p)))
        out = ctypes.create_string_buffer(64)
        R.blake2b_final(ctypes.byref(s), out, ctypes.c_size_t(outlen))
        return out.raw[:outlen]

    for n in (0, 5, 128, 129, 223, 223 * 7, 10000):
        d = _util.xorshift_noise(n, seed=3 * n + 1)
        assert oracle.blake2b(d) == ref([d])
        # rs-mrzip style: 223-byte rows (rs-mrzip/rs-mrzip.c:132-138)
        rows = [d[i:i + 223] for i in range(0, len(d), 223)]
        assert oracle.blake2b(d, pieces=rows) == ref(rows)
        assert oracle.blake2b(d, 32) == ref([d], 32)


def test_lz4_golden_sizes(oracle):
    from tests.golden import make_lz4_golden
    with open(os.path.join(HERE, "golden", "lz4_sizes.json")) as f:
        gold = json.load(f)["sizes"]
    for name, data in make_lz4_golden.cases():
        assert oracle.lz4_size(data) == gold[name], name


def test_lz4_against_system_liblz4(oracle):
    try:
        z = ctypes.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("liblz4.so.1 not installed")
    import random
    random.seed(11)
    for n in list(range(0, 40)) + [64, 255, 1000, 4096, 65535, 65546, 65547, 65548, 200000]:
        for kind in range(4):
            if kind == 0:
                d = _util.xorshift_noise(n, seed=n + 7)
            elif kind == 1:
                d = _util.zipf_text(n, seed=n + 1) if n else b""
            elif kind == 2:
                d = bytes(n)
            else:
                d = (_util.xorshift_noise(97, seed=n) * (n // 97 + 1))[:n]
            for cap in (n + 1, n + n // 255 + 16, max(1, n // 2)):
                dst = ctypes.create_string_buffer(max(1, cap))
                r = z.LZ4_compress_default(d, dst, n, cap)
                mine, blob = oracle.lz4_compress(d, cap)
                assert mine == r, (n, kind, cap)
                assert blob == dst.raw[:r]


def test_lz4_gate_verdicts(oracle):
    assert oracle.lz4_compresses(_util.xorshift_noise(300000), 100) == 0
    t = _util.zipf_text(300000, seed=5)
    want = int(100 * oracle.lz4_size(t) / len(t))
    assert oracle.lz4_compresses(t, 100) == want
    # threshold below the achievable ratio => "does not compress enough"
    assert oracle.lz4_compresses(t, want - 5) == 0


def test_plan_and_chunking(oracle):
    mc, bs = oracle.plan(1000)
    assert mc == (60 << 30) // 3 * 2 and bs == 10 << 20
    mc, bs = oracle.plan(0)
    assert bs == 4096
    mc, bs = oracle.plan(300 << 20, window=1)
    assert mc == 100 << 20 and bs == 300 << 20
    mc, bs = oracle.plan(123456789, unlimited=True)
    assert mc == 123456789


def test_multi_chunk_roundtrip_and_victim_round(oracle):
    """src/rzip.c:259: victim_round survives chunk boundaries."""
    base = _util.rep64k(40, seed=21)
    data = base * 1
    o = oracle
    # force 2 chunks by a tiny ramsize (max_chunk = ramsize/3*2, page-rounded)
    mrz, st, _ = o.compress(data, ramsize=3 * (len(data) // 2 + 4096) // 2)
    rc, back = o.decompress(mrz)
    assert rc == 0 and back == data
    a = o.rzip_chunk(data[:len(data) // 2], victim_round=0)
    b0 = o.rzip_chunk(data[len(data) // 2:], victim_round=0)
    b1 = o.rzip_chunk(data[len(data) // 2:], victim_round=a["victim_round"])
    assert a["victim_round"] != 0 or b0 == b1


def test_rs_encoder_against_reference_build(oracle):
    """oracle/_ref/librs_ref.so = the reference's rs-mrzip/reed-solomon.c compiled in place:
    rse32 (parity, :115-141) and scatter (interleave, :311-321)."""
    ref_path = os.path.join(ROOT, "oracle", "_ref", "librs_ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    R = ctypes.CDLL(ref_path)
    for trial in range(64):
        row = bytes(223) if trial == 0 else _util.xorshift_noise(223, seed=trial)
        buf = (ctypes.c_uint8 * 255)(*row, *([0] * 32))
        R.rse32(buf, ctypes.byref(buf, 223))
        assert bytes(buf)[:223] == row and bytes(buf)[223:] == oracle.rs_parity(row)
    # one whole burst through the reference's rse32 + scatter == the oracle's encoder body
    rows, k, n = 8176, 223, 255
    data = _util.xorshift_noise(rows * k - 1000, seed=9)
    ec = (ctypes.c_uint8 * (rows * n))()
    padded = data + bytes(1000)
    for r in range(rows):
        ctypes.memmove(ctypes.byref(ec, r * n), padded[r * k:(r + 1) * k], k)
        R.rse32(ctypes.byref(ec, r * n), ctypes.byref(ec, r * n + k))
    tr = (ctypes.c_uint8 * (rows * n))()
    R.scatter(ec, tr, rows, n)
    enc = oracle.rs_encode(data)
    assert len(enc) == rows * n + 68
    assert enc[: rows * n] == bytes(tr)
    assert enc[rows * n: rows * n + 64] == hashlib.blake2b(padded).digest()
    ki, kj = divmod(len(data), k)
    assert enc[-4:] == bytes([ki & 255, ki >> 8, kj & 255, kj >> 8])


def test_rs_burst_count_follows_feof(oracle):
    """rs-mrzip.c:125: feof() only turns true after a short read, so an input that is an exact
    multiple of a burst produces one more all-padding burst."""
    burst = 223 * 8176
    assert len(oracle.rs_encode(b"")) == 255 * 8176 + 68
    assert oracle.L.mrzo_rs_encoded_size(burst - 1) == 255 * 8176 + 68
    assert oracle.L.mrzo_rs_encoded_size(burst) == 2 * 255 * 8176 + 68
