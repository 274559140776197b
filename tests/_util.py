"""Test helpers: ctypes view of the oracle, seeded workload generators."""
import ctypes
import random

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Buf(ctypes.Structure):
    _fields_ = [("p", ctypes.POINTER(ctypes.c_uint8)), ("len", ctypes.c_int64), ("cap", ctypes.c_int64)]


class OStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in
                ("inserts", "literals", "literal_bytes", "matches", "match_bytes", "tag_hits", "tag_misses")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Params(ctypes.Structure):
    _fields_ = [("level", ctypes.c_int), ("window", ctypes.c_int64), ("unlimited", ctypes.c_int),
                ("ramsize", ctypes.c_int64), ("page_size", ctypes.c_int64)]


class ChunkStreams(ctypes.Structure):
    _fields_ = [("chunk_size", ctypes.c_int64), ("s0", ctypes.c_char_p), ("s0_len", ctypes.c_int64),
                ("s1", ctypes.c_char_p), ("s1_len", ctypes.c_int64)]


class Oracle:
    def __init__(self, path):
        L = self.L = ctypes.CDLL(path)
        L.mrzo_matcher_new.restype = ctypes.c_void_p
        L.mrzo_matcher_new.argtypes = [ctypes.c_int]
        L.mrzo_matcher_free.argtypes = [ctypes.c_void_p]
        L.mrzo_matcher_stats.restype = ctypes.POINTER(OStats)
        L.mrzo_matcher_stats.argtypes = [ctypes.c_void_p]
        for f in ("mrzo_matcher_get_victim_round", "mrzo_matcher_min_mask", "mrzo_matcher_hash_count"):
            getattr(L, f).restype = ctypes.c_int64
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.mrzo_matcher_set_victim_round.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.mrzo_matcher_table.restype = ctypes.c_void_p
        L.mrzo_matcher_table.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        L.mrzo_rzip_chunk.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int,
                                      ctypes.POINTER(Buf), ctypes.POINTER(Buf), ctypes.POINTER(ctypes.c_uint32)]
        L.mrzo_chunk_bytes.argtypes = [ctypes.c_int64]
        L.mrzo_crc32.restype = ctypes.c_uint32
        L.mrzo_crc32.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int64]
        L.mrzo_compress.argtypes = [ctypes.POINTER(Params), ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(Buf),
                                    ctypes.POINTER(OStats), ctypes.c_char_p]
        L.mrzo_compress_stream.argtypes = [ctypes.POINTER(Params), ctypes.c_char_p, ctypes.c_int64, ctypes.c_int,
                                           ctypes.POINTER(Buf), ctypes.POINTER(OStats), ctypes.c_char_p,
                                           ctypes.POINTER(ctypes.c_int)]
        L.mrzo_decompress.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(Buf)]
        L.mrzo_frame.argtypes = [ctypes.POINTER(Params), ctypes.c_int64, ctypes.POINTER(ChunkStreams), ctypes.c_int,
                                 ctypes.c_char_p, ctypes.POINTER(Buf)]
        L.mrzo_plan.restype = ctypes.c_int64
        L.mrzo_plan.argtypes = [ctypes.POINTER(Params), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        L.mrzo_lz4_compress.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.mrzo_lz4_compressed_size.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        L.mrzo_lz4_compresses.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int]
        L.mrzo_buf_free.argtypes = [ctypes.POINTER(Buf)]
        L.mrzo_rs_encoded_size.restype = ctypes.c_int64
        L.mrzo_rs_encoded_size.argtypes = [ctypes.c_int64]
        L.mrzo_rs_encode.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p]

    @staticmethod
    def _take(L, b):
        out = ctypes.string_at(b.p, b.len) if b.len else b""
        L.mrzo_buf_free(ctypes.byref(b))
        return out

    def hash_index(self):
        H = (ctypes.c_int64 * 256)()
        self.L.mrzo_hash_index(H)
        return list(H)

    def rzip_chunk(self, data, level=7, victim_round=0, want_table=False, bytewise=False):
        """-> dict(s0, s1, crc, stats, victim_round, min_mask, hash_count[, table]); bytewise: the reference's
        byte-at-a-time match extension (same results, timing variant)"""
        L = self.L
        m = ctypes.c_void_p(L.mrzo_matcher_new(level))
        assert m
        try:
            L.mrzo_matcher_set_victim_round(m, victim_round)
            if bytewise:
                L.mrzo_matcher_set_bytewise(m, 1)
            s0, s1, crc = Buf(), Buf(), ctypes.c_uint32()
            cb = L.mrzo_chunk_bytes(len(data))
            rc = L.mrzo_rzip_chunk(m, data, len(data), cb, ctypes.byref(s0), ctypes.byref(s1), ctypes.byref(crc))
            assert rc == 0
            r = dict(s0=self._take(L, s0), s1=self._take(L, s1), crc=crc.value,
                     stats=L.mrzo_matcher_stats(m).contents.as_dict(),
                     victim_round=L.mrzo_matcher_get_victim_round(m), min_mask=L.mrzo_matcher_min_mask(m),
                     hash_count=L.mrzo_matcher_hash_count(m))
            if want_table:
                ns = ctypes.c_int64()
                p = L.mrzo_matcher_table(m, ctypes.byref(ns))
                r["table"] = ctypes.string_at(p, ns.value * 16)
            return r
        finally:
            L.mrzo_matcher_free(m)

    def compress(self, data, level=7, window=0, unlimited=False, ramsize=60 << 30):
        prm = Params(level, window, 1 if unlimited else 0, ramsize, 4096)
        out, st, md5 = Buf(), OStats(), ctypes.create_string_buffer(16)
        rc = self.L.mrzo_compress(ctypes.byref(prm), data, len(data), ctypes.byref(out), ctypes.byref(st), md5)
        assert rc == 0, rc
        return self._take(self.L, out), st.as_dict(), md5.raw

    def compress_stream(self, data, to_stdout=False, level=7, window=0, ramsize=60 << 30):
        """`mrzip -n` reading STDIN: returns (archive, stats, md5, number of chunks)."""
        prm = Params(level, window, 0, ramsize, 4096)
        out, st, md5 = Buf(), OStats(), ctypes.create_string_buffer(16)
        nch = ctypes.c_int(0)
        rc = self.L.mrzo_compress_stream(ctypes.byref(prm), data, len(data), 1 if to_stdout else 0, ctypes.byref(out),
                                         ctypes.byref(st), md5, ctypes.byref(nch))
        assert rc == 0, rc
        return self._take(self.L, out), st.as_dict(), md5.raw, nch.value

    def decompress(self, mrz):
        out = Buf()
        rc = self.L.mrzo_decompress(mrz, len(mrz), ctypes.byref(out))
        data = self._take(self.L, out)
        return rc, data

    def frame(self, st_size, chunks, md5, level=7, window=0, unlimited=False, ramsize=60 << 30):
        prm = Params(level, window, 1 if unlimited else 0, ramsize, 4096)
        arr = (ChunkStreams * len(chunks))()
        for i, (csz, s0, s1) in enumerate(chunks):
            arr[i] = ChunkStreams(csz, s0, len(s0), s1, len(s1))
        out = Buf()
        rc = self.L.mrzo_frame(ctypes.byref(prm), st_size, arr, len(chunks), md5, ctypes.byref(out))
        assert rc == 0, rc
        return self._take(self.L, out)

    def plan(self, st_size, level=7, window=0, unlimited=False, ramsize=60 << 30):
        prm = Params(level, window, 1 if unlimited else 0, ramsize, 4096)
        bs = ctypes.c_int64()
        mc = self.L.mrzo_plan(ctypes.byref(prm), st_size, ctypes.byref(bs))
        return mc, bs.value

    def crc32(self, data):
        return self.L.mrzo_crc32(0, data, len(data))

    def lz4_compress(self, data, cap):
        dst = ctypes.create_string_buffer(max(1, cap))
        r = self.L.mrzo_lz4_compress(data, len(data), dst, cap)
        return r, dst.raw[:r]

    def lz4_size(self, data, cap=None):
        return self.L.mrzo_lz4_compressed_size(data, len(data), len(data) + 1 if cap is None else cap)

    def lz4_compresses(self, data, threshold=100):
        return self.L.mrzo_lz4_compresses(data, len(data), threshold)

    def rs_encode(self, data):
        out = ctypes.create_string_buffer(self.L.mrzo_rs_encoded_size(len(data)))
        assert self.L.mrzo_rs_encode(data, len(data), out) == 0
        return out.raw

    def rs_parity(self, row):
        p = ctypes.create_string_buffer(32)
        self.L.mrzo_rs_parity(row, p)
        return p.raw

    def blake2b(self, data, outlen=64, pieces=None):
        class St(ctypes.Structure):
            _fields_ = [("h", ctypes.c_uint64 * 8), ("t", ctypes.c_uint64 * 2), ("buf", ctypes.c_uint8 * 128),
                        ("buflen", ctypes.c_size_t), ("outlen", ctypes.c_size_t)]
        s = St()
        self.L.mrzo_blake2b_init(ctypes.byref(s), ctypes.c_size_t(outlen))
        for part in (pieces if pieces is not None else [data]):
            self.L.mrzo_blake2b_update(ctypes.byref(s), part, ctypes.c_size_t(len(part)))
        out = ctypes.create_string_buffer(64)
        self.L.mrzo_blake2b_final(ctypes.byref(s), out)
        return out.raw[:outlen]


# ---- seeded workloads ----------------------------

def golden_inputs():
    """The six inputs of SURVEY 8c (pure functions of the seeds given there)."""
    random.seed(42)
    blk42 = bytes(random.getrandbits(8) for _ in range(4096))
    random.seed(1234)
    base = bytearray(random.getrandbits(8) for _ in range(65536))
    syn = bytearray()
    for i in range(1024):
        b = bytearray(base)
        b[(i * 37) % 65536] = i & 0xFF
        syn += b
    return {
        "empty": b"",
        "range30": bytes(range(30)),
        "a1000": b"a" * 1000,
        "range256x64": bytes(range(256)) * 64,
        "seed42x64": blk42 * 64,
        "syn64": bytes(syn),
    }


from modern_rzip_amd.workloads import zipf_text, rep64k, tar_like  # noqa: E402,F401
from modern_rzip_amd.workloads import noise as xorshift_noise  # noqa: E402,F401
