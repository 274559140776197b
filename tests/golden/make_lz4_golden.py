"""Generates tests/golden/lz4_sizes.json from the system liblz4.so.1 (1.9.3 in
the build image): what LZ4_compress_default(src, dst, n, n + 1) returns for a
set of seeded inputs, plus the reference gate's verdict computed from those
sizes.  lz4 is an un-vendored submodule of the reference, so these vectors pin
the version the oracle restates.  Run:  python tests/golden/make_lz4_golden.py"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _util  # noqa: E402


def cases():
    yield "zeros_100", bytes(100)
    yield "zeros_70000", bytes(70000)
    yield "text_64", _util.zipf_text(64, seed=3)
    yield "text_4096", _util.zipf_text(4096, seed=3)
    yield "text_65546", _util.zipf_text(65546, seed=4)
    yield "text_65547", _util.zipf_text(65547, seed=4)
    yield "text_300000", _util.zipf_text(300000, seed=5)
    yield "noise_1000", _util.xorshift_noise(1000, seed=6)
    yield "noise_200000", _util.xorshift_noise(200000, seed=7)
    yield "rep_4x65536", _util.rep64k(4, seed=8)
    yield "tar_400000", _util.tar_like(400000, seed=9)
    yield "ctl_19048", _util.zipf_text(19048, seed=10)


def main():
    z = ctypes.CDLL("liblz4.so.1")
    z.LZ4_versionString.restype = ctypes.c_char_p
    out = {"_liblz4": z.LZ4_versionString().decode(), "sizes": {}}
    for name, data in cases():
        dst = ctypes.create_string_buffer(len(data) + 1)
        out["sizes"][name] = z.LZ4_compress_default(data, dst, len(data), len(data) + 1)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lz4_sizes.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)


if __name__ == "__main__":
    main()
