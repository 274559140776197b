import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modern_rzip_amd as m
from tests import _util, _parity
lib = m.load_library()
o = _util.Oracle("oracle/liboracle.so")
import torch
for name, data in (("text64k", _util.zipf_text(1 << 16)), ("rep2m", _util.rep64k(32))):
    try:
        _parity.check_chunk(lib, o, data)
        print(name, "OK", flush=True)
    except BaseException as e:
        print(name, "FAIL", repr(e)[:300], flush=True)

r = o.rzip_chunk(_util.zipf_text(1 << 16))
s0 = r["s0"]; print("oracle first records:", s0[:40].hex())
