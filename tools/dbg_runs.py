"""Run one chunk several times on a -DMRZ_DBG_HITS build and compare the runs position by position (hits / misses per
look-up, signatures of every preparation and pre-commit): finds where two runs that should be identical part ways.
  hipcc ... -DMRZ_DBG_HITS -shared -o tools/_ab/lib_dbg.so <SRCS>;  python tools/dbg_runs.py FILE LEVEL VICTIM_ROUND"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import modern_rzip_amd as m
from tests import _util
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
o = _util.Oracle(os.path.join(root, 'oracle', 'liboracle.so'))
data = open(sys.argv[1], 'rb').read(); level = int(sys.argv[2]); vr = int(sys.argv[3])
lib = m.load_library(os.path.join(root, 'tools', '_ab', os.environ.get('DBGLIB', 'lib_dbg.so')))
want = o.rzip_chunk(data, level=level, victim_round=vr)
runs = []
for rep in range(8):
    N = len(data) + 64
    dbg = torch.zeros(4 * N, dtype=torch.int32, device='cuda')
    lib.mrz_dbg_hits_set.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    assert lib.mrz_dbg_hits_set(dbg.data_ptr(), N) == 0
    with m.RzipContext(level=level, max_chunk=len(data), lib=lib) as ctx:
        ctx.victim_round = vr
        res, s0, s1 = ctx.rzip_chunk(data)
    ok = res.stats.as_dict() == want['stats']
    arr = dbg.cpu().numpy().view(np.uint32).copy()
    runs.append((ok, arr, res.stats.tag_hits, res.stats.tag_misses))
    print('run', rep, 'OK' if ok else 'FAIL', res.stats.tag_hits, res.stats.tag_misses, flush=True)
good = [r for r in runs if r[0]]
bad = [r for r in runs if not r[0]]
if good and bad:
    g = good[0][1]
    for k, (ok, arr, h, ms) in enumerate(bad[:3]):
        for pl, nm in ((1, 'prep'), (2, 'precommit'), (3, 'window')):
            dd = np.nonzero(arr[pl * N:(pl + 1) * N] != g[pl * N:(pl + 1) * N])[0]
            print('bad run', k, 'plane', nm, 'differs at', len(dd), 'positions; first', [(int(x), hex(int(arr[pl * N + x])), hex(int(g[pl * N + x]))) for x in dd[:6]])
        d = np.nonzero(arr[:N] != g[:N])[0]
        print('bad run', k, 'differs at', len(d), 'positions')
        for q in d[:12]:
            a, b = int(arr[q]), int(g[q])
            print('   q', int(q), 'bad: visits w/b/c', (a >> 28) & 15, (a >> 24) & 15, 'hits', (a >> 16) & 255, 'miss', a & 0xffff,
                  '| good: visits', (b >> 28) & 15, (b >> 24) & 15, 'hits', (b >> 16) & 255, 'miss', b & 0xffff)
